mkdir -p gpurun_out/r04
for k in 8 64 512; do
  timeout 600 python bench.py --gpus 2 --same-device --steps 2 --warmup 1 --no-cpu-baseline --replicas 0 --no-extra --sync-every $k > gpurun_out/r04/b2_k$k.json 2> gpurun_out/r04/b2_k$k.err
done
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --replicas 0 --no-extra --no-measure-traffic --no-early-stop > gpurun_out/r04/b1.json 2> gpurun_out/r04/b1.err
python - <<'PY'
import json
for k in ('b2_k8','b2_k64','b2_k512','b1'):
    try:
        d=json.loads(open('gpurun_out/r04/%s.json'%k).read().strip().splitlines()[-1])
        print(k, d['value'], d['roofline']['us_per_grad_step'], d.get('replicas_bit_identical_after_run'), d['phase_ms_per_iter'])
    except Exception as e: print(k,'ERR',e)
PY
