mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_peer.py -q -x 2>&1 | tail -2
for k in 8; do
  timeout 600 python bench.py --gpus 2 --same-device --steps 3 --warmup 1 --no-cpu-baseline --replicas 0 --no-extra --sync-every $k > gpurun_out/r04/b2_k$k.json 2> gpurun_out/r04/b2_k$k.err
done
python - <<'PY'
import json
for k in ('b2_k8',):
    try:
        d=json.loads(open('gpurun_out/r04/%s.json'%k).read().strip().splitlines()[-1])
        print(k, d['value'], d['roofline']['us_per_grad_step'], d.get('replicas_bit_identical_after_run'), d['exchange']['flag_wait_per_rank'])
    except Exception as e: print(k,'ERR',e)
PY
