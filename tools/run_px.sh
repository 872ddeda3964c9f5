mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_peer.py -x -q -s 2>&1 | tail -25 > gpurun_out/r04/px_tests.txt
for k in 1 8; do
  timeout 600 python bench.py --gpus 2 --same-device --steps 2 --warmup 1 --no-cpu-baseline --replicas 0 --no-extra --sync-every $k > gpurun_out/r04/bench_2ranks_same_device_k$k.json 2> gpurun_out/r04/bench_2ranks_k$k.err
done
python - <<'PY'
import json
for k in (1,8):
    try:
        d=json.loads(open('gpurun_out/r04/bench_2ranks_same_device_k%d.json'%k).read().strip().splitlines()[-1])
        print(k, d['value'], d['grad_steps_per_s'], d['roofline']['us_per_grad_step'], d.get('replicas_bit_identical_after_run'), json.dumps(d.get('exchange'))[:900])
    except Exception as e: print(k,'ERR',e)
PY
