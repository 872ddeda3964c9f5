python bench.py --steps 3 --warmup 1 --no-cpu-baseline --replicas 0 --no-extra --no-measure-traffic --no-early-stop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['value'], d['roofline']['us_per_grad_step'], d.get('early_stop_env_steps_per_s'))"
python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --replicas 0 --no-extra --no-measure-traffic --no-early-stop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['value'], d['roofline']['us_per_grad_step'])"
