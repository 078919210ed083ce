#!/bin/bash
# round 3: the K3 -> convolution hand-off on and off: bench line of each, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3h
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3h/bench_on.json 2> gpurun_out/r3h/bench_on_err.txt
WC_HANDOFF=0 timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3h/bench_off.json 2> gpurun_out/r3h/bench_off_err.txt
python - <<'PY'
import json
for tag in ('on', 'off'):
    try:
        d = json.loads(open('gpurun_out/r3h/bench_%s.json' % tag).read().strip().splitlines()[-1])
        print(tag, 'value', d['value'], 'ms', d['ms_per_step'], 'launch', d['config']['launch'], 'ratio1', d.get('training_ratio_1'), 'eager', d.get('eager_launch'))
    except Exception as e:
        print(tag, 'failed', e)
PY
tail -3 gpurun_out/r3h/bench_on_err.txt
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3h/pytest_gpu.txt
cat gpurun_out/r3h/pytest_gpu.txt
