#!/bin/bash
# round 3: the whole GPU suite, the seed sweep, then the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3/pytest_gpu.txt
cat gpurun_out/r3/pytest_gpu.txt
timeout 1500 python tools/seed_sweep.py 5 > gpurun_out/r3/seed_sweep.txt 2>&1; echo "seed sweep rc $?"; grep "WORST\|worst" gpurun_out/r3/seed_sweep.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_line.json 2> gpurun_out/r3/bench_err.txt
tail -3 gpurun_out/r3/bench_err.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3/bench_line.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'launch', d['config']['launch'])
print('K3 split us', r['launch_us'], 'frac', r['frac'], 'of copy', r['frac_of_stream_copy'], '| fp32-input', r['fp32_input_kernel'])
print('site', r['forward_site_us'], 'on planes', r['forward_site_on_planes_us'])
for k, v in r['site_stages'].items(): print('  ', k, v)
print('ratio1', d.get('training_ratio_1'), 'eager', d.get('eager_launch'))
PY
