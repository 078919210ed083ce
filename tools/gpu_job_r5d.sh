#!/bin/bash
# round 5: the history-scaled split (WC_SPLIT_HIST): per-call times, whole GPU suite with it on, bench A/B against the two-launch form
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/split_hist_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5d_split_hist_bench.txt
timeout 2400 python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r5d_suite.txt
for V in 0 1 0 1; do WC_SPLIT_HIST=$V timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('split_hist=$V', d['value'], d['ms_per_step'], r.get('launch_us'), r.get('error'))"; done > gpurun_out/r5d_summary.txt 2>&1
