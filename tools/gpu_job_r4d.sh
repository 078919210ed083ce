#!/bin/bash
# round 4: A/B of the step with and without the producer's planes (same box), kernel-trace summaries; rest of the GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
WC_SPLIT_PRODUCER=1 bash tools/gpu_job_step_profile.sh r4d_on
WC_SPLIT_PRODUCER=0 bash tools/gpu_job_step_profile.sh r4d_off
for v in 1 0 1 0; do WC_SPLIT_PRODUCER=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('producer=$v', d['value'], d['ms_per_step'], 'eager', d.get('eager_launch',{}).get('ms_per_step'))"; done > gpurun_out/r4d_ab.txt 2>&1
python tools/k6_spread.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4d_k6_spread.txt
timeout 2400 python -m pytest tests -q -m gpu --tb=short --deselect tests/test_producer_gpu.py --deselect tests/test_split_gpu.py --deselect tests/test_fast_gpu.py --deselect tests/test_parity_gpu.py -k "not test_bench" 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r4d_tests.txt
cat gpurun_out/r4d_ab.txt gpurun_out/r4d_k6_spread.txt; tail -6 gpurun_out/r4d_tests.txt; cat gpurun_out/r4d_on_step.log gpurun_out/r4d_off_step.log | grep ms/step
