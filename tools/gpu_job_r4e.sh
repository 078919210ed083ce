#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_producer_gpu.py -q --tb=short -x 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r4e_tests.txt
tail -12 gpurun_out/r4e_tests.txt
for v in 1 0 1 0; do WC_BWD_XSPLIT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xsplit=$v', d['value'], d['ms_per_step'], 'eager', d.get('eager_launch',{}).get('ms_per_step'))"; done > gpurun_out/r4e_ab.txt 2>&1
cat gpurun_out/r4e_ab.txt
