#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab
for v in 1 0 1 0; do
WC_WHITEN=$v timeout 900 python bench.py --steps 20 --warmup 5 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WC_WHITEN=$v', d['value'], d['ms_per_step'], 'eager', d['eager_launch']['ms_per_step'], 'ratio1', d['training_ratio_1']['ms_per_step'], 'fwd site', d['roofline']['forward_site_us'])"
done
timeout 1200 python -m pytest tests/test_layers_gpu.py tests/test_fast_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "registered_operator or whiten or seed_sweep" 2>&1 | tail -4
