#!/bin/bash
# the other three BASELINE configurations through bench.py (6-10 timed steps each, no CPU baseline) -> gpurun_out/r3z_configs.txt
cd "$GRAFT_REPO_ROOT" || exit 1
for c in cifar10_cond stl10_uncond tinyimagenet_cond_sa; do
timeout 1200 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config']['launch'])"
done | tee gpurun_out/r3z_configs.txt
