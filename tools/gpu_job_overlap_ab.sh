#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in 1 0 1 0; do
WC_OVERLAP_SHORTCUT=$v timeout 900 python bench.py --steps 20 --warmup 5 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WC_OVERLAP_SHORTCUT=$v', d['value'], d['ms_per_step'], 'eager', d['eager_launch']['ms_per_step'], 'ratio1', d['training_ratio_1']['ms_per_step'])"
done
