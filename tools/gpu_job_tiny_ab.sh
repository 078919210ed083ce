#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in 1 0; do
WC_WHITEN=$v timeout 1200 python bench.py --config tinyimagenet_cond_sa --steps 6 --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WC_WHITEN=$v', d['value'], d['ms_per_step'])"
done
