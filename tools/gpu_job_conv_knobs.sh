#!/bin/bash
# development: the convolution kernels' split heuristics re-checked on the current step (bench.py per setting, one box)
cd "$GRAFT_REPO_ROOT"
run() { env "$@" timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'])"; }
for S in "$@"; do run $S; done
