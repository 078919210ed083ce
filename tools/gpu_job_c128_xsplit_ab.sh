cd "$GRAFT_REPO_ROOT"
for V in 0 1 0 1; do WC_BWD_XSPLIT=$V timeout 600 python bench.py --config cifar10_cond --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cifar10_cond bwd_xsplit=$V', d['value'], d['ms_per_step'])"; done
for V in 0 1; do WC_BWD_XSPLIT=$V timeout 900 python bench.py --config tinyimagenet_cond_sa --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tinyimagenet bwd_xsplit=$V', d['value'], d['ms_per_step'])"; done
