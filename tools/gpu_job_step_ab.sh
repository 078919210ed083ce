#!/bin/bash
# the step (bench.py --no-cpu-baseline) for two builds, alternated: $1 = extra flags of variant B
for round in 1 2 3; do
  for v in "" "$1"; do
    WC_EXTRA_FLAGS="$v" python -m wc_gan_amd.build --force > /dev/null 2>&1
    timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] step', d['value'], d['ms_per_step'], 'K3', d['roofline']['launch_us'], 'K6', d['roofline']['site_stages']['K6 wc_bwd_apply_f32']['us'])"
  done
done
python -m wc_gan_amd.build --force > /dev/null 2>&1
