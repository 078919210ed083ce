#!/bin/bash
# nontemporal stores: K3 + stream copy by HIP events (tools/k3_time.py) and the step (bench.py --no-cpu-baseline), builds alternated
R=$PWD
for round in 1 2; do
  for v in "" "-DWC_NT_STORE=1 -DWC_NT_COPY=1"; do
    WC_EXTRA_FLAGS="$v" python -m wc_gan_amd.build --force > /dev/null 2>&1
    echo "== flags [$v]"
    timeout 120 python tools/k3_time.py 2>&1 | grep "K3 us"
    timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['value'], d['ms_per_step'], 'K3', d['roofline']['launch_us'], 'copy GB/s', d['roofline']['stream_copy_GBs'])"
  done
done
python -m wc_gan_amd.build --force > /dev/null 2>&1
