#!/bin/bash
# wave 0's work segments (the leaf) and the kernel time for leaf variants: flags in "$@", one variant per argument
for f in "$@"; do
  WC_EXTRA_FLAGS="-DCF_STAMPS=1 $f" python -m wc_gan_amd.build --force > /dev/null 2>&1
  echo "== $f"; timeout 120 python tools/k2_stamps.py 2>&1 | grep -A2 "wave 0" | cut -c1-260
done
python -m wc_gan_amd.build --force > /dev/null 2>&1
