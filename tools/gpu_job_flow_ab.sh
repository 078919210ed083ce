#!/bin/bash
# the layer's cold flow (tools/site_timeline.py under rocprofv3) for two builds, alternated: $1 = extra flags of variant B
R=$PWD
for round in 1 2; do
  for v in "" "$1"; do
    WC_EXTRA_FLAGS="$v" python -m wc_gan_amd.build --force > /dev/null 2>&1
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/flow_ab -o s -- python3 $R/tools/site_timeline.py > /dev/null 2>&1
    cd $R
    echo "[$v]"; python tools/site_timeline_print.py gpurun_out/flow_ab/s_kernel_trace.csv 2>&1 | grep -E "xty_f16x3_kernel<256, true|onepass|total" | head -3
    rm -rf gpurun_out/flow_ab
  done
done
python -m wc_gan_amd.build --force > /dev/null 2>&1
