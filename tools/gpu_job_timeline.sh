#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3z_site_tl -o s -- python3 $R/tools/site_timeline.py > /dev/null 2>&1
cd $R
python tools/site_timeline_print.py gpurun_out/r3z_site_tl/s_kernel_trace.csv > gpurun_out/r3z_site_timeline.txt 2>&1
tail -3 gpurun_out/r3z_site_timeline.txt; grep total gpurun_out/r3z_site_timeline.txt
