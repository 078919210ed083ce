#!/bin/bash
# rocprofv3 kernel trace of every WC site (tools/sites_all.py) -> gpurun_out/r6_sites_all.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sites_prof -o s -- python3 $R/tools/sites_all.py > $R/gpurun_out/sites_all.log 2>&1
cd $R
python tools/sites_all_print.py gpurun_out/sites_prof/s_kernel_trace.csv gpurun_out/sites_all.log > gpurun_out/r6_sites_all.txt 2>&1; cat gpurun_out/r6_sites_all.txt | cut -c1-200
