#!/bin/bash
# rocprofv3 kernel trace of K2 alone (tools/factor_only.py: 20 calls at C = 256) -> gpurun_out/k2_prof.md
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k2_prof -o s -- python3 $R/tools/factor_only.py ${1:-256} ${2:-1} > $R/gpurun_out/k2_prof.log 2>&1
cd $R
python tools/summarize_profile.py gpurun_out/k2_prof gpurun_out/k2_prof.md >/dev/null; head -30 gpurun_out/k2_prof.md | cut -c1-170
