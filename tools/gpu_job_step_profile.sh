#!/bin/bash
# steady-state step profile only (rocprofv3 kernel trace of tools/step_only.py) -> gpurun_out/<tag>_step{,.md};  $2 = config name
R=$PWD; TAG=${1:-r2g}; CFG=${2:-cifar10_uncond}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_step -o s -- python3 $R/tools/step_only.py 3 $CFG > $R/gpurun_out/${TAG}_step.log 2>&1
cd $R
python tools/summarize_profile.py gpurun_out/${TAG}_step gpurun_out/${TAG}_step.md gap > /dev/null
