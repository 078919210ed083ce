#!/bin/bash
# round 2 evidence run: the whole GPU suite, the bench line, the steady-state step profile, K3 / K1 / K4 / K2 kernel profiles + PMC
R=$PWD
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r2z_gputests.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r2z_bench.json 2> gpurun_out/r2z_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2z_step -o s -- python3 $R/tools/step_only.py 3 > $R/gpurun_out/r2z_step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2z_site -o s -- python3 $R/tools/kernel_bench.py > $R/gpurun_out/r2z_site.log 2>&1
cd $R
python tools/summarize_profile.py gpurun_out/r2z_step gpurun_out/r2z_step.md gap > /dev/null
python tools/summarize_profile.py gpurun_out/r2z_site gpurun_out/r2z_site.md > /dev/null
bash tools/gpu_job_pmc.sh tools/apply_only.py r2z_k3 > /dev/null 2>&1
bash tools/gpu_job_pmc.sh tools/xty_only.py r2z_xty > /dev/null 2>&1
bash tools/gpu_job_pmc.sh tools/bwd_apply_only.py r2z_k6 > /dev/null 2>&1
bash tools/gpu_job_k2_profile.sh 256 1 > /dev/null 2>&1; cp gpurun_out/k2_prof.md gpurun_out/r2z_k2.md
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2z_site_tl -o s -- python3 $R/tools/site_timeline.py > /dev/null 2>&1
cd $R
python tools/site_timeline_print.py gpurun_out/r2z_site_tl/s_kernel_trace.csv > gpurun_out/r2z_site_timeline.txt 2>&1
