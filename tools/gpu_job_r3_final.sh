#!/bin/bash
# round 3 evidence run: the whole GPU suite, the bench line, the steady-state step profile, the headline-site kernel stats,
# per-kernel PMC passes (one kernel per run) and the layer-path timeline.  Everything lands under gpurun_out/r3z_*.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r3z_gputests.log 2>&1; tail -3 gpurun_out/r3z_gputests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3z_bench.json 2> gpurun_out/r3z_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3z_step -o s -- python3 $R/tools/step_only.py 3 > $R/gpurun_out/r3z_step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3z_site -o s -- python3 $R/tools/kernel_bench.py > $R/gpurun_out/r3z_site.log 2>&1
cd $R
python tools/summarize_profile.py gpurun_out/r3z_step gpurun_out/r3z_step.md gap > /dev/null
python tools/summarize_profile.py gpurun_out/r3z_site gpurun_out/r3z_site.md > /dev/null
for m in k3 k3split k3planes k1 k1split k4mask k6; do bash tools/gpu_job_pmc_mode.sh $m r3z_$m; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3z_site_tl -o s -- python3 $R/tools/site_timeline.py > /dev/null 2>&1
cd $R
python tools/site_timeline_print.py gpurun_out/r3z_site_tl/s_kernel_trace.csv > gpurun_out/r3z_site_timeline.txt 2>&1
python tools/k3_inflow.py > gpurun_out/r3z_k3_inflow.txt 2>&1
python tools/seed_sweep.py 5 > gpurun_out/r3z_seed_sweep.txt 2>&1
ls gpurun_out | grep r3z | head -60
