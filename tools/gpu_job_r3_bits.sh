#!/bin/bash
# round 3, the bits-only backward kernels' PMC passes (one kernel per run)
cd "$GRAFT_REPO_ROOT" || exit 1
for m in k4bits k6bits; do bash tools/gpu_job_pmc_mode.sh $m r3z_$m; done
PYTHONUNBUFFERED=1 python tools/bwd_bits_time.py 128 32 256 300 > gpurun_out/r3z_bwd_bits.txt 2>&1; tail -8 gpurun_out/r3z_bwd_bits.txt
