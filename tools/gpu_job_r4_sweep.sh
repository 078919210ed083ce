#!/bin/bash
# round 4: the seed sweep on the gaussian-mix family (5 seeds, fp32 route; 3 seeds through the producer's planes) and
# on the families where the off-diagonal bias compensation over- / under-corrects (uniform, post-ReLU, heavy-tailed; C = 256 / 128 / 64);
# --ref32: the reference's unfused op order in fp32 on the host against the same float64 oracle, beside every family
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/seed_sweep.py 5 --ref32 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_seed_sweep_base.txt
python tools/seed_sweep.py 3 --planes 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_seed_sweep_base_planes.txt
python tools/seed_sweep.py 3 --families --ref32 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_seed_sweep_families.txt
python tools/seed_sweep.py 2 --families --planes 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_seed_sweep_families_planes.txt
WC_K1_NO_BIAS_COMP=1 python tools/seed_sweep.py 2 --families 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_seed_sweep_families_nocomp.txt
grep -h "WORST\|worst\|cond of\|REF32" gpurun_out/r4_seed_sweep_*.txt
