#!/bin/bash
# gpurun_out/r3z_* (tools/gpu_job_r3_final.sh, tools/gpu_job_r3_bits.sh) -> profiles/r3_*: the summaries that are committed
set -e
cd "$(dirname "$0")/.."
tail -1 gpurun_out/r3z_bench.json > profiles/r3_bench_line.json
cp gpurun_out/r3z_step.md profiles/r3_bench_step_steady_state.md
{ cat gpurun_out/r3z_site.md; echo; echo '## stage times of the same run (tools/kernel_bench.py, HIP events)'; echo '```'; cat gpurun_out/r3z_site.log | grep -v amdgpu.ids; echo '```'; } > profiles/r3_kernels_headline_site_128x32x32x256.md
cp gpurun_out/r3z_site_timeline.txt profiles/r3_site_timeline_layer_path.txt
grep "behind" gpurun_out/r3z_k3_inflow.txt > profiles/r3_k3_inflow.txt
grep -v amdgpu.ids gpurun_out/r3z_seed_sweep.txt > profiles/r3_seed_sweep.txt
P="python tools/summarize_pmc2.py gpurun_out"
$P r3z_k3 "affine_ring_kernel" 268698624 profiles/r3_apply_k3_pmc.json "K3, fp32-input ring kernel (wc_apply_f32 with plan: what the layers run), 128x32x32x256, SURVEY 8d input (cond ~1e6), a loop of this kernel alone (tools/stage_only.py k3): algorithmic bytes 2*M*C*4 + table" > /dev/null
$P r3z_k3split "apply_split_kernel" 268698624 profiles/r3_apply_k3split_pmc.json "K3 on the pre-split planes (wc_apply_split_f16x2, bias folded: one launch), same site and input, a loop of this kernel alone (tools/stage_only.py k3split): algorithmic bytes 2*M*C*4 + table" > /dev/null
$P r3z_k3planes "affine_ring_kernel" 272892928 profiles/r3_apply_k3planes_pmc.json "K3 + ReLU + 1-bit mask + the next convolution's fp16 planes (wc_apply_planes_f32): TWO dispatches per call -- the pass and the gated launch that leaves at once -- so every per-dispatch average here is half the pass's (its kernel time is kernel_max_us); algorithmic bytes M*C*4 in + M*C*4 of planes + M*C/8 of mask + table" > /dev/null
$P r3z_k1 "xty_f16x3_kernel" 134217728 profiles/r3_k1_xty_pmc.json "K1 reduction (fp32 input) at 128x32x32x256, cond ~1e6 input: algorithmic bytes = M*C*4 (x read once)" > /dev/null
$P r3z_k1split "xtx_split_kernel" 134217728 profiles/r3_k1_xtx_split_pmc.json "K1 on the pre-split planes (wc_stats_split_f16x2) at 128x32x32x256: algorithmic bytes = M*C*4 (the planes read once)" > /dev/null
if [ -d gpurun_out/r3z_k4bits_stats ]; then
$P r3z_k4bits "xty_f16x3_kernel<256, true" 272629760 profiles/r3_k4_bits_pmc.json "K4 with the 1-bit ReLU mask applied in its staging and NO masked copy written (wc_bwd_reduce_bits_f32) at 128x32x32x256: algorithmic bytes = 2*M*C*4 + M*C/8" > /dev/null
$P r3z_k6bits "onepass_ring_kernel" 406847488 profiles/r3_k6_onepass_bits_pmc.json "K6 in one pass applying the 1-bit mask to gy while it converts it (wc_bwd_apply_bits_f32) at 128x32x32x256: algorithmic bytes = 3*M*C*4 + M*C/8" > /dev/null
fi
$P r3z_k6 "onepass_ring_kernel" 402653184 profiles/r3_k6_onepass_pmc.json "K6 in one pass (no mask) at 128x32x32x256: algorithmic bytes = 3*M*C*4 (gy, x read once; dx written once)" > /dev/null
ls -la profiles | grep r3_
