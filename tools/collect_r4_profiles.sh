#!/bin/bash
# gpurun_out/<tag>_* (tools/gpu_job_r4_final.sh: tag r4z; the mid-round job tools/gpu_job_r4f.sh: tag r4f) -> profiles/r4_*: the summaries that are committed
# usage: tools/collect_r4_profiles.sh [tag]
set -e
cd "$(dirname "$0")/.."
T=${1:-r4z}
[ -f gpurun_out/${T}_bench.json ] && tail -1 gpurun_out/${T}_bench.json > profiles/r4_bench_line.json
[ -f gpurun_out/${T}_step.md ] && cp gpurun_out/${T}_step.md profiles/r4_bench_step_steady_state.md
[ -f gpurun_out/${T}_site_timeline.txt ] && cp gpurun_out/${T}_site_timeline.txt profiles/r4_site_timeline_layer_path.txt
[ -f gpurun_out/${T}_summary.txt ] && cp gpurun_out/${T}_summary.txt profiles/r4_kernel_trace_stages_128x32x32x256.txt
[ -f gpurun_out/${T}_k6_spread.txt ] && cp gpurun_out/${T}_k6_spread.txt profiles/r4_k6_spread.txt
[ -f gpurun_out/${T}_k3_zero_planes.txt ] && cp gpurun_out/${T}_k3_zero_planes.txt profiles/r4_k3_zero_planes.txt
[ -f gpurun_out/${T}_k2_stamps.txt ] && cp gpurun_out/${T}_k2_stamps.txt profiles/r4_k2_wave_timeline.txt
[ -f gpurun_out/${T}_k2_pipe_check.txt ] && cp gpurun_out/${T}_k2_pipe_check.txt profiles/r4_k2_pipe_check.txt
[ -f gpurun_out/${T}_k6_variants.txt ] && cp gpurun_out/${T}_k6_variants.txt profiles/r4_k6_ablations.txt
[ -f gpurun_out/${T}_other_configs.txt ] && cp gpurun_out/${T}_other_configs.txt profiles/r4_other_configs.txt
[ -f gpurun_out/${T}_tests.txt ] && grep -E "passed|failed|error" gpurun_out/${T}_tests.txt | tail -3 > profiles/r4_gpu_suite.txt
if [ -f gpurun_out/r4_seed_sweep_base.txt ]; then
  { echo "# tools/seed_sweep.py (tools/gpu_job_r4_sweep.sh): relative errors (max-abs / max-abs reference) against the float64 oracle, full-size sites";
    for f in base base_planes families families_planes families_nocomp; do echo; echo "## $f"; grep -v "^$" gpurun_out/r4_seed_sweep_$f.txt; done; } > profiles/r4_seed_sweep.txt
fi
P="python tools/summarize_pmc2.py gpurun_out"
M=131072; C=256; XB=$((M*C*4)); TAB=$(((C*C+C)*4))
if [ -d gpurun_out/${T}_k3splitmask_pmc_fetch ]; then
$P ${T}_k3splitmask "apply_split_kernel" $((2*XB+TAB+XB/32)) profiles/r4_apply_k3splitmask_pmc.json "K3 as the layers run it at Generator.BN.Final (128x32x32x256): apply_split_kernel<256, false, true, false> -- pre-split planes in (the residual add's), fp32 out, ReLU + one-bit mask (wc_apply_split_ex_f16x2); SURVEY 8d input (cond ~1e6), a loop of this kernel alone (tools/stage_only.py k3splitmask): algorithmic bytes 2*M*C*4 + table + M*C/8 of mask" > /dev/null
$P ${T}_k3mask "affine_ring_kernel" $((2*XB+TAB+XB/32)) profiles/r4_apply_k3mask_pmc.json "K3 on an fp32 input with ReLU + one-bit mask (wc_apply_mask_f32: sites whose input is not a residual add's planes), same site: algorithmic bytes 2*M*C*4 + table + M*C/8" > /dev/null
$P ${T}_k3splitplanes "apply_split_kernel" $((2*XB+TAB+XB/32)) profiles/r4_apply_k3splitplanes_pmc.json "K3 planes in, ReLU + mask + the next convolution's planes out (bn1 of a block): TWO dispatches per call -- the pass and the gated launch that leaves at once -- so every per-dispatch average here is half the pass's (its kernel time is kernel_max_us)" > /dev/null
$P ${T}_k1wsplit "xtx_split_kernel" $XB profiles/r4_k1_xtx_split_pmc.json "K1 on the pre-split planes as the layers run it (wc_whiten_split_f16x2) at 128x32x32x256: algorithmic bytes = M*C*4" > /dev/null
$P ${T}_k1 "xty_f16x3_kernel" $XB profiles/r4_k1_xty_pmc.json "K1 reduction on an fp32 input (sites not fed by a residual add) with the conflict-free stage-write swizzle of round 4: algorithmic bytes = M*C*4" > /dev/null
$P ${T}_k4bits "xty_f16x3_kernel<256, true" $((2*XB+XB/32)) profiles/r4_k4_bits_pmc.json "K4 with the 1-bit ReLU mask, fp32 x (wc_bwd_reduce_bits_f32), conflict-free stage-write swizzle: algorithmic bytes = 2*M*C*4 + M*C/8" > /dev/null
$P ${T}_k4xsplit "xty_f16x3_kernel<256, true" $((2*XB+XB/32)) profiles/r4_k4_xsplit_pmc.json "K4 with x read from the producer's planes (wc_bwd_reduce_xsplit_f32; X staged by byte permutes): algorithmic bytes = 2*M*C*4 + M*C/8" > /dev/null
$P ${T}_k6bits "onepass_ring_kernel" $((3*XB+XB/32)) profiles/r4_k6_onepass_bits_pmc.json "K6 in one pass, fp32 x, 1-bit mask (wc_bwd_apply_bits_f32): algorithmic bytes = 3*M*C*4 + M*C/8" > /dev/null
$P ${T}_k6xsplit "onepass_ring_kernel" $((3*XB+XB/32)) profiles/r4_k6_onepass_xsplit_pmc.json "K6 in one pass with x read from the producer's planes (wc_bwd_apply_xsplit_f32): algorithmic bytes = 3*M*C*4 + M*C/8" > /dev/null
fi
ls -la profiles | grep r4_
