#!/bin/bash
# gpurun_out/<tag>_* (tools/gpu_job_r5_final.sh: tag r5z) -> profiles/r5_*: the summaries that are committed
# usage: tools/collect_r5_profiles.sh [tag]
set -e
cd "$(dirname "$0")/.."
T=${1:-r5z}
[ -f gpurun_out/${T}_bench.json ] && tail -1 gpurun_out/${T}_bench.json > profiles/r5_bench_line.json
[ -f gpurun_out/${T}_step.md ] && cp gpurun_out/${T}_step.md profiles/r5_bench_step_steady_state.md
[ -f gpurun_out/${T}_site_timeline.txt ] && cp gpurun_out/${T}_site_timeline.txt profiles/r5_site_timeline_layer_path.txt
[ -f gpurun_out/${T}_summary.txt ] && cp gpurun_out/${T}_summary.txt profiles/r5_kernel_trace_stages_128x32x32x256.txt
[ -f gpurun_out/${T}_other_configs.txt ] && cp gpurun_out/${T}_other_configs.txt profiles/r5_other_configs.txt
[ -f gpurun_out/${T}_tests.txt ] && grep -E "passed|failed|error" gpurun_out/${T}_tests.txt | tail -3 > profiles/r5_gpu_suite.txt
P="python tools/summarize_pmc2.py gpurun_out"
M=131072; C=256; XB=$((M*C*4)); TAB=$(((C*C+C)*4))
if [ -d gpurun_out/${T}_k3splitmask_pmc_fetch ]; then
$P ${T}_k3splitmask "apply_split_kernel" $((2*XB+TAB+XB/32)) profiles/r5_apply_k3splitmask_pmc.json "K3 as the layers run it at Generator.BN.Final (128x32x32x256): apply_split_kernel<256, false, true, false> -- pre-split planes in (the residual add's), fp32 out, ReLU + one-bit mask (wc_apply_split_ex_f16x2); SURVEY 8d input (cond ~1e6), a loop of this kernel alone (tools/stage_only.py k3splitmask): algorithmic bytes 2*M*C*4 + table + M*C/8 of mask" > /dev/null
fi
if [ -d gpurun_out/${T}_resaddstats_pmc_fetch ]; then
$P ${T}_resaddstats "resadd_xtx_kernel<256, false, false>" $((XB+XB/4+XB)) profiles/r5_resadd_xtx_pmc.json "The producer feeding K1 (wc_resadd_stats_split_f32) at 128x32x32x256: resadd_xtx_kernel<256, false, false> -- h + up(s) summed, centred, scaled, split; planes out; the next site's covariance partials from the same pass (sampling inside the kernel).  Algorithmic bytes = h + s + planes = 2.25 * M*C*4 (the 38 MB of float64 partials are extra).  A loop of this call alone (tools/stage_only.py resaddstats); the gated second launch (resadd_xtx_kernel<256, false, true>, 4.7 us) is not in these counters" > /dev/null
fi
ls -la profiles | grep r5_
