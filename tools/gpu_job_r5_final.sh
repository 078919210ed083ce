#!/bin/bash
# round 5, evidence of the final build -> gpurun_out/r5z_*; tools/collect_r5_profiles.sh turns them into profiles/r5_*
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=$PWD
timeout 2400 python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r5z_tests.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r5z_bench.json 2> gpurun_out/r5z_bench.err
bash tools/gpu_job_step_profile.sh r5z
for MODE in k3splitmask resaddstats; do bash tools/gpu_job_pmc_mode.sh $MODE r5z_$MODE; done
cd /tmp && export TMPDIR=/tmp
for MODE in resaddstats resaddsplit resaddstatsk2 resaddsplitk1k2 k3splitmask k3splitplanes k4xsplit k6xsplit; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5z_${MODE}_stats -o s -- python3 $R/tools/stage_only.py 20 $MODE > /dev/null 2>&1
done
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5z_site_tl -o s -- python3 $R/tools/site_timeline_r5.py > /dev/null 2>&1
cd $R
python tools/site_timeline_print.py gpurun_out/r5z_site_tl/s_kernel_trace.csv "resadd_xtx_kernel<256, false, false>" > gpurun_out/r5z_site_timeline.txt 2>&1
for CFG in cifar10_cond stl10_uncond tinyimagenet_cond_sa; do
  timeout 600 python bench.py --config $CFG --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFG', d['value'], 'images/sec', d['ms_per_step'], 'ms/step', d['config']['launch'])"
done > gpurun_out/r5z_other_configs.txt 2>&1
python - <<'PY' > gpurun_out/r5z_summary.txt
import csv, glob, collections
for mode in "resaddstats resaddsplit resaddstatsk2 resaddsplitk1k2 k3splitmask k3splitplanes k4xsplit k6xsplit".split():
    fs = glob.glob(f'gpurun_out/r5z_{mode}_stats/**/*kernel_trace.csv', recursive=True)
    if not fs:
        print(mode, "no trace"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[r['Kernel_Name'][:100]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    tot = 0
    for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if len(d) < 15: continue
        d = sorted(d); tot += d[len(d)//2] * (len(d) / 20.0)
        print(f"{mode:16s} {k:100s} n={len(d):3d} min {d[0]:7.1f} med {d[len(d)//2]:7.1f} avg {sum(d)/len(d):7.1f} max {d[-1]:7.1f}")
    print(f"{mode:16s} sum of medians per call: {tot:.1f} us")
PY
tail -4 gpurun_out/r5z_tests.txt; cat gpurun_out/r5z_summary.txt gpurun_out/r5z_other_configs.txt gpurun_out/r5z_site_timeline.txt; tail -c 800 gpurun_out/r5z_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5z_bench.json').read().strip().splitlines()[-1]); r = d['roofline']
print("value", d['value'], "ms", d['ms_per_step'], "eager", d.get('eager_launch'), "ratio1", d.get('training_ratio_1'))
for k in ('kernel', 'launch_us', 'frac', 'frac_of_isolated_copy', 'frac_of_loop_copy', 'isolated_copy_GBs', 'loop_copy_GBs', 'back_to_back_us', 'in_flow_us', 'forward_site_us', 'forward_site_round4_route_us', 'forward_site_fp32_input_us', 'producer_us', 'forward_site_plus_producer_us', 'error'):
    print(k, r.get(k))
for k, v in r.get('k3_kernels', {}).items(): print("  ", v['launch_us'], v['frac_of_isolated_copy'], v['frac_of_loop_copy'], k[:80])
print(d.get('cpu_baseline'))
PY
