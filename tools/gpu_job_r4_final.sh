#!/bin/bash
# round 4, evidence of the final build -> gpurun_out/r4z_*; tools/collect_r4_profiles.sh turns them into profiles/r4_*
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=$PWD
timeout 2400 python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r4z_tests.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r4z_bench.json 2> gpurun_out/r4z_bench.err
bash tools/gpu_job_step_profile.sh r4z
for MODE in k3splitmask k3mask k1wsplit k1 k4bits k4xsplit k6bits k6xsplit k3splitplanes; do bash tools/gpu_job_pmc_mode.sh $MODE r4z_$MODE; done
cd /tmp && export TMPDIR=/tmp
for MODE in resadd resaddsplit resaddtorch k3split k3planes; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4z_${MODE}_stats -o s -- python3 $R/tools/stage_only.py 20 $MODE > /dev/null 2>&1
done
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4z_site_tl -o s -- python3 $R/tools/site_timeline_r4.py > /dev/null 2>&1
cd $R
python tools/site_timeline_print.py gpurun_out/r4z_site_tl/s_kernel_trace.csv resadd_sample > gpurun_out/r4z_site_timeline.txt 2>&1
python tools/k6_spread.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4z_k6_spread.txt
python tools/k3_zero_planes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4z_k3_zero_planes.txt
# K2: every wave's barrier arrivals / departures (a -DCF_STAMPS=1 build of wc_small.hip made beforehand: tools/build_var.py wc_small st=-DCF_STAMPS=1) and the per-call times
[ -f wc_gan_amd/csrc/build/var/lib_st.so ] && python tools/k2_stamps.py 256 wc_gan_amd/csrc/build/var/lib_st.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r4z_k2_stamps.txt
python tools/k2_pipe_check.py 10 2>&1 | grep -v amdgpu.ids > gpurun_out/r4z_k2_pipe_check.txt
python tools/k6_variants.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4z_k6_variants.txt
for CFG in cifar10_cond stl10_uncond tinyimagenet_cond_sa; do
  timeout 600 python bench.py --config $CFG --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFG', d['value'], 'images/sec', d['ms_per_step'], 'ms/step', d['config']['launch'])"
done > gpurun_out/r4z_other_configs.txt 2>&1
python - <<'PY' > gpurun_out/r4z_summary.txt
import csv, glob, collections
for mode in "k3splitmask k3mask k1wsplit k1 k4bits k4xsplit k6bits k6xsplit k3splitplanes resadd resaddsplit resaddtorch k3split k3planes".split():
    fs = glob.glob(f'gpurun_out/r4z_{mode}_stats/**/*kernel_trace.csv', recursive=True)
    if not fs:
        print(mode, "no trace"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[r['Kernel_Name'][:100]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if len(d) < 15: continue
        d = sorted(d)
        print(f"{mode:14s} {k:100s} n={len(d):3d} min {d[0]:7.1f} med {d[len(d)//2]:7.1f} avg {sum(d)/len(d):7.1f} max {d[-1]:7.1f}")
PY
tail -4 gpurun_out/r4z_tests.txt; cat gpurun_out/r4z_summary.txt gpurun_out/r4z_other_configs.txt gpurun_out/r4z_k6_spread.txt gpurun_out/r4z_k3_zero_planes.txt gpurun_out/r4z_k2_pipe_check.txt; tail -c 1500 gpurun_out/r4z_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4z_bench.json').read().strip().splitlines()[-1]); r = d['roofline']
print("value", d['value'], "ms", d['ms_per_step'], "eager", d.get('eager_launch'), "ratio1", d.get('training_ratio_1'))
for k in ('kernel', 'launch_us', 'frac', 'frac_of_stream_copy', 'frac_of_stream_copy_loop', 'stream_copy_loop_GBs', 'back_to_back_us', 'in_flow_us', 'forward_site_us', 'forward_site_fp32_input_us', 'producer_us', 'forward_site_plus_producer_us', 'stream_copy_GBs', 'error'):
    print(k, r.get(k))
for k, v in r.get('k3_kernels', {}).items(): print("  ", v['launch_us'], v['frac_of_stream_copy'], k[:80])
print(d.get('cpu_baseline'))
PY
