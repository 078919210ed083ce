#!/bin/bash
# round 4, mid-way evidence: the whole GPU suite, kernel traces of every stage variant, the site timeline, the step profile, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r4f_tests.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for MODE in k1 k1wsplit k3mask k3splitmask k3planes k3splitplanes k4bits k4xsplit k6bits k6xsplit resadd resaddsplit resaddtorch; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4f_${MODE}_stats -o s -- python3 $R/tools/stage_only.py 20 $MODE > /dev/null 2>&1
done
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4f_site_tl -o s -- python3 $R/tools/site_timeline_r4.py > /dev/null 2>&1
cd $R
python tools/site_timeline_print.py gpurun_out/r4f_site_tl/s_kernel_trace.csv resadd_sample > gpurun_out/r4f_site_timeline.txt 2>&1
bash tools/gpu_job_step_profile.sh r4f
python - <<'PY' > gpurun_out/r4f_summary.txt
import csv, glob, collections
for mode in "k1 k1wsplit k3mask k3splitmask k3planes k3splitplanes k4bits k4xsplit k6bits k6xsplit resadd resaddsplit resaddtorch".split():
    fs = glob.glob(f'gpurun_out/r4f_{mode}_stats/**/*kernel_trace.csv', recursive=True)
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
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4f_bench.json 2> gpurun_out/r4f_bench.err
tail -6 gpurun_out/r4f_tests.txt; cat gpurun_out/r4f_summary.txt; cat gpurun_out/r4f_site_timeline.txt; head -30 gpurun_out/r4f_step.md
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4f_bench.json').read().strip().splitlines()[-1]); r = d['roofline']
print("value", d['value'], "ms", d['ms_per_step'])
for k in ('kernel', 'launch_us', 'frac', 'frac_of_stream_copy', 'back_to_back_us', 'in_flow_us', 'forward_site_us', 'forward_site_fp32_input_us', 'producer_us', 'forward_site_plus_producer_us', 'error'):
    print(k, r.get(k))
for k, v in r.get('k3_kernels', {}).items(): print("  ", v['launch_us'], v['frac_of_stream_copy'], k[:80])
for k, v in r.get('site_stages', {}).items(): print("  ", v, k[:90])
PY
