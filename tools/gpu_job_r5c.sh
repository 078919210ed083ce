#!/bin/bash
# round 5: the fused producer with the sampling inside the kernel: tests, kernel-trace durations (A/B against the sampling launch), bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests/test_producer_gpu.py tests/test_sync_wc_two_ranks_gpu.py -q -m gpu --tb=short -x 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r5c_producer.txt
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  WC_RX_SAMPLE_KERNEL=$V rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5c_rs${V}_stats -o s -- python3 $R/tools/stage_only.py 20 resaddstats > /dev/null 2>&1
done
cd $R
python - <<'PY' > gpurun_out/r5c_summary.txt
import csv, glob, collections
for mode in "rs0 rs1".split():
    fs = glob.glob(f'gpurun_out/r5c_{mode}_stats/**/*kernel_trace.csv', recursive=True)
    if not fs:
        print(mode, "no trace"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[r['Kernel_Name'][:90]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    tot = 0
    for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if len(d) < 15: continue
        d = sorted(d); tot += d[len(d)//2] * (len(d) / 20.0)
        print(f"{mode:6s} {k:90s} n={len(d):3d} min {d[0]:7.1f} med {d[len(d)//2]:7.1f} avg {sum(d)/len(d):7.1f} max {d[-1]:7.1f}")
    print(f"{mode:6s} sum of medians per call: {tot:.1f} us")
PY
for V in 0 1 0 1; do WC_RX_SAMPLE_KERNEL=$V timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('sample_kernel=$V', d['value'], d['ms_per_step'], r.get('forward_site_us'), r.get('producer_us'), r.get('forward_site_plus_producer_us'), r.get('launch_us'), r.get('error'))"; done >> gpurun_out/r5c_summary.txt 2>&1
tail -12 gpurun_out/r5c_producer.txt; cat gpurun_out/r5c_summary.txt
