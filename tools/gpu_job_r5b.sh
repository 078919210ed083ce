#!/bin/bash
# round 5: the fused producer (resadd_xtx_kernel): tests, kernel-trace durations beside round 4's chain, PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests/test_producer_gpu.py -q -m gpu --tb=short -x -s 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/r5b_producer.txt
cd /tmp && export TMPDIR=/tmp
for MODE in resaddstats resaddsplit resaddstatsk2 resaddsplitk1k2 k1wsplit; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5b_${MODE}_stats -o s -- python3 $R/tools/stage_only.py 20 $MODE > /dev/null 2>&1
done
cd $R
bash tools/gpu_job_pmc_mode.sh resaddstats r5b_rx
python - <<'PY' > gpurun_out/r5b_summary.txt
import csv, glob, collections
for mode in "resaddstats resaddsplit resaddstatsk2 resaddsplitk1k2 k1wsplit".split():
    fs = glob.glob(f'gpurun_out/r5b_{mode}_stats/**/*kernel_trace.csv', recursive=True)
    if not fs:
        print(mode, "no trace"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[r['Kernel_Name'][:90]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    tot = 0
    for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if len(d) < 15: continue
        d = sorted(d); tot += d[len(d)//2] * (len(d) / 20.0)
        print(f"{mode:16s} {k:90s} n={len(d):3d} min {d[0]:7.1f} med {d[len(d)//2]:7.1f} avg {sum(d)/len(d):7.1f} max {d[-1]:7.1f}")
    print(f"{mode:16s} sum of medians per call: {tot:.1f} us")
PY
tail -40 gpurun_out/r5b_producer.txt; cat gpurun_out/r5b_summary.txt
