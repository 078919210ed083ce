#!/bin/bash
# round 4, first GPU job: the producer path's tests, then kernel-trace timings + the LDS-conflict counters of K1 / K4 with the new
# stage-write swizzle, and the planes route's K3 epilogues
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_producer_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r4a_tests_producer.txt
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_fast_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r4a_tests_other.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for MODE in k1 k4bits k1split k3split k3splitmask k3splitplanes k3mask k3planes resadd resaddsplit resaddtorch k1wsplit; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4a_${MODE}_stats -o s -- python3 $R/tools/stage_only.py 20 $MODE > /dev/null 2>&1
done
for MODE in k1 k4bits; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/r4a_${MODE}_pmc_sq2 -o s -- python3 $R/tools/stage_only.py 6 $MODE > /dev/null 2>&1
done
cd $R
python - <<'PY' > gpurun_out/r4a_summary.txt
import csv, glob, collections
for mode in "k1 k4bits k1split k3split k3splitmask k3splitplanes k3mask k3planes resadd resaddsplit resaddtorch k1wsplit".split():
    fs = glob.glob(f'gpurun_out/r4a_{mode}_stats/**/*kernel_trace.csv', recursive=True)
    if not fs:
        print(mode, "no trace"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[r['Kernel_Name'][:90]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if sum(d) < 20: continue
        d = sorted(d)
        print(f"{mode:14s} {k:90s} n={len(d):3d} min {d[0]:7.1f} med {d[len(d)//2]:7.1f} avg {sum(d)/len(d):7.1f} max {d[-1]:7.1f}")
for mode in ("k1", "k4bits"):
    fs = glob.glob(f'gpurun_out/r4a_{mode}_pmc_sq2/**/*counter_collection.csv', recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in fs:
        for r in csv.DictReader(open(f)):
            if 'xty_f16x3' in r['Kernel_Name']:
                acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    c = {k: sum(v.values()) / len(v) for k, v in acc.items()}
    print(mode, c, "conflict/active = %.4f" % (c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
cat gpurun_out/r4a_tests_producer.txt gpurun_out/r4a_tests_other.txt gpurun_out/r4a_summary.txt
