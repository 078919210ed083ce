#!/bin/bash
# round 5: the narrow-input weight gradient against MIOpen's: kernel trace of both, then the step with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for V in new miopen; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5g_$V -o s -- python3 $R/tools/narrow_wrw_bench.py $V > /dev/null 2>&1
done
cd $R
python - <<'PY' > gpurun_out/r5g_summary.txt
import csv, glob, collections
for mode in "new miopen".split():
    fs = glob.glob(f'gpurun_out/r5g_{mode}/**/*kernel_trace.csv', recursive=True)
    if not fs:
        print(mode, "no trace"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[r['Kernel_Name'][:110]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if len(d) < 15: continue
        d = sorted(d)
        print(f"{mode:6s} {k:110s} n={len(d):3d} min {d[0]:7.1f} med {d[len(d)//2]:7.1f} max {d[-1]:7.1f}")
PY
for V in 0 1; do WC_NARROW_WRW=$V timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('narrow_wrw=$V', d['value'], d['ms_per_step'], r.get('launch_us'), r.get('error'))"; done >> gpurun_out/r5g_summary.txt 2>&1
