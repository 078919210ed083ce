#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_job_pmc.sh tools/split_only.py r3_split 2>&1 | tail -3
for k in "apply_split_kernel" "affine_ring_kernel<256, false>" "xtx_split_kernel" "xty_f16x3_kernel<256, false"; do
python - "$k" <<'PY'
import csv, glob, sys
k = sys.argv[1]
f = glob.glob('gpurun_out/r3_split_stats/**/*kernel_trace.csv', recursive=True)[0]
d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if k in r['Kernel_Name'])
print(k, len(d), 'min %.1f med %.1f avg %.1f max %.1f' % (d[0], d[len(d)//2], sum(d)/len(d), d[-1]))
PY
done
