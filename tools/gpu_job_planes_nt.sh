#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3p_stats -o s -- python3 $R/tools/stage_only.py 20 k3planes > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3p_pmc_write -o s -- python3 $R/tools/stage_only.py 6 k3planes > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3p_pmc_fetch -o s -- python3 $R/tools/stage_only.py 6 k3planes > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3p_site_tl -o s -- python3 $R/tools/site_timeline.py > /dev/null 2>&1
cd $R
python tools/site_timeline_print.py gpurun_out/r3p_site_tl/s_kernel_trace.csv | grep "affine_ring\|total"
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3p_stats/**/*kernel_trace.csv', recursive=True)[0]
d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if 'affine_ring_kernel' in r['Kernel_Name'])
big = [v for v in d if v > 20]
print('planes pass', len(big), 'min %.1f med %.1f avg %.1f max %.1f' % (big[0], big[len(big)//2], sum(big)/len(big), big[-1]), '| gate', 'avg %.1f' % (sum(v for v in d if v <= 20) / max(1, len([v for v in d if v <= 20]))))
for c in ('write', 'fetch'):
    f = glob.glob('gpurun_out/r3p_pmc_%s/**/*counter_collection.csv' % c, recursive=True)[0]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'affine_ring_kernel' in r['Kernel_Name']]
    import collections
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'affine_ring_kernel' in r['Kernel_Name']: acc[r['Dispatch_Id']] += float(r['Counter_Value'])
    vals = sorted(acc.values())
    print(c, 'KB per dispatch (max)', vals[-1], 'MB', vals[-1] * 1024 / 1e6 * (2 if c == 'fetch' else 1))
PY
python tools/planes_check.py | grep "us"
