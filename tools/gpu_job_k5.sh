#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
python tools/k5_chain_graph.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/k5_tl -o s -- python3 $R/tools/k5_chain_graph.py > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/k5_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the graph replays: the last 2 x 10 x 8 kernels before the eager ones; take a window in the middle of the timed replay
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'bwd_tail_kernel' in n]
# calls 3 warm + 10 capture(not executed) + 10 replay + 10 replay + 10 eager: tails executed: 3 + 10 + 10 + 10 = 33
i1 = idx[15]; i0 = idx[14] + 1          # one call inside the second replay
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 1]:
    print('  %-60s start %6.1f dur %5.1f' % (r['Kernel_Name'][:58], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
print('  one call inside the graph: %.1f us' % ((int(rows[i1]['End_Timestamp']) - t0) / 1e3))
PY
