import csv, sys
# usage: site_timeline_print.py <kernel_trace.csv> [marker kernel of a call's first launch: default subsample_mean (rounds 2-3); round 4: resadd_sample]
marker = sys.argv[2] if len(sys.argv) > 2 else 'subsample_mean'
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if marker in n]
def show(i0, i1, title):
    t0 = int(rows[i0]['Start_Timestamp']); print(title)
    for r in rows[i0:i1]:
        print('  %-70s start %7.1f dur %6.1f' % (r['Kernel_Name'][:68], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    print('  total %.1f us, %d launches' % ((int(rows[i1]['Start_Timestamp']) - t0) / 1e3, i1 - i0))
show(idx[4], idx[5], ('producer + ' if marker != 'subsample_mean' else '') + 'forward + backward, 128x32x32x256, ReLU epilogue (1-bit mask)')
show(idx[10], idx[11], ('producer + ' if marker != 'subsample_mean' else '') + 'grouped forward (5 groups), 320x32x32x256')
if len(idx) >= 18:
    show(idx[16], idx[17], 'forward + backward with the K3 -> convolution hand-off (planes out, gated second launch), 128x32x32x256')
k3 = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'affine_ring_kernel<256, false, true, false>' in r['Kernel_Name']]
if len(k3) >= 26:
    flow, loop = k3[:6], k3[-20:]
    print('K3 (ReLU + bit mask) under this profiler: in the layer\'s flow %.1f us (6 launches: %s); in a loop of its own %.1f us (20 launches, min %.1f max %.1f)'
          % (sum(flow) / len(flow), ' '.join('%.1f' % v for v in flow), sum(loop) / len(loop), min(loop), max(loop)))
