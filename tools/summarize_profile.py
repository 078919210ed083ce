"""Turns a rocprofv3 --kernel-trace --stats output directory into a small committed summary under profiles/.
usage: python tools/summarize_profile.py gpurun_out/<dir> profiles/<name>.md [t_start_s t_end_s]"""
import collections
import csv
import glob
import sys

src, dst = sys.argv[1], sys.argv[2]
window = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else None
after_gap = len(sys.argv) == 4 and sys.argv[3] == 'gap'        # keep what follows the last >= 0.3 s pause
trace = glob.glob(src + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
if window:
    rows = [r for r in rows if window[0] <= (int(r['Start_Timestamp']) - t0) / 1e9 < window[1]]
if after_gap:
    cut = 0
    for i in range(1, len(rows)):
        if int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp']) > 3e8:
            cut = i
    rows = rows[cut:]
    window = ('after the last 0.3 s pause', '')
d = collections.defaultdict(list)
for r in rows:
    d[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
with open(dst, 'w') as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary\n\nsource: `{trace}`" +
            ((f", window {window[0]}..{window[1]} s after first dispatch" if window and not after_gap else "") + (", steady-state steps only (after the last pause)" if after_gap else "")) +
            f"\n\n{len(rows)} dispatches, {tot/1e3:.2f} ms of kernel time" +
            "\n\n(The sum of kernel durations is NOT the wall time of the steps: the generator's forward of the update runs on a second stream beside the critic updates and" +
            " the forward-only passes put each block's shortcut convolution on a side stream, so durations overlap; and kernels run longer under the profiler than in the bench's" +
            " un-profiled graph replay -- compare with `ms_per_step` of the bench line, not with this total.)\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:45]:
        f.write(f"| `{k[:110]}` | {len(v)} | {sum(v)/1e3:.3f} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {100*sum(v)/tot:.1f} |\n")
print("wrote", dst)
