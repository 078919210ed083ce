"""Reads the kernel trace of tools/sites_all.py: clusters of launches separated by > 2 ms; per site (after the warm-up cluster) cluster A = three
forward calls, cluster B = three forward + backward calls.  usage: sites_all_print.py <kernel_trace.csv> <stdout of sites_all.py>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
labels = [l.strip() for l in open(sys.argv[2]) if l.startswith("site ")]
clusters, cur = [], []
def is_mark(r): return any(k in r['Kernel_Name'].lower() for k in ("scan", "cumsum"))
for r in rows:
    if is_mark(r):
        clusters.append(cur); cur = []
    else: cur.append(r)
clusters.append(cur)
clusters = [c for c in clusters[1:] if c]        # (in front of the first marker: the first site's warm-up; empty stretches: consecutive markers)
# per site: [forward x 3] [forward + backward x 3] [the next site's warm-up, cut off by its own first marker]
per = []
k = 0
while k + 1 < len(clusters) and len(per) < len(labels):
    per.append((clusters[k], clusters[k + 1])); k += 3
# drop a leading cluster of library initialisation if the count is not 3 per site
STREAM = ("resadd", "xtx", "xty", "apply", "affine", "onepass", "split_kernel", "rows_", "stream", "elementwise", "vectorized")   # kernels that sweep the activation tensor
def span(c): return (int(c[-1]['End_Timestamp']) - int(c[0]['Start_Timestamp'])) / 1e3
def busy(c): return sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in c if any(s in r['Kernel_Name'] for s in STREAM))
print("%-46s %-8s | %9s %8s %9s | %9s %8s %9s" % ("site (three calls per figure, per call)", "route", "fwd us", "launches", "HBM idle", "fwd+bwd", "launches", "HBM idle"))
for k, lab in enumerate(labels):
    if k >= len(per): break
    a, b = per[k]
    route = "planes" if "planes" in lab else "fp32"
    name = lab.split(":", 1)[1].split("  route")[0].strip()
    print("%-46s %-8s | %9.1f %8.1f %9.1f | %9.1f %8.1f %9.1f" % (name, route, span(a) / 3, len(a) / 3, (span(a) - busy(a)) / 3, span(b) / 3, len(b) / 3, (span(b) - busy(b)) / 3))
print()
print("(fwd = producer + K1 tail + K2 + colouring + K3; the three calls of a cluster run back to back, so a call's share includes the launch gaps")
print(" between them; 'HBM idle' = the cluster's span minus the kernels that sweep the activation tensor: the small-matrix chain and every gap.)")
k = 6
if k < len(per):
    a = per[k][0]; n = len(a) // 3
    print("\nlaunches of ONE forward call of the headline site (cifar10 uncond final, 128 x 32 x 32 x 256):")
    t0 = int(a[2 * n]['Start_Timestamp'])
    for r in a[2 * n:]: print("  %-90s start %7.1f dur %6.1f" % (r['Kernel_Name'][:88], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    b = per[k][1]; n = len(b) // 3
    print("launches of ONE forward + backward call of the same site:")
    t0 = int(b[2 * n]['Start_Timestamp'])
    for r in b[2 * n:]: print("  %-90s start %7.1f dur %6.1f" % (r['Kernel_Name'][:88], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
for k, title in ((0, "the smallest site (cifar10 uncond block1.bn1, 128 x 4 x 4 x 256)"), (4, "cifar10 uncond block3.bn1 (128 x 16 x 16 x 256, planes)")):
    if k < len(per):
        b = per[k][1]; n = len(b) // 3
        print("launches of ONE forward + backward call of %s:" % title)
        t0 = int(b[2 * n]['Start_Timestamp'])
        for r in b[2 * n:]: print("  %-90s start %7.1f dur %6.1f" % (r['Kernel_Name'][:88], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
