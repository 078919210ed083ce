"""Runs ONE stage of the WC path n times at the headline site 128x32x32x256 on the SURVEY section 8d kernel-bench input: the target
of the round-3 rocprofv3 --kernel-trace / --pmc passes (tools/gpu_job_pmc_mode.sh <mode> <tag>; one kernel per run, so that what
the previous launch left in the 256-MiB memory-side cache is the same tensor every time, as in bench.py's timing loops).
usage: stage_only.py <n> <mode>;  mode = k3 | k3split | k3planes | k3mask | k1 | k1split | k4 | k4mask | k4bits | k6 | k6bits |
        k3splitmask | k3splitplanes | k1wsplit | resadd | resaddsplit | resaddtorch   (round 4: the producer and the planes route's epilogues)
        | resaddstats | resaddstatsk2 | resaddsplitk1k2   (round 5: the producer with K1's partials accumulated in its own pass)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import _lib
if os.environ.get("WC_LIB"):          # development: another build of the library (tools/build_var.py)
    _lib.LIB_PATH = os.environ["WC_LIB"]
from wc_gan_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mode = sys.argv[2] if len(sys.argv) > 2 else "k3"
N, H, C = 128, 32, 256
M = N * H * H
g = torch.Generator(device="cpu"); g.manual_seed(1234)
z = torch.randn(M, C, generator=g)
mix = torch.randn(C, C, generator=g) / C ** 0.5 + 0.3 * (torch.randn(C, 8, generator=g) @ torch.randn(8, C, generator=g)) / 8 ** 0.5
x = (z @ mix + 0.2).view(N, H, H, C).cuda()
gy = torch.randn(N, H, H, C, generator=g).cuda()
gamma = (torch.randn(1, C, C, generator=g) / C ** 0.5).cuda(); b = (0.1 * torch.randn(1, C, generator=g)).cuda()
y = torch.empty_like(x)
s, xtx = ops.stats(x.view(M, C))
mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, x.device, want_scale=True)
A, At, plan = ops.color(W, gamma, cs)
if mode in ("k3split", "k1split", "k3splitmask", "k3splitplanes", "k1wsplit", "k4xsplit", "k6xsplit"):
    xs = ops.split(x)
    A2, At2, plan2 = ops.color(W, gamma, xs.scale)
    be = ops.split_bias(A2, b, xs, mu)
if mode in ("k4mask", "k3mask", "k4bits", "k6bits", "k4xsplit", "k6xsplit"):
    _, mask = ops.apply(x, mu, A, b, None, plan=plan, relu=True, want_mask=True, out=y)
if mode in ("k3planes", "k3splitplanes"):
    rec = ops.out_scale(gamma, b, C, x.device)
if mode in ("k6", "k6bits", "k6xsplit"):
    R, gsum, scales = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True)
    _, _, S, gm = ops.bwd_factor(R, gsum, W, L, gamma, A, M, 1e-3, 1, True)
if mode.startswith("resadd"):
    hh = torch.randn(N, H, H, C, generator=g).cuda(); ss = torch.randn(N, H // 2, H // 2, C, generator=g).cuda()
run = {
    "k3splitmask": lambda: ops.apply_split(xs, None, A2, be, None, plan=plan2, out=y, relu=True, folded=True, want_mask=True),
    "k3splitplanes": lambda: ops.apply_split(xs, None, A2, be, None, plan=plan2, relu=True, folded=True, want_mask=True, oscale=rec),
    "k4xsplit": lambda: ops.bwd_reduce_xsplit(xs, mu, gy, None, 1, relu_mask=mask),
    "k6xsplit": lambda: ops.bwd_apply_xsplit(gy, xs, mu, At, S, gm, None, scales, relu_mask=mask),
    "k1wsplit": lambda: ops.whiten_split(xs, 1e-3, 0.99, 1, None, None),
    "resadd": lambda: ops.resadd(hh, ss, True),
    "resaddsplit": lambda: ops.resadd_split(hh, ss, True),
    "resaddstats": lambda: ops.resadd_stats_split(hh, ss, True, 1),                                 # round 5: the add's pass + K1's partials
    "resaddstatsk2": lambda: ops.whiten_presummed(ops.resadd_stats_split(hh, ss, True, 1), 1e-3, 0.99, 1, None, None),      # ... + K1 tail + K2
    "resaddsplitk1k2": lambda: ops.whiten_split(ops.resadd_split(hh, ss, True), 1e-3, 0.99, 1, None, None),                # round 4's chain
    "resaddtorch": lambda: (hh.view(N, H // 2, 2, H // 2, 2, C) + ss.view(N, H // 2, 1, H // 2, 1, C)),
    "k3": lambda: ops.apply(x, mu, A, b, None, out=y, plan=plan),
    "k3mask": lambda: ops.apply(x, mu, A, b, None, out=y, plan=plan, relu=True, want_mask=True),
    "k3split": lambda: ops.apply_split(xs, None, A2, be, None, plan=plan2, out=y, folded=True),
    "k3planes": lambda: ops.apply_planes(x, mu, A, b, None, plan, rec, relu=True, want_mask=True),
    "k1": lambda: ops.stats(x.view(M, C)),
    "k1split": lambda: ops.stats_split(xs),
    "k4": lambda: ops.bwd_reduce(x, mu, gy, None, 1),
    "k4mask": lambda: ops.bwd_reduce(x, mu, gy, None, 1, relu_mask=mask),
    "k4bits": lambda: ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=mask, write_masked=False),
    "k6": lambda: ops.bwd_apply(gy, x, mu, At, S, gm, None, scales=scales),
    "k6bits": lambda: ops.bwd_apply(gy, x, mu, At, S, gm, None, scales=scales, relu_mask=mask),
}[mode]
torch.cuda.synchronize()
for _ in range(n):
    run()
torch.cuda.synchronize()
