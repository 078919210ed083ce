"""Round 4: the signed error of K1's off-diagonal sums against float64 with the kappa compensation OFF (WC_K1_NO_BIAS_COMP=1), for the fp32-input
kernel (ops.stats) and the planes kernel (ops.stats_split), per input family -- run once per library variant (tools/build_var.py
wc_fast_xty alt=-DXTY_ALT=1, wc_split_xty alt=-DSXT_ALT=1): does the alternating sign of the MFMA chains remove the bias the constant was
fitted to?   usage: k1_alt_bias.py [lib.so]"""
import os, sys, numpy as np, torch
os.environ["WC_K1_NO_BIAS_COMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = sys.argv[1]
from oracle import wc_oracle as o
from wc_gan_amd import ops
def t(fn, it=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(it):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(400000); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
shape = (128, 32, 32, 256)
C = shape[-1]
for fam in ("ill", "uniform", "relu", "laplace"):
    rng = np.random.default_rng(11)
    x = (o.synth_activation(rng, shape, "ill") if fam == "ill" else o.synth_activation_family(rng, shape, fam)).astype(np.float32)
    X = torch.from_numpy(x.reshape(-1, C)).double()
    M = X.shape[0]
    _, sig_ref = o.moments_to_stats(X.sum(0).numpy(), (X.t() @ X).numpy(), M)
    sd = np.sqrt(np.diag(sig_ref)); iu = np.triu_indices(C, 1)
    xg = torch.from_numpy(x).cuda()
    xs = ops.split(xg)
    for name, fn in (("fp32-input K1", lambda: ops.stats(xg.view(-1, C))), ("planes K1", lambda: ops.stats_split(xs))):
        s, xtx = fn()[:2]
        _, sig = o.moments_to_stats(s.double().cpu().numpy(), xtx.double().cpu().numpy(), M)
        off = (sig - sig_ref) / np.outer(sd, sd)
        print("%-8s %-14s offdiag err / sqrt(sii sjj): mean %+.3e  std %.3e  max|.| %.3e   call %.1f us" %
              (fam, name, off[iu].mean(), off[iu].std(), np.abs(off[iu]).max(), t(fn)), flush=True)
