"""How stable is the off-diagonal bias of the fast K1 (the matrix pipe's fp32 accumulation is not correctly rounded: DESIGN.md
section 2)?  For a range of inputs: a = mean over i != j of (Sigma_gpu - Sigma_ref)_ij / sqrt(Sigma_ii Sigma_jj), its spread, and
the fit a + b rho.  Reference: float64 matmul on the GPU."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wc_oracle as o
from wc_gan_amd import ops
SPLIT = "--split" in sys.argv          # the K1 on pre-split planes (wc_stats_split_f16x2) instead of the fp32-input kernel
def case(name, x):
    M, C = x.shape
    xd = x.cuda()
    if SPLIT:
        if not ops.stats_split_supported(M, C):
            return
        s, xtx = ops.stats_split(ops.split(xd.view(M // 1024, 32, 32, C)))
    else:
        s, xtx = ops.stats(xd)
    X = xd.double()
    s_ref = X.sum(0); xtx_ref = X.t() @ X
    cov = lambda s_, x_: (x_ - torch.outer(s_, s_) / M) / (M - 1)
    sg, sr = cov(s, xtx), cov(s_ref, xtx_ref)
    sd = sr.diagonal().sqrt()
    E = (sg - sr) / torch.outer(sd, sd)
    rho = sr / torch.outer(sd, sd)
    iu = torch.triu_indices(C, C, 1)
    e, r = E[iu[0], iu[1]], rho[iu[0], iu[1]]
    A = torch.stack([torch.ones_like(r), r], 1)
    sol = torch.linalg.lstsq(A, e.unsqueeze(1)).solution.flatten()
    print("%-44s M %7d C %3d | diag mean %9.2e | offdiag mean %9.2e std %8.2e | fit %9.2e + %9.2e rho | mean|rho| %.2f"
          % (name, M, C, float(E.diagonal().mean()), float(e.mean()), float(e.std()), float(sol[0]), float(sol[1]), float(r.abs().mean())), flush=True)
g = torch.Generator(device="cpu"); g.manual_seed(7)
def ill(M, C, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(o.synth_activation(rng, (M // 1024, 32, 32, C), "ill").astype(np.float32).reshape(-1, C))
M, C = 131072, 256
for seed in (100, 101, 102):
    case("ill (seed %d)" % seed, ill(M, C, seed))
case("gaussian, independent", torch.randn(M, C, generator=g))
case("gaussian + mean 3", torch.randn(M, C, generator=g) + 3.0)
case("relu(gaussian) (half the values zero)", torch.relu(torch.randn(M, C, generator=g)))
case("relu(ill)", torch.relu(ill(M, C, 100)))
case("uniform(-1, 1)", torch.rand(M, C, generator=g) * 2 - 1)
case("laplace-like (heavy tails)", torch.randn(M, C, generator=g) * torch.randn(M, C, generator=g).abs())
case("gaussian, channel scales 1e-2 .. 1e2", torch.randn(M, C, generator=g) * torch.logspace(-2, 2, C))
case("ill x channel scales 1e-2 .. 1e2", ill(M, C, 100) * torch.logspace(-2, 2, C))
case("strongly correlated (rank 4 + 0.1 noise)", torch.randn(M, 4, generator=g) @ torch.randn(4, C, generator=g) + 0.1 * torch.randn(M, C, generator=g))
case("ill, M = 32768", ill(32768, C, 100))
case("ill, M = 65536", ill(65536, C, 100))
case("ill, M = 524288", ill(524288, C, 100))
case("ill, C = 128", ill(M, 128, 100))
case("gaussian, C = 128", torch.randn(M, 128, generator=g))
case("gaussian, C = 64, M = 262144", torch.randn(262144, 64, generator=g))
