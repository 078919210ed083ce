"""Where does the error of dx come from at a full-size ill-conditioned site?  Replace one stage's output at a time by
the GPU's in an otherwise float64 computation (development; behind DESIGN.md section 5)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wc_oracle as o
from wc_gan_amd import ops
shape = tuple(int(v) for v in sys.argv[1].split('x')) if len(sys.argv) > 1 else (128, 32, 32, 256)
C = shape[-1]
rng = np.random.default_rng(11)
x = o.synth_activation(rng, shape, "ill").astype(np.float32)
G, B = o.synth_coloring(rng, C, 1)
G = G.astype(np.float32); B = B.astype(np.float32)
slot = rng.integers(0, 1, shape[0])
gy = rng.standard_normal(shape).astype(np.float32)
X = x.reshape(-1, C).astype(np.float64); M = X.shape[0]
g = gy.reshape(-1, C).astype(np.float64)
eps = 1e-3
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())

def forward(sigma, mu):
    L, W = o.whitening_matrix(sigma, eps)
    return L, W, W.T @ G[0].astype(np.float64)

def backward(L, W, A, f, R, gsum):
    Gm = G[0].astype(np.float64)
    Wbar = Gm @ R.T
    Lbar = -np.tril(W.T @ Wbar @ W.T)
    P = np.tril(L.T @ Lbar); P[np.diag_indices(C)] *= 0.5
    Sb = W.T @ P @ W; Sb = 0.5 * (Sb + Sb.T)
    S = (2.0 * (1 - eps) / (M - 1)) * Sb
    fbar = g @ A.T + f @ S
    return fbar - fbar.mean(0, keepdims=True), S

s_ref = X.sum(0); xtx_ref = X.T @ X
mu_ref, sig_ref = o.moments_to_stats(s_ref, xtx_ref, M)
f_ref = X - mu_ref
L0, W0, A0 = forward(sig_ref, mu_ref)
R0 = f_ref.T @ g; gs0 = g.sum(0)
dx0, S0 = backward(L0, W0, A0, f_ref, R0, gs0)
y0 = f_ref @ A0 + B[0]

xt = torch.from_numpy(x).cuda()
s, xtx = ops.stats(xt.view(-1, C))
mu_g, sig_g = o.moments_to_stats(s.cpu().numpy(), xtx.cpu().numpy(), M)
# 1. GPU covariance only
L1, W1, A1 = forward(sig_g, mu_g)
f1 = X - mu_g
dx1, S1 = backward(L1, W1, A1, f1, f1.T @ g, gs0)
print(shape, "GPU covariance only:   y %.2e  dx %.2e   (A %.2e, S %.2e)" % (rel(f1 @ A1 + B[0], y0), rel(dx1, dx0), rel(A1, A0), rel(S1, S0)))
# 2. GPU K4 only (exact mu as float32)
mu32 = torch.tensor(mu_ref, dtype=torch.float32, device='cuda')
R, gsum = ops.bwd_reduce(xt, mu32, torch.from_numpy(gy).cuda(), None, 1)
Rg = R[0].cpu().numpy(); gsg = gsum[0].cpu().numpy()
f32mu = X - mu32.cpu().numpy().astype(np.float64)
dx2, S2 = backward(L0, W0, A0, f_ref, Rg, gsg)
print("GPU K4 (R, gsum) only:  R %.2e -> dx %.2e  (S %.2e)" % (rel(Rg, f32mu.T @ g), rel(dx2, dx0), rel(S2, S0)))
# 3. exact small stage, GPU K6 with float32 tables
At32 = torch.tensor(A0.T.copy(), dtype=torch.float32, device='cuda').view(1, C, C)
S32 = torch.tensor(S0, dtype=torch.float32, device='cuda')
gmean = torch.tensor((gs0 @ A0.T) / M, dtype=torch.float32, device='cuda')
dx3 = ops.bwd_apply(torch.from_numpy(gy).cuda(), xt, mu32, At32, S32, gmean, None).cpu().numpy().reshape(-1, C)
print("GPU K6 only (fp32 tables At, S, gmean): dx %.2e" % rel(dx3, dx0))
# 4. float32 rounding of mu alone
f4 = X - mu32.cpu().numpy().astype(np.float64)
dx4, _ = backward(L0, W0, A0, f4, f4.T @ g, gs0)
print("mu rounded to float32 only: dx %.2e" % rel(dx4, dx0))
