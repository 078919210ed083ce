"""Sync-WC on the planes route under TWO real ranks (the `st` + process_group path -- stats on planes / from the fused producer's
partials -> all_reduce -> factor; K4 on planes (flat) -> all_reduce -> K5 -> K6 -- shipped with a one-rank test only).  One GPU is all this box has,
so the two ranks share cuda:0 and meet over gloo (which all-reduces CUDA tensors through the host): the collectives are real, the ranks are separate
processes with their own halves of the batch.  Each rank's y and dx rows must be the rows a SINGLE process computes for the whole batch (sync-WC =
global-batch statistics), and the per-rank parameter gradients must add up to the global ones -- SURVEY section 8e, reference generator.py:24."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
rank, rdzv, out = int(sys.argv[1]), sys.argv[2], sys.argv[3]
dist.init_process_group("gloo", init_method="file://" + rdzv, rank=rank, world_size=2)
from wc_gan_amd.functional import residual_add, split_of, whiten_color
d = np.load(out + "/inputs.npz")
lo, hi = (0, 32) if rank == 0 else (32, 64)
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
C = d["h"].shape[-1]
h, s = dev(d["h"][lo:hi]).requires_grad_(True), dev(d["s"][lo:hi]).requires_grad_(True)
G, B = dev(d["G"]).requires_grad_(True), dev(d["B"]).requires_grad_(True)
mm, mc = torch.zeros(C, 1, device="cuda"), torch.eye(C, device="cuda")
x = residual_add(h, s, True, planes=True, x32=False, stat_groups=1)
st = split_of(x)
assert st is not None and st.moments is not None, "the fused producer did not run"
y = whiten_color(x, G, B, None, mm, mc, True, process_group=dist.group.WORLD, relu=False)
y.backward(dev(d["gy"][lo:hi]))
torch.cuda.synchronize()
np.savez(out + "/rank%%d.npz" %% rank, y=y.detach().cpu().numpy(), dh=h.grad.cpu().numpy(), ds=s.grad.cpu().numpy(), dG=G.grad.cpu().numpy(),
         dB=B.grad.cpu().numpy(), mc=mc.cpu().numpy(), mm=mm.cpu().numpy())
dist.barrier()
dist.destroy_process_group()
''' % ROOT


def test_sync_wc_on_the_planes_route_with_two_ranks(tmp_path):
    from oracle import wc_oracle as o
    from wc_gan_amd.functional import residual_add, whiten_color
    rng = np.random.default_rng(17)
    shape = (64, 32, 32, 256)
    N, H, W, C = shape
    x = o.synth_activation(rng, shape, "well").astype(np.float32)
    s = (0.5 * rng.standard_normal((N, H // 2, W // 2, C))).astype(np.float32)
    h = (x - np.repeat(np.repeat(s, 2, axis=1), 2, axis=2)).astype(np.float32)
    G, B = o.synth_coloring(rng, C, 1)
    gy = rng.standard_normal(shape).astype(np.float32)
    np.savez(tmp_path / "inputs.npz", h=h, s=s, G=G.astype(np.float32), B=B.astype(np.float32), gy=gy)
    env = dict(os.environ, WC_K2_TWO_LAUNCH="1")          # two processes time-slice one GPU: the K2 form without an in-launch wait
    procs = [subprocess.Popen([sys.executable, "-c", RANK, str(r), str(tmp_path / "rdzv"), str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o_[-1500:] for o_ in outs)
    # the single-process run of the WHOLE batch: per-replica statistics of 64 samples == sync-WC statistics of 2 x 32
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    ht, st_ = dev(h).requires_grad_(True), dev(s).requires_grad_(True)
    Gt, Bt = dev(G).requires_grad_(True), dev(B).requires_grad_(True)
    mm, mc = torch.zeros(C, 1, device="cuda"), torch.eye(C, device="cuda")
    xin = residual_add(ht, st_, True, planes=True, x32=False, stat_groups=1)
    y = whiten_color(xin, Gt, Bt, None, mm, mc, True, relu=False)
    y.backward(dev(gy))
    torch.cuda.synchronize()
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(np.asarray(b)).max(), 1e-30))
    cat = lambda k: np.concatenate([r0[k], r1[k]], axis=0)
    errs = dict(y=rel(cat("y"), y.detach().cpu().numpy()), dh=rel(cat("dh"), ht.grad.cpu().numpy()), ds=rel(cat("ds"), st_.grad.cpu().numpy()),
                dG=rel(r0["dG"] + r1["dG"], Gt.grad.cpu().numpy()), dB=rel(r0["dB"] + r1["dB"], Bt.grad.cpu().numpy()),
                mc0=rel(r0["mc"], mc.cpu().numpy()), mc1=rel(r1["mc"], mc.cpu().numpy()), mm0=rel(r0["mm"], mm.cpu().numpy()))
    print(errs)
    # (the planes' centre / scales are sampled per rank and the slabs are cut differently: rounding-level differences on a well-conditioned
    #  batch; no ReLU here, so that no mask flips stand between the two runs' gradients)
    assert all(v < 2e-5 for v in errs.values()), errs
