#!/usr/bin/env python3
"""bench.py -- images/sec of the G+D training step, ResNet-SN + WC, batch 64 per GPU (default: CIFAR-10 uncond).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the environment
this process only LAUNCHES the job -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...`
as a child, before anything here touches the GPU -- and exits with the child's code; when the driver starts the ranks
itself (WORLD_SIZE set) each rank checks WORLD_SIZE == --gpus.  One rank per GPU over RCCL; rank 0 prints ONE JSON line.

  step      one "G+D step" = training_ratio critic updates (64 real + 64 generated, generator forward in
            train mode) + one generator update at batch 64 x 2 (run.py:101,293-294); images/sec = 64 x steps/s
            per GPU (weak scaling: every GPU keeps batch 64).
  roofline  the dominant hand-written kernel, the fused WC apply (K3) at the headline site
            128 x 32 x 32 x 256: algorithmic bytes 2*M*C*4 + (C*C + C)*4 per launch over the launch time
            measured here with HIP events on the launching stream, against the 8 TB/s HBM3E peak (and against the
            same-run stream copy).
  cpu_baseline  the reference's UNFUSED fp32 op order (SURVEY.md rows a2 + a6: transpose -> mean -> f f^T/(M-1) ->
            shrink -> cholesky -> triangular solve vs I -> W f -> transpose back -> 1x1 conv + bias; autograd for the
            backward) as the torch-CPU sequence of BASELINE.md section 3, timed on this box's host cores over the WC
            sites of one G+D step; the float64 oracle only CHECKS its output.

--config picks one of BASELINE.json's GPU configurations (wc_gan_amd.train.CONFIGS); the default is the headline one.
--dry-run is the CPU rehearsal of the N-rank path (gloo, a tiny model without HIP layers): launch, rendezvous, the
flat gradient buckets and the place of every all-reduce -- what tests/test_bench_launch.py checks with two ranks.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
TRAINING_RATIO = 5             # default of the missing gan.cmd parser [UPSTREAM-RECALL]; ratio 1 also reported
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md); the power-limited sustained rate measured here is 1430
CONFIG_NAMES = ("cifar10_uncond", "cifar10_cond", "stl10_uncond", "tinyimagenet_cond_sa")
WORKLOADS = {
    "cifar10_uncond": "CIFAR-10 ResNet SN uncond + WC (scripts/cifar10_resnet_sn_uncond.sh)",
    "cifar10_cond": "CIFAR-10 ResNet SN conditional, class-conditional coloring (scripts/cifar10_resnet_sn_cond.sh)",
    "stl10_uncond": "STL-10 ResNet SN uncond + WC, 48x48 (scripts/stl10_resnet_sn_uncond.sh)",
    "tinyimagenet_cond_sa": "Tiny-ImageNet ResNet SN cond-SA, 64x64, 200 classes (scripts/tinyimagenet_resnet_sn_cond_sa.sh)",
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=CONFIG_NAMES, default="cifar10_uncond",
                    help="which BASELINE.json configuration to step (default: the headline one)")
    ap.add_argument("--training-ratio", type=int, default=TRAINING_RATIO)
    ap.add_argument("--sync-wc", action="store_true", help="all-reduce WC statistics across replicas")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-port", action="store_true", help="also time the C-ABI CPU restatement (oracle/wc_cpu.cpp; ~45 s more)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole G+D step as one hipGraph.  The default on one GPU: with ~2000 launches per step the "
                         "eager loop is at the edge of host-bound; the eager step is reported next to it")
    ap.add_argument("--segments", action="store_true",
                    help="replay the step as a chain of hipGraphs cut at the gradient all-reduces, which stay plain RCCL calls "
                         "(the default with several GPUs, unless --sync-wc puts collectives inside the WC layers); if any "
                         "rank fails to record its chain, ALL ranks run eagerly")
    ap.add_argument("--eager", action="store_true", help="launch kernel by kernel from Python")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the N-rank launch path: gloo backend, a tiny model without HIP layers")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """--gpus N > 1 without a launcher around us: start N fresh rank processes.  Nothing in THIS process has touched
    the GPU (no torch.cuda call so far), and it never execs: it waits for the child and returns its exit code."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def time_kernel(fn, iters=20, warm=3, graph=False):
    """Average duration of one fn() on the GPU between two HIP events on the launching stream.
    graph=False: `iters` calls issued from Python (the launches are separated by whatever the host needs per call -- the form
    rounds 1 and 2 used, and the one whose per-kernel average the rocprofv3 kernel trace of the same loop reproduces).
    graph=True: the same calls recorded into ONE hipGraph and replayed -- launches truly back to back.  The two differ for
    the streaming kernels (round 3): a K3 that starts the moment the previous K3 retires runs beside
    that launch's write-back (134 MB of dirty lines still on their way to HBM) and takes 55-61 us instead of 47-50."""
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = None
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()                        # once untimed: the first replay uploads the graph
        torch.cuda.synchronize()
    ts = []
    for _ in range(3 if g is not None else 1):      # (a graph replay is timed three times and the MEDIAN kept: one replay in some dozen
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)     # reads milliseconds -- a stall of the box, not of the kernels)
        e0.record()                       # torch's current stream == the stream the C ABI launches on
        if g is not None:
            g.replay()
        else:
            for _ in range(iters):
                fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e-3)
    return sorted(ts)[len(ts) // 2]


_SPIN = {}


def _spin_cycles(us=200.0):
    """torch.cuda._sleep cycles that keep the GPU busy (no memory traffic) for about `us` microseconds: calibrated once with events."""
    import torch
    if "per_us" not in _SPIN:
        torch.cuda._sleep(1000); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(2_000_000); e1.record(); torch.cuda.synchronize()
        _SPIN["per_us"] = 2_000_000 / max(e0.elapsed_time(e1) * 1e3, 1.0)
    return max(int(_SPIN["per_us"] * us), 1000)


def time_isolated(fn, iters=20, warm=3):
    """Median duration of ONE fn() between two HIP events on the launching stream, with nothing else touching memory around it: each
    launch is queued behind a ~200 us register-only spin (torch.cuda._sleep), so the host has finished enqueueing event, kernel and event
    before the GPU reaches them (no host latency inside the bracket -- with an idle GPU in front the ctypes call itself, 5-10 us for a
    20-argument entry, lands between the events) and the kernel runs beside no predecessor's write-back.  That is the kernel by itself,
    what rocprofv3's per-dispatch durations of a spaced loop give.  A plain loop of raw C-ABI calls is NOT that: the host queues 20 launches
    within a few hundred microseconds and every launch but the first runs beside its predecessor's write-back (round 4: the masked K3
    variants read 60-67 us that way against 49-53 in the kernel trace) -- that figure is reported separately as back_to_back_us."""
    import torch
    for _ in range(warm):
        fn()
    spin = _spin_cycles()
    evs = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(spin)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
        if len(evs) % 5 == 0:
            torch.cuda.synchronize()                                  # bounded queue depth; the spin in front of the next launch restores the lead
    torch.cuda.synchronize()
    ts = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
    return ts[len(ts) // 2] * 1e-3                                   # the median: one launch in 20 met a multi-millisecond pause of the box in round 4


def roofline_apply(dev):
    """roofline.kernel = the K3 kernel THE LAYERS launch at the headline site: the site is built as the generator
    builds Generator.BN.Final (create_norm('d', 'uconv'), generator.py:154), fed as the generator feeds it -- by the residual add of
    the last block (functional.residual_add: pre-split planes since round 4) -- run once with ops.TRACE on, and the K3 entry point
    that call made is the one timed.  Every other K3 variant is listed under k3_kernels with the same timing rule; none is picked."""
    import torch
    from wc_gan_amd import ops
    from wc_gan_amd import functional as WF
    from wc_gan_amd.generator import create_norm
    N, H, C = 128, 32, 256
    M = N * H * H
    g = torch.Generator(device="cpu"); g.manual_seed(1234)
    # SURVEY section 8d kernel-bench input: x = z Mix + 0.2 (cond ~1e6), Gamma = randn/sqrt(C), beta = 0.1 randn
    z = torch.randn(M, C, generator=g)
    mix = torch.randn(C, C, generator=g) / C ** 0.5 + 0.3 * (torch.randn(C, 8, generator=g) @ torch.randn(8, C, generator=g)) / 8 ** 0.5
    x = (z @ mix + 0.2).view(N, H, H, C).to(dev)
    gamma = (torch.randn(1, C, C, generator=g) / C ** 0.5).to(dev)
    b = (0.1 * torch.randn(1, C, generator=g)).to(dev)
    xb = M * C * 4
    alg_bytes = 2 * xb + (C * C + C) * 4

    # ---- the site as the generator runs it: the last block's residual add (h + upsampled shortcut = x), then the norm stack + ReLU
    s_half = (0.05 * torch.randn(N, H // 2, H // 2, C, generator=g)).to(dev)
    hh = (x.view(N, H // 2, 2, H // 2, 2, C) - s_half.view(N, H // 2, 1, H // 2, 1, C)).reshape(N, H, H, C).contiguous()
    site = create_norm('d', 'uconv')(axis=-1, name='Generator.BN.Final', channels=C).to(dev)
    with torch.no_grad():
        site.branches[0].kernel.copy_(gamma.view(1, 1, C, C)); site.branches[0].bias.copy_(b.view(C))
        sg = site.wants_moments(x.shape) if site.takes_split(x.shape) else 0       # (as ResBlockUp.forward asks its readers)
        xin = WF.residual_add(hh, s_half, True, planes=site.takes_split(x.shape), stat_groups=sg)
        on_planes = WF.split_of(xin) is not None
        fused_k1 = on_planes and WF.split_of(xin).moments is not None
        ops.TRACE = []
        try:
            site(xin, None, relu=True)
            traced = list(ops.TRACE)
        finally:
            ops.TRACE = None
    if len(traced) != 1:
        raise RuntimeError(f"expected one K3 launch from the site, traced {[t[:2] for t in traced]}")
    entry, kernel, k3_layers, _keep_layers = traced[0]       # k3_layers: the raw foreign call of that launch, repeated (no Python wrapper in the loop)
    masked = ", true, " in kernel.split("<")[1] if "apply_split" in kernel or "affine_ring" in kernel else False
    alg_layers = alg_bytes + (xb // 32 if masked else 0)          # + the one-bit ReLU mask the epilogue writes

    # ---- the other K3 variants on the same input (listed, never picked)
    s, xtx = ops.stats(x.view(M, C))
    mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, dev, want_scale=True)
    A, At, plan = ops.color(W, gamma, cs)
    y = torch.empty_like(x)
    xs = ops.split(x)
    A2, At2, plan2 = ops.color(W, gamma, xs.scale)
    be = ops.split_bias(A2, b, xs, mu)
    ws_split = ops.apply_split_workspace(C, 1, dev)
    mask_buf = torch.empty(M // 32, C, dtype=torch.int32, device=dev)
    planes_buf = torch.empty(2, N, H, H, C, dtype=torch.float16, device=dev)
    orec = ops.out_scale(gamma, b, C, dev)
    variants = {
        "affine_ring_kernel<256, false, false, false> (wc_apply_f32: fp32 in, fp32 out)":
            (lambda: ops.apply(x, mu, A, b, None, out=y, plan=plan), alg_bytes),
        "affine_ring_kernel<256, false, true, false> (wc_apply_mask_f32: fp32 in, ReLU + bit mask)":
            (lambda: ops.apply(x, mu, A, b, None, out=y, plan=plan, relu=True, want_mask=True, _mask_out=mask_buf), alg_bytes + xb // 32),
        "affine_ring_kernel<256, false, true, true> (wc_apply_planes_f32: fp32 in, ReLU + bit mask + the next convolution's planes; two launches)":
            (lambda: ops.apply_planes(x, mu, A, b, None, plan, orec, relu=True, want_mask=True, _mask_out=mask_buf, _planes_out=planes_buf), alg_bytes + xb // 32),
        "apply_split_kernel<256, false, false, false> (wc_apply_split_f16x2: planes in, fp32 out)":
            (lambda: ops.apply_split(xs, None, A2, be, None, plan=plan2, out=y, folded=True, ws=ws_split), alg_bytes),
        "apply_split_kernel<256, false, true, false> (wc_apply_split_ex_f16x2: planes in, ReLU + bit mask)":
            (lambda: ops.apply_split(xs, None, A2, be, None, plan=plan2, out=y, relu=True, folded=True, ws=ws_split, want_mask=True, _mask_out=mask_buf), alg_bytes + xb // 32),
        "apply_split_kernel<256, false, true, true> (wc_apply_split_ex_f16x2: planes in, ReLU + bit mask + the next convolution's planes; two launches)":
            (lambda: ops.apply_split(xs, None, A2, be, None, plan=plan2, relu=True, folded=True, ws=ws_split, want_mask=True, oscale=orec,
                                     _mask_out=mask_buf, _planes_out=planes_buf), alg_bytes + xb // 32),
    }
    y2 = torch.empty_like(x)
    t_copy = time_isolated(lambda: ops.stream_copy(x, y2))           # the yardstick, timed by the same rule as the kernels it is compared with
    copy_gbs = 2 * xb / t_copy / 1e9
    copy_loop_gbs = 2 * xb / time_kernel(lambda: ops.stream_copy(x, y2)) / 1e9      # rounds 1-3's yardstick: a plain loop, launches overlapping head to tail
    k3 = {}
    keep_alive = []
    for name, (fn, nbytes) in variants.items():
        ops.TRACE = []          # every variant is timed as the layers' kernel is: its raw foreign call, taken from one traced wrapper call
        try:
            fn()
            fn, keep = ops.TRACE[0][2], ops.TRACE[0][3]
            keep_alive.append(keep)
        finally:
            ops.TRACE = None
        t = time_isolated(fn)
        k3[name] = {"launch_us": round(t * 1e6, 2), "algorithmic_bytes": nbytes, "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4),
                    "frac_of_isolated_copy": round(nbytes / t / 1e9 / copy_gbs, 4), "frac_of_loop_copy": round(nbytes / t / 1e9 / copy_loop_gbs, 4),
                    "run_by_the_layers_at_this_site": kernel in name}

    # ---- the layers' kernel: isolated, back to back in a graph, and in the site's own flow
    t_layers = time_isolated(k3_layers)
    t_b2b = time_kernel(k3_layers, graph=True)
    def front():      # what stands in front of K3 at the site: K1 + K2 + color (+ the planes' bias fold), the layer's own calls
        if on_planes:
            st = WF.split_of(xin)
            mu_, _, W_ = (ops.whiten_presummed if fused_k1 else ops.whiten_split)(st, 1e-3, 0.99, 1, None, None)
            ops.color_split(W_, gamma, st, mu_, b)
        else:
            mu_, _, W_, cs_ = ops.whiten(x.view(M, C), 1e-3, 0.99, 1, None, None)
            ops.color(W_, gamma, cs_)
    ev = []
    for _ in range(23):
        front()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); k3_layers(); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(bb) for a, bb in ev[3:]]
    t_flow = sum(ts) / len(ts) * 1e-3
    achieved = alg_layers / t_layers / 1e9

    # HBM bytes per launch: NOT measured in this run -- read from the committed PMC passes of the same kernel
    # (FETCH_SIZE x2 + WRITE_SIZE, collected as MI355X_MICROARCH.md's HBM section prescribes; see the file)
    traffic = src = None
    cands = (("r5_apply_k3splitmask_pmc.json", "r4_apply_k3splitmask_pmc.json", "r3_apply_k3split_pmc.json") if "apply_split" in kernel else
             ("r4_apply_k3mask_pmc.json", "r3_apply_k3_pmc.json", "r2_apply_k3_pmc.json"))
    for name in cands:
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            traffic, src = json.load(open(pmc)).get("traffic_bytes_per_launch"), "profiles/" + name
            break

    # ---- the whole forward site through the layer object (K1 -> K2 -> color -> K3, the layers' own route), and the producer
    # (hipGraph replays: through the layer objects the Python loop, not the GPU, would set the pace -- ~300 us of host time per call)
    with torch.no_grad():
        t_site = time_kernel(lambda: site(xin, None, relu=True), iters=10, graph=True)
        t_prod = time_kernel(lambda: WF.residual_add(hh, s_half, True, planes=on_planes, stat_groups=sg), iters=10, graph=True)
        t_prod_r4 = time_kernel(lambda: WF.residual_add(hh, s_half, True, planes=on_planes), iters=10, graph=True)
        xin_r4 = WF.residual_add(hh, s_half, True, planes=on_planes)
        t_site_r4 = time_kernel(lambda: site(xin_r4, None, relu=True), iters=10, graph=True)
        t_prod32 = time_kernel(lambda: ops.resadd(hh, s_half, True), iters=10, graph=True)
        t_torch_add = time_kernel(lambda: hh.view(N, H // 2, 2, H // 2, 2, C) + s_half.view(N, H // 2, 1, H // 2, 1, C), iters=10, graph=True)
        x32 = ops.resadd(hh, s_half, True)
        t_site32 = time_kernel(lambda: site(x32, None, relu=True), iters=10, graph=True)
    # every stage of the site on its own (HIP events, same inputs): the algorithmic bytes of SURVEY section 8d per stage
    gy = torch.randn(N, H, H, C, generator=g).to(dev)
    y_relu, relu_bits = ops.apply(x, mu, A, b, None, plan=plan, relu=True, want_mask=True)
    from wc_gan_amd import conv as fconv
    R, gsum, scales = ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True)
    dg, db, S, gm = ops.bwd_factor(R, gsum, W, L, gamma, A, M, 1e-3, 1, True)
    stage = lambda fn, nbytes: (lambda tt: {"us": round(tt * 1e6, 1), "frac_of_peak": round(nbytes / tt / 1e9 / HBM_PEAK_GBS, 3)})(time_kernel(fn, iters=10))
    only_us = lambda fn: {"us": round(time_kernel(fn, iters=10) * 1e6, 1)}
    stages = {
        "producer: residual add -> pre-split planes + K1's partials (wc_resadd_stats_split_f32: sample + one pass + gate; what the generator runs in front of this site)": stage(lambda: ops.resadd_stats_split(hh, s_half, True, 1), int(2.25 * xb)),
        "producer: residual add -> pre-split planes (wc_resadd_split_f32: sample + one pass + gate; round 4's route)": stage(lambda: ops.resadd_split(hh, s_half, True), int(2.25 * xb)),
        "K1 tail + K2 wc_whiten_presummed_f16x2 (what the layers run at this site: no pass over the tensor)": only_us(lambda: ops.whiten_presummed(WF.split_of(xin), 1e-3, 0.99, 1, None, None)) if fused_k1 else None,
        "producer: residual add -> fp32 (wc_resadd_f32)": stage(lambda: ops.resadd(hh, s_half, True), int(2.25 * xb)),
        "K1+K2 wc_whiten_split_f16x2 (planes, K1 as a pass of its own: round 4's route)": only_us(lambda: ops.whiten_split(xs, 1e-3, 0.99, 1, None, None)),
        "K1+K2 wc_whiten_f32 (fp32 input)": only_us(lambda: ops.whiten(x.view(M, C), 1e-3, 0.99, 1, None, None)),
        "K1 wc_stats_f32": stage(lambda: ops.stats(x.view(M, C)), xb),
        "K1 wc_stats_split_f16x2 (planes)": stage(lambda: ops.stats_split(xs), xb),
        "K2 wc_factor_f64": only_us(lambda: ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, dev)),
        "color wc_color_f32": only_us(lambda: ops.color(W, gamma, cs)),
        "color wc_color_split_f32 (planes route: tables for the planes' scales + the additive term, same launch count)": only_us(lambda: ops.color_split(W, gamma, xs, mu, b)),
        "  replaces: wc_conv_split_f32 of y (absmax + split, two launches) where K3 writes the next convolution's planes": only_us(lambda: fconv.split_planes(y_relu)),
        "K4 wc_bwd_reduce_f32": stage(lambda: ops.bwd_reduce(x, mu, gy, None, 1), 2 * xb),
        "K4 wc_bwd_reduce_bits_f32 (bit mask in, no masked copy out: as the generator runs it)":
            stage(lambda: ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=relu_bits, write_masked=False), 2 * xb + xb // 32),
        "K6 wc_bwd_apply_bits_f32 (applies the same bits to gy: as the generator runs it)":
            stage(lambda: ops.bwd_apply(gy, x, mu, At, S, gm, None, scales=scales, relu_mask=relu_bits), 3 * xb + xb // 32),
        "K4 wc_bwd_reduce_mask_f32 (writes the masked copy: shapes without the bits route)": stage(lambda: ops.bwd_reduce(x, mu, gy, None, 1, want_scales=True, relu_mask=relu_bits), 3 * xb),
        "K4 wc_bwd_reduce_xsplit_f32 (x from the producer's planes, bit mask in: as the generator runs the sites fed by a residual add)":
            stage(lambda: ops.bwd_reduce_xsplit(xs, mu, gy, None, 1, relu_mask=relu_bits), 2 * xb + xb // 32),
        "K6 wc_bwd_apply_xsplit_f32 (x from the planes, the same bits applied to gy)":
            stage(lambda: ops.bwd_apply_xsplit(gy, xs, mu, At, S, gm, None, scales, relu_mask=relu_bits), 3 * xb + xb // 32),
        "K5 wc_bwd_factor_f64": only_us(lambda: ops.bwd_factor(R, gsum, W, L, gamma, A, M, 1e-3, 1, True)),
        "K6 wc_bwd_apply_f32": stage(lambda: ops.bwd_apply(gy, x, mu, At, S, gm, None, scales=scales), 3 * xb),
    }
    stages = {k: v for k, v in stages.items() if v is not None}
    return {"bound": "hbm", "kernel": kernel + ", 128x32x32x256", "kernel_match": kernel, "entry_point": entry,
            "kernel_choice": "the K3 launch traced (ops.TRACE) from the layer call WhiteningColoring('Generator.BN.Final')(x, relu=True), x handed over by "
                             "functional.residual_add as the generator hands it over; k3_kernels lists every variant, none is picked",
            "site_input": ("pre-split planes + K1's partials from the residual add's own pass (wc_resadd_stats_split_f32)" if fused_k1 else
                           "pre-split planes (wc_resadd_split_f32)") if on_planes else "fp32",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_from_committed_profile": src,
            "launch_us": round(t_layers * 1e6, 2), "algorithmic_bytes": alg_layers,
            "timing": "launch_us: median of 20 single launches of the raw C-ABI call, each between two HIP events on the launching stream, queued behind a register-only spin so that no host latency and no predecessor's write-back falls inside the bracket (the same rule for every entry of k3_kernels and for the stream copy); "
                      "back_to_back_us: the same 20 launches replayed as one hipGraph (median of 3 replays); in_flow_us: events around single launches behind the site's own K1 -> K2 -> color",
            # Two yardsticks, each under ONE name for good (round 4 moved the denominator under an unchanged field name):
            #   isolated copy: the hand-written read+write stream kernel timed by the rule launch_us is timed by (one launch at a time behind a spin)
            #   loop copy:     the same kernel in a plain loop, launches overlapping head to tail -- rounds 1-3's `frac_of_stream_copy`
            "isolated_copy_GBs": round(copy_gbs, 1),
            "loop_copy_GBs": round(copy_loop_gbs, 1),
            "frac_of_isolated_copy": round(achieved / copy_gbs, 4),
            "frac_of_loop_copy": round(achieved / copy_loop_gbs, 4),
            "back_to_back_us": round(t_b2b * 1e6, 2), "in_flow_us": round(t_flow * 1e6, 2),
            "in_flow_frac_of_isolated_copy": round(alg_layers / t_flow / 1e9 / copy_gbs, 4),
            "k3_kernels": k3, "site_stages": stages,
            "forward_site_us": round(t_site * 1e6, 1),
            "forward_site_route": "layer object (hipGraph replay of 10 calls), training mode, input " + ("on planes" if on_planes else "fp32") + " (producer not included: it replaces the block's residual add, timed below)",
            # the site alone moves 2 M C 4 bytes since the producer carries K1 (one read of x on planes + the write of y, counted
            # as fp32); rounds 1-4 divided 3 M C 4 by a site that made all three passes.  With the producer: its two reads and one write of the
            # sum (the shortcut at a quarter of the rows: 2.25 M C 4) + the site's 2 M C 4 = 4.25 M C 4.
            "forward_site_frac_of_peak": round((2 if fused_k1 else 3) * xb / t_site / 1e9 / HBM_PEAK_GBS, 4),
            "forward_site_bytes_counted": ("2 M C 4 (K1 rides on the producer's pass)" if fused_k1 else "3 M C 4"),
            "forward_site_plus_producer_frac_of_peak": round(4.25 * xb / (t_site + t_prod) / 1e9 / HBM_PEAK_GBS, 4),
            "forward_site_fp32_input_us": round(t_site32 * 1e6, 1),
            "producer_us": {"residual add as the layers run it": round(t_prod * 1e6, 1), "residual add -> planes only (round 4)": round(t_prod_r4 * 1e6, 1),
                            "residual add -> fp32 (HIP)": round(t_prod32 * 1e6, 1), "torch broadcast add (rounds 1-3)": round(t_torch_add * 1e6, 1)},
            "forward_site_round4_route_us": round(t_site_r4 * 1e6, 1),
            "forward_site_plus_producer_us": {"planes": round((t_site + t_prod) * 1e6, 1), "planes, K1 as a pass of its own (round 4)": round((t_site_r4 + t_prod_r4) * 1e6, 1),
                                              "fp32 (HIP add)": round((t_site32 + t_prod32) * 1e6, 1),
                                              "fp32 (torch add, rounds 1-3)": round((t_site32 + t_torch_add) * 1e6, 1)}}


def safe(fn, what):
    """A secondary measurement must not cost the run its result line (the first 8-GPU run cannot come back empty
    because rank 0's roofline threw): -> fn()'s result, or {"error": ...} in its place."""
    try:
        return fn()
    except Exception as exc:       # noqa: BLE001 -- anything: the line is printed regardless
        print(f"[bench] {what} failed ({type(exc).__name__}: {str(exc)[:300]}); the line is printed without it", file=sys.stderr, flush=True)
        return {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}


def conv_roofline(dev):
    """The kernel the step spends most time in after the WC path: the 3x3 256->256 block convolution at 128x32x32
    (conv_f16x3_kernel<4,4>), MFMA-bound: 3 fp16 MFMA products per fp32 product."""
    import torch
    from wc_gan_amd import conv as C
    N, H, Cc = 128, 32, 256
    g = torch.Generator(device="cpu"); g.manual_seed(1234)
    x = torch.randn(N, H, H, Cc, generator=g).to(dev)
    w = (torch.randn(Cc, Cc, 3, 3, generator=g) / (9 * Cc) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    (gf, kf, nf), _ = C._geoms('same', N, H, H, w)
    planes, img = C.split_planes(x), C.weight_image(w, gf, kf, nf)
    t = time_kernel(lambda: C.run(planes, img, gf))
    flop32 = 2.0 * N * H * H * 9 * Cc * Cc
    return {"bound": "mfma", "kernel": "conv_f16x3_kernel<4,4> (3x3 'same' 256->256 at 128x32x32, split-fp16 operands)",
            "achieved": round(3 * flop32 / t / 1e12, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s (fp16 MFMA flops issued)",
            "frac": round(3 * flop32 / t / 1e12 / MFMA_F16_PEAK_TFLOPS, 4), "launch_us": round(t * 1e6, 1),
            "fp32_equivalent_TFLOPs": round(flop32 / t / 1e12, 1), "frac_of_sustained_1430": round(3 * flop32 / t / 1e12 / 1430.0, 4)}


def site_list(config_name):
    from wc_gan_amd.train import CONFIGS, wc_sites
    return [(h, w, c) for _, _, h, w, c in wc_sites(CONFIGS[config_name], 64)]


def wc_sites_gpu(dev, ratio, sites):
    """GPU time of the WC sites of one G+D step (what cpu_baseline times on the host), unconditional coloring."""
    import torch
    from wc_gan_amd.functional import whiten_color, whiten_color_grouped
    g = torch.Generator(device="cpu"); g.manual_seed(1234)
    work = []
    for H, W, C in sites:
        G = (torch.randn(1, C, C, generator=g) / C ** 0.5).to(dev).requires_grad_(True)
        B = torch.zeros(1, C, device=dev, requires_grad=True)
        xd = torch.randn(64 * ratio, H, W, C, generator=g).to(dev)      # the critic-phase passes, stacked (train.generate)
        x128 = torch.randn(128, H, W, C, generator=g).to(dev).requires_grad_(True)
        work.append((xd, x128, G, B, torch.randn(128, H, W, C, generator=g).to(dev)))

    def one():
        for xd, x128, G, B, gy in work:
            with torch.no_grad():
                whiten_color_grouped(xd, ratio, G, B)        # `ratio` forward passes with per-pass statistics
            whiten_color(x128, G, B).backward(gy)
    return time_kernel(one, iters=5, warm=2)


# ---------------------------------------------------------------------------------------------------------------------
# cpu_baseline: the reference's unfused fp32 op order on the host cores
# ---------------------------------------------------------------------------------------------------------------------
def cpu_unfused_site(x, kernel, bias, eps=1e-3):
    """DecorelationNormalization.call + Conv2D 1x1 as the TF graph runs them, op by op, fp32:
    transpose to (C, M) -> mean -> centre -> f f^T/(M-1) -> (1-eps) Sigma + eps I -> cholesky -> triangular solve
    against I -> W f -> reshape/transpose back to NHWC -> 1x1 convolution + bias."""
    import torch
    N, H, W_, C = x.shape
    xt = x.permute(3, 0, 1, 2).reshape(C, -1)
    M = xt.shape[1]
    mu = xt.mean(dim=1, keepdim=True)
    f = xt - mu
    sigma = (f @ f.t()) / (M - 1)
    eye = torch.eye(C, dtype=x.dtype)
    T = (1.0 - eps) * sigma + eps * eye
    L = torch.linalg.cholesky(T)
    Wm = torch.linalg.solve_triangular(L, eye, upper=False)
    xh = (Wm @ f).reshape(C, N, H, W_).permute(1, 2, 3, 0)
    return xh @ kernel + bias                   # Keras Conv2D(kernel_size=(1,1)) on NHWC: a per-pixel (C_in, C_out) product


def cpu_baseline(ratio, sites, config_name):
    import numpy as np
    import torch
    from oracle import wc_oracle as o
    cores = os.cpu_count() or 1
    g = torch.Generator(device="cpu"); g.manual_seed(1234)
    work = []
    for H, W, C in sites:
        k = torch.randn(C, C, generator=g) / C ** 0.5
        b = torch.zeros(C)
        work.append((torch.randn(64, H, W, C, generator=g), torch.randn(128, H, W, C, generator=g),
                     torch.randn(128, H, W, C, generator=g), k, b))

    def one():
        for x64, x128, gy, k, b in work:
            with torch.no_grad():
                for _ in range(ratio):
                    cpu_unfused_site(x64, k, b)
            xr = x128.detach().requires_grad_(True); kr = k.detach().requires_grad_(True); br = b.detach().requires_grad_(True)
            cpu_unfused_site(xr, kr, br).backward(gy)

    def timed(n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); one(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    # the checker: the float64 oracle on the baseline's own output (one mid-sized site)
    H, W, C = sites[min(2, len(sites) - 1)]
    rng = np.random.default_rng(7)
    xs = o.synth_activation(rng, (64, H, W, C), "well").astype(np.float32)
    Gs, Bs = o.synth_coloring(rng, C, 1)
    y_cpu = cpu_unfused_site(torch.from_numpy(xs), torch.from_numpy(Gs[0].astype(np.float32)), torch.from_numpy(Bs[0].astype(np.float32)))
    y_ref, _ = o.wc_forward(xs, Gs, Bs)
    err = float(np.abs(y_cpu.numpy() - y_ref).max() / np.abs(y_ref).max())

    prev = torch.get_num_threads()
    cands = sorted({t for t in (8, 16, 32, 64, 128) if t <= cores} | {min(cores, 128)})
    sweep = {}
    for t in cands:                       # short sweep (1 warm-up + 2 runs each), then the median of 10 at the best count
        torch.set_num_threads(t)
        one()
        sweep[t] = timed(2)
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    one(); one()
    t_med = timed(10)
    torch.set_num_threads(prev)
    return {"value": round(64.0 / t_med, 3), "unit": "images/sec (WC sites of one G+D step only)",
            "cores": best, "host_cores": cores, "kind": "port",
            "port_of": "the reference's UNFUSED fp32 op order (SURVEY rows a2 + a6) as a torch-CPU sequence -- the reference itself (py2 / TF 1.5 / Keras 2.0.8, arithmetic in an empty submodule) cannot run here",
            "threads_sweep_s": {str(k): round(v, 3) for k, v in sweep.items()},
            "checked_vs_oracle_rel_err": err,
            "sample": f"torch-CPU fp32, the reference's unfused op order (transpose, mean, f f^T/(M-1), shrink, cholesky, "
                      f"solve_triangular vs I, W f, transpose, 1x1 conv + bias; autograd backward) over the {len(sites)} generator "
                      f"WC sites of {config_name}: {ratio} forward passes at N=64 + 1 forward+backward at N=128 = the WC work "
                      f"of one G+D step; median of 10 after 2 warm-ups = {t_med:.3f} s on {best} threads"}


def cpu_port_baseline(ratio, sites, config_name):
    """The second CPU leg SURVEY.md section 8d names: the `_cpu` C-ABI restatement (oracle/wc_cpu.cpp, OpenMP, float64
    accumulation) on the same WC work of one step.  Test infrastructure timed beside the product, never a fallback."""
    import numpy as np
    from oracle import cpu_port as cp
    from oracle import wc_oracle as o
    rng = np.random.default_rng(1234)
    work = []
    for H, W, C in sites:
        G = (rng.standard_normal((1, C, C)) / C ** 0.5).astype(np.float32); B = np.zeros((1, C), np.float32)
        work.append((rng.standard_normal((64, H, W, C), dtype=np.float32), rng.standard_normal((128, H, W, C), dtype=np.float32),
                     rng.standard_normal((128, H, W, C), dtype=np.float32), G, B))

    def one():
        for x64, x128, gy, G, B in work:
            C = x64.shape[-1]
            for _ in range(ratio):
                M = x64.size // C
                s, xtx = cp.stats(x64.reshape(M, C))
                mu, L, Wm, _ = cp.factor(s, xtx, M, C)
                A, _ = cp.color(Wm, G)
                cp.apply(x64, mu, A, B, None)
            cp.forward_backward(x128, G, B, None, gy)

    # the checker: the numpy oracle on the port's own output (one mid-sized site)
    H, W, C = sites[min(2, len(sites) - 1)]
    xs = o.synth_activation(rng, (16, H, W, C), "well").astype(np.float32)
    Gs, Bs = o.synth_coloring(rng, C, 1)
    y_port = cp.forward_backward(xs, Gs.astype(np.float32), Bs.astype(np.float32), None, np.zeros_like(xs))[0]
    y_ref, _ = o.wc_forward(xs, Gs, Bs)
    err = float(np.abs(y_port - y_ref).max() / np.abs(y_ref).max())
    one()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); one(); ts.append(time.perf_counter() - t0)
    t_med = sorted(ts)[1]
    return {"value": round(64.0 / t_med, 3), "unit": "images/sec (WC sites of one G+D step only)",
            "cores": int(cp.load().wc_cpu_threads()), "kind": "port", "checked_vs_oracle_rel_err": err,
            "sample": f"oracle/wc_cpu.cpp through its `_cpu` C ABI (OpenMP loops, float64 accumulation, stage by stage as the HIP "
                      f"library) over the {len(sites)} generator WC sites of {config_name}: {ratio} forward passes at N=64 + 1 "
                      f"forward+backward at N=128; median of 3 after 1 warm-up = {t_med:.3f} s"}


# ---------------------------------------------------------------------------------------------------------------------
def multi_rank_diagnostics(trainer, eager_step, my_s_per_step, world, group, dev, nsteps):
    """What the first N > 1 run should say about itself: every rank's own ms per step (min / max over the
    ranks: a straggler shows), the size of the process group as the backend reports it, and the share of a step spent in the
    gradient all-reduces -- measured on `nsteps` extra eager steps OUTSIDE the timed region, with events around the
    collectives (train.AllreduceTimer).  All ranks call this; the numbers are the same on each."""
    import torch
    import torch.distributed as dist
    from wc_gan_amd.train import AllreduceTimer
    tm = AllreduceTimer()
    for b in (trainer.g_bucket, trainer.d_bucket):
        b.timer = tm
    for _ in range(nsteps):
        eager_step()
    if dev is not None:
        torch.cuda.synchronize()
    for b in (trainer.g_bucket, trainer.d_bucket):
        b.timer = None
    ar_ms = tm.total_ms() / nsteps
    per_rank = [my_s_per_step * 1e3]
    ar_all = [ar_ms]
    seen = 1
    if group is not None:
        seen = dist.get_world_size(group)
        t = torch.tensor([my_s_per_step * 1e3, ar_ms], dtype=torch.float64, device=dev if dev is not None else "cpu")
        out = [torch.zeros_like(t) for _ in range(seen)]
        dist.all_gather(out, t, group=group)
        per_rank = [float(o[0]) for o in out]
        ar_all = [float(o[1]) for o in out]
    return {"ranks_seen_by_backend": seen, "ms_per_step_rank_min": round(min(per_rank), 3), "ms_per_step_rank_max": round(max(per_rank), 3),
            "allreduce_ms_per_step": round(max(ar_all), 3), "allreduce_calls_per_step": tm.calls // max(nsteps, 1),
            "allreduce_bytes_per_step": int(sum(b.flat.numel() * 4 * n for b, n in ((trainer.d_bucket, trainer.training_ratio), (trainer.g_bucket, 1))
                                            if b.flat is not None))}


def dry_run(args, world, rank):
    """The N-rank path on the CPU: rendezvous over gloo, flat gradient buckets, GanTrainer.step with the all-reduces,
    the cut points capture_segments() would use -- with a tiny model that contains no HIP layer (batch norm + diagonal
    coloring, plain convolutions), since the WC path has no CPU fallback."""
    import torch
    import torch.distributed as dist
    from wc_gan_amd.discriminator import make_discriminator
    from wc_gan_amd.generator import make_generator
    from wc_gan_amd.train import GanTrainer, broadcast_state
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        group = dist.group.WORLD
        assert dist.get_world_size() == args.gpus
    torch.manual_seed(1234)
    G = make_generator(block_sizes=(8, 8), resamples=("UP", "UP"), first_block_shape=(4, 4, 8), block_norm='b',
                       block_after_norm='ucs', last_norm='b', last_after_norm='ucs')
    D = make_discriminator(input_image_shape=(16, 16, 3), block_sizes=(8, 8), resamples=('DOWN', 'SAME'), type=None,
                           spectral=False, sum_pool=True)
    with torch.no_grad():                        # build the lazy layers, then make the replicas identical
        G(torch.zeros(2, 128), torch.zeros(2, 1, dtype=torch.int32))
    broadcast_state(G, group=group); broadcast_state(D, group=group)
    ratio = min(args.training_ratio, 2)
    tr = GanTrainer(G, D, batch_size=4, training_ratio=ratio, process_group=group, seed=1234, flat_buckets=True)
    g = torch.Generator(device="cpu"); g.manual_seed(1234 + rank)
    reals = [torch.rand(4, 16, 16, 3, generator=g) * 2 - 1 for _ in range(ratio)]
    for _ in range(args.warmup):
        tr.step(reals)
    order, _ = tr.trace_boundaries(reals)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step(reals)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    diag = multi_rank_diagnostics(tr, lambda: tr.step(reals), dt / args.steps, world, group, None, max(1, min(2, args.steps)))
    w = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())])
    same = True
    orders = [order]
    if world > 1:
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        same = all(torch.equal(ws[0], t) for t in ws)          # averaged gradients + identical start -> identical replicas
        orders = [None] * world
        dist.all_gather_object(orders, order)
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        print(json.dumps({"metric": "images/sec G+D step (dry run: tiny CPU model, not a measurement)", "dry_run": True,
                          "value": round(4.0 * world * args.steps / dt, 2), "unit": "images/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "replicas_identical": bool(same), "allreduce_order": orders, "finite": bool(torch.isfinite(w).all()),
                          "config": {"launch": "eager", "launch_fallback": None}, "multi_gpu": diag,
                          # the GPU-only legs go through safe() here too: on the CPU they throw, and the line is printed all the same
                          "roofline": safe(lambda: roofline_apply(torch.device("cpu")), "roofline_apply (dry run: expected to fail without a GPU)")}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if same and all(o_ == orders[0] for o_ in orders) else 1


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args, argv)           # before any GPU call in this process
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a {world}-rank run as {args.gpus} GPUs",
              file=sys.stderr)
        return 2
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the WC path has no CPU fallback (--dry-run rehearses the launch path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group = None
    if world > 1 or os.environ.get("WC_FORCE_COLLECTIVES") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        group = dist.group.WORLD
        if dist.get_world_size() != args.gpus:
            print(f"[bench] RCCL sees {dist.get_world_size()} ranks, --gpus {args.gpus}", file=sys.stderr)
            return 2

    from wc_gan_amd import _lib
    from wc_gan_amd.train import CONFIGS, build_trainer
    _lib.load()
    cfg = CONFIGS[args.config]
    torch.manual_seed(1234)
    kw = dict(flat_buckets=True) if os.environ.get("WC_FORCE_COLLECTIVES") == "1" else {}      # development: the N > 1 layout on one GPU
    trainer = build_trainer(cfg, dev, process_group=group, sync_wc=args.sync_wc, **kw,
                            training_ratio=args.training_ratio, seed=1234)      # (the trainer adds the rank to the seed)
    H, W, Ci = cfg['image_shape']
    K = cfg['generator']['number_of_classes']
    g = torch.Generator(device="cpu"); g.manual_seed(1234 + rank)
    reals = [(torch.rand(64, H, W, Ci, generator=g) * 2 - 1).to(dev) for _ in range(args.training_ratio)]
    labels = None
    if cfg['conditional']:            # real images come with their labels (run.py:317 supervised=True)
        labels = [torch.randint(0, K, (64, 1), generator=g, dtype=torch.int32).to(dev) for _ in range(args.training_ratio)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_ok(ok):
        """the launch mode is agreed by ALL ranks: one rank replaying graphs next to an eager one is not a supported mix"""
        if group is None:
            return ok
        flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    eager_step = lambda: trainer.step(reals, labels)
    step = eager_step
    for _ in range(args.warmup):
        step()
    launch_mode = "eager"
    launch_fallback = None            # why a requested / default graph mode was not used (None: it was)
    in_group = group is not None
    if args.graph and in_group:
        # tried on one GPU with a one-rank group: capturing the RCCL all-reduce aborts the process (SIGABRT inside the
        # capture) -- the chain of graphs with the collectives between them is the supported form
        print("[bench] --graph with a process group: using --segments (collectives cannot be captured)", file=sys.stderr)
        args.graph, args.segments = False, True
    if args.segments or (world > 1 and not args.eager and not args.sync_wc):
        replay, ok = None, True
        try:
            replay = trainer.capture_segments(reals, labels)
        except Exception as exc:
            ok = False
            launch_fallback = f"rank {rank}: segment capture failed ({type(exc).__name__}: {str(exc)[:160]})"
            print(f"[bench] {launch_fallback}", file=sys.stderr)
            torch.cuda.synchronize()
        if all_ok(ok):
            step, launch_mode = replay, "hipgraph-segments"
            step()
        else:
            launch_fallback = launch_fallback or "segment capture failed on another rank: all ranks run eagerly"
            print("[bench] segment capture did not succeed on every rank: all ranks run eagerly", file=sys.stderr)
    elif args.graph or (world == 1 and not args.eager):
        try:
            step = trainer.capture(reals, labels)        # the whole G+D step as one hipGraph: the host leaves the loop
            launch_mode = "hipgraph"
            step()
        except Exception as exc:                 # fall back, say so
            launch_fallback = f"graph capture failed ({type(exc).__name__}: {str(exc)[:160]})"
            print(f"[bench] {launch_fallback}; running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            step = eager_step
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dt_mine = dt
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # training_ratio is an upstream-recall default -> also report the step at ratio 1.  Launched the way the
    # headline step is (round 2 timed it eagerly, where it is host-bound: 3 675 images/sec on the driver's box against
    # 5 400-5 940 on the builder's)
    dt1 = None
    launch1 = "eager"
    if args.training_ratio != 1:
        trainer.training_ratio = 1
        step1 = eager_step
        for _ in range(2):
            eager_step()
        if launch_mode != "eager":
            ok1, rep1 = True, None
            try:
                rep1 = (trainer.capture_segments if launch_mode == "hipgraph-segments" else trainer.capture)(reals[:1], labels[:1] if labels else None)
            except Exception as exc:
                ok1 = False
                print(f"[bench] rank {rank}: ratio-1 capture failed ({type(exc).__name__}: {exc}); timing it eagerly", file=sys.stderr)
                torch.cuda.synchronize()
            if all_ok(ok1):
                step1, launch1 = rep1, launch_mode
                step1()
        barrier()
        t1 = time.perf_counter()
        n1 = max(3, args.steps // 2)
        for _ in range(n1):
            step1()
        barrier()
        dt1 = (time.perf_counter() - t1) / n1
        if world > 1:
            tm = torch.tensor([dt1], device=dev, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt1 = float(tm.item())
        trainer.training_ratio = args.training_ratio

    dt_eager = None
    if launch_mode != "eager":                   # the same step launched kernel by kernel, for the record
        ne = max(3, args.steps // 2)
        eager_step(); barrier()
        te = time.perf_counter()
        for _ in range(ne):
            eager_step()
        barrier()
        dt_eager = (time.perf_counter() - te) / ne

    # N > 1: what the run says about itself (all ranks take part; outside every timed region).  Rank 0 then spends a few
    # seconds in roofline_apply() while the others wait in the final barrier: fine under the process group's 10-minute watchdog.
    diag = None
    if in_group:
        diag = multi_rank_diagnostics(trainer, eager_step, dt_mine / args.steps, world, group, dev, 2)
        diag["launch_agreed"] = launch_mode
        diag["launch_fallback"] = launch_fallback
    if rank == 0:
        extra = {}
        if diag is not None:
            extra["multi_gpu"] = diag
        if dt_eager is not None:
            extra["eager_launch"] = {"value": round(64.0 * world / dt_eager, 2), "unit": "images/sec", "ms_per_step": round(dt_eager * 1e3, 3)}
        if dt1 is not None:
            extra["training_ratio_1"] = {"value": round(64.0 * world / dt1, 2), "unit": "images/sec", "ms_per_step": round(dt1 * 1e3, 3),
                                         "launch": launch1}
        roof = safe(lambda: roofline_apply(dev), "roofline_apply")
        cpu = None
        if world == 1:
            sites = site_list(args.config)
            extra["roofline_conv"] = safe(lambda: conv_roofline(dev), "conv_roofline")
            def _wc_gpu():
                wc_gpu = wc_sites_gpu(dev, args.training_ratio, sites)
                return {"value": round(64.0 / wc_gpu, 1), "unit": "images/sec (WC sites of one G+D step only)", "ms": round(wc_gpu * 1e3, 3)}
            extra["wc_sites_gpu"] = safe(_wc_gpu, "wc_sites_gpu")
            if not args.no_cpu_baseline:
                cpu = safe(lambda: cpu_baseline(args.training_ratio, sites, args.config), "cpu_baseline")
                if args.cpu_port:         # the C-ABI CPU restatement, 4.5 x slower than the torch-CPU leg and ~45 s of wall time: on request
                    try:
                        extra["cpu_port"] = cpu_port_baseline(args.training_ratio, sites, args.config)
                    except Exception as ex:                       # the port is test infrastructure: its absence is not an error of the bench
                        extra["cpu_port"] = {"value": None, "kind": "port", "error": repr(ex)[:200]}
        out = {
            "metric": "images/sec G+D step, CIFAR-10 ResNet-SN+WC, batch 64" if args.config == "cifar10_uncond"
                      else f"images/sec G+D step, {args.config} ResNet-SN+WC, batch 64",
            "value": round(64.0 * world * args.steps / dt, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{WORKLOADS[args.config]}, batch 64/GPU, training_ratio {args.training_ratio}, "
                                   "generator_batch_multiple 2",
                       "name": args.config, "parallelism": f"dp{world}",
                       "wc_statistics": "sync" if args.sync_wc else "per-replica", "launch": launch_mode,
                       "launch_fallback": launch_fallback},
            "roofline": roof, "cpu_baseline": cpu,
        }
        out.update(extra)
        # whatever the C side has buffered for stdout (RCCL's version banner when a communicator exists) goes out FIRST: the result is
        # then the last line a reader of this process's stdout sees, not followed by C-stdio text flushed at exit
        try:
            import ctypes
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if in_group:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
