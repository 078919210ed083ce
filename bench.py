#!/usr/bin/env python3
"""bench.py -- images/sec of the G+D training step, CIFAR-10 ResNet-SN + WC, batch 64 per GPU.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched through
torch.distributed.run, one rank per GPU over RCCL.  Rank 0 prints ONE JSON line.

  step      one "G+D step" = training_ratio critic updates (64 real + 64 generated, generator forward in
            train mode) + one generator update at batch 64 x 2 (run.py:101,293-294); images/sec = 64 x steps/s
            per GPU (weak scaling: every GPU keeps batch 64).
  roofline  the dominant hand-written kernel, the fused WC apply (K3) at the headline site
            128 x 32 x 32 x 256: algorithmic bytes 2*M*C*4 + (C*C + C)*4 per launch over the launch time
            measured here with HIP events on the launching stream, against the 8 TB/s HBM3E peak.
  cpu_baseline  the float64 numpy oracle (kind "port") timed on this box's host cores over the WC sites of
            one G+D step -- the hot path only, the convolutions are not part of it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
TRAINING_RATIO = 5             # default of the missing gan.cmd parser [UPSTREAM-RECALL]; ratio 1 also reported

# WC sites of the CIFAR-10 unconditional generator: (H=W, C) per site
CIFAR_SITES = [(4, 256), (8, 256), (8, 256), (16, 256), (16, 256), (32, 256), (32, 256)]


def time_kernel(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()                       # torch's current stream == the stream the C ABI launches on
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def roofline_apply(dev):
    from wc_gan_amd import ops
    N, H, C = 128, 32, 256
    M = N * H * H
    g = torch.Generator(device="cpu"); g.manual_seed(1234)
    # SURVEY section 8d kernel-bench input: x = z Mix + 0.2 (cond ~1e6), Gamma = randn/sqrt(C), beta = 0.1 randn
    z = torch.randn(M, C, generator=g)
    mix = torch.randn(C, C, generator=g) / C ** 0.5 + 0.3 * (torch.randn(C, 8, generator=g) @ torch.randn(8, C, generator=g)) / 8 ** 0.5
    x = (z @ mix + 0.2).view(N, H, H, C).to(dev)
    gamma = (torch.randn(1, C, C, generator=g) / C ** 0.5).to(dev)
    b = (0.1 * torch.randn(1, C, generator=g)).to(dev)
    # the apply exactly as the layer runs it: statistics -> factor -> color (which prepares the plan) -> K3
    s, xtx = ops.stats(x.view(M, C))
    mu, L, W, cs = ops.factor(s, xtx, M, C, 1e-3, 0.99, 1, True, None, None, dev, want_scale=True)
    A, At, plan = ops.color(W, gamma, cs)
    y = torch.empty_like(x)
    t = time_kernel(lambda: ops.apply(x, mu, A, b, None, out=y, plan=plan))
    alg_bytes = 2 * M * C * 4 + (C * C + C) * 4
    y2 = torch.empty_like(x)
    t_copy = time_kernel(lambda: ops.stream_copy(x, y2))
    achieved = alg_bytes / t / 1e9
    traffic = None          # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see the file)
    pmc = os.path.join(ROOT, "profiles", "r1_apply_k3_pmc.json")
    if os.path.exists(pmc):
        traffic = json.load(open(pmc)).get("traffic_bytes_per_launch")
    return {"bound": "hbm", "kernel": "affine_ring_kernel<256,false> (wc_apply_f32 with plan, 128x32x32x256 fp32)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "launch_us": round(t * 1e6, 2), "algorithmic_bytes": alg_bytes,
            "stream_copy_GBs": round(2 * M * C * 4 / t_copy / 1e9, 1),
            "frac_of_stream_copy": round(achieved / (2 * M * C * 4 / t_copy / 1e9), 4)}


MFMA_F16_PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md); the power-limited sustained rate measured here is 1430


def conv_roofline(dev):
    """The kernel the step spends most time in after the WC path: the 3x3 256->256 block convolution at 128x32x32
    (conv_f16x3_kernel<4,4>), MFMA-bound: 3 fp16 MFMA products per fp32 product."""
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


def wc_sites_gpu(dev, ratio):
    """GPU time of the WC sites of one G+D step (what cpu_baseline times on the host)."""
    from wc_gan_amd.functional import whiten_color, whiten_color_grouped
    g = torch.Generator(device="cpu"); g.manual_seed(1234)
    work = []
    for H, C in CIFAR_SITES:
        G = (torch.randn(1, C, C, generator=g) / C ** 0.5).to(dev).requires_grad_(True)
        B = torch.zeros(1, C, device=dev, requires_grad=True)
        xd = torch.randn(64 * ratio, H, H, C, generator=g).to(dev)      # the critic-phase passes, stacked (train.generate)
        x128 = torch.randn(128, H, H, C, generator=g).to(dev).requires_grad_(True)
        work.append((xd, x128, G, B, torch.randn(128, H, H, C, generator=g).to(dev)))

    def one():
        for xd, x128, G, B, gy in work:
            with torch.no_grad():
                whiten_color_grouped(xd, ratio, G, B)        # `ratio` forward passes with per-pass statistics
            whiten_color(x128, G, B).backward(gy)
    return time_kernel(one, iters=5, warm=2)


def cpu_baseline(ratio):
    import numpy as np
    from oracle import wc_oracle as o
    rng = np.random.default_rng(1234)
    t_total = 0.0
    for H, C in CIFAR_SITES:
        G, B = o.synth_coloring(rng, C, 1)
        x64 = rng.standard_normal((64, H, H, C)).astype(np.float32)
        x128 = rng.standard_normal((128, H, H, C)).astype(np.float32)
        gy = rng.standard_normal((128, H, H, C)).astype(np.float32)
        t0 = time.perf_counter()
        for _ in range(ratio):
            o.wc_forward(x64, G, B)
        y, cache = o.wc_forward(x128, G, B)
        o.wc_backward(gy, cache)
        t_total += time.perf_counter() - t0
    cores = os.cpu_count() or 1
    return {"value": round(64.0 / t_total, 3), "unit": "images/sec (WC sites of one G+D step only)",
            "cores": cores, "kind": "port",
            "sample": f"float64 numpy oracle, 7 generator WC sites, {ratio} forward passes at N=64 + 1 forward+backward "
                      f"at N=128 = the WC work of one G+D step ({t_total:.1f} s), BLAS threads = {cores}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--training-ratio", type=int, default=TRAINING_RATIO)
    ap.add_argument("--sync-wc", action="store_true", help="all-reduce WC statistics across replicas")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole G+D step as one hipGraph.  The default on one GPU: with ~2000 launches per step the "
                         "eager loop is at the edge of host-bound (26-32 ms per step depending on the box's CPU, against "
                         "26.5 ms of GPU work); the eager step is reported next to it")
    ap.add_argument("--segments", action="store_true",
                    help="replay the step as a chain of hipGraphs cut at the gradient all-reduces, which stay plain RCCL calls "
                         "(the default with several GPUs, unless --sync-wc puts collectives inside the WC layers)")
    ap.add_argument("--eager", action="store_true", help="launch kernel by kernel from Python (the default with several GPUs: "
                    "capturing the RCCL all-reduces could not be tried on the one-GPU development box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the WC path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group = None
    if world > 1 or os.environ.get("WC_FORCE_COLLECTIVES") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        group = dist.group.WORLD

    from wc_gan_amd import _lib
    from wc_gan_amd.train import CIFAR10_UNCOND, build_trainer
    _lib.load()
    torch.manual_seed(1234)
    kw = dict(flat_buckets=True) if os.environ.get("WC_FORCE_COLLECTIVES") == "1" else {}      # development: the N > 1 layout on one GPU
    trainer = build_trainer(CIFAR10_UNCOND, dev, process_group=group, sync_wc=args.sync_wc, **kw,
                            training_ratio=args.training_ratio, seed=1234 + rank)
    g = torch.Generator(device="cpu"); g.manual_seed(1234 + rank)
    reals = [(torch.rand(64, 32, 32, 3, generator=g) * 2 - 1).to(dev) for _ in range(args.training_ratio)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = lambda: trainer.step(reals)
    for _ in range(args.warmup):
        step()
    launch_mode = "eager"
    eager_step = step
    in_group = dist.is_available() and dist.is_initialized()
    if args.graph and in_group:
        # tried on one GPU with a one-rank group: capturing the RCCL all-reduce aborts the process (SIGABRT inside the
        # capture) -- the chain of graphs with the collectives between them is the supported form
        print("[bench] --graph with a process group: using --segments (collectives cannot be captured)", file=sys.stderr)
        args.graph, args.segments = False, True
    if args.segments or (world > 1 and not args.eager and not args.sync_wc):
        try:
            step = trainer.capture_segments(reals)
            launch_mode = "hipgraph-segments"
            step()
        except Exception as exc:
            print(f"[bench] segment capture failed ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            step = lambda: trainer.step(reals)
    elif args.graph or (world == 1 and not args.eager):
        try:
            step = trainer.capture(reals)        # the whole G+D step as one hipGraph: the host leaves the loop
            launch_mode = "hipgraph"
            step()
        except Exception as exc:                 # e.g. a collective that refuses capture: fall back, say so
            print(f"[bench] graph capture failed ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            step = lambda: trainer.step(reals)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # training_ratio is an upstream-recall default -> also report the step at ratio 1
    dt1 = None
    if args.training_ratio != 1:
        trainer.training_ratio = 1
        for _ in range(2):
            trainer.step(reals)
        barrier()
        t1 = time.perf_counter()
        n1 = max(3, args.steps // 2)
        for _ in range(n1):
            trainer.step(reals)
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

    extra = {}
    if rank == 0:
        if dt_eager is not None:
            extra["eager_launch"] = {"value": round(64.0 * world / dt_eager, 2), "unit": "images/sec", "ms_per_step": round(dt_eager * 1e3, 3)}
        if dt1 is not None:
            extra["training_ratio_1"] = {"value": round(64.0 * world / dt1, 2), "unit": "images/sec", "ms_per_step": round(dt1 * 1e3, 3)}
        roof = roofline_apply(dev)
        extra["roofline_conv"] = conv_roofline(dev)
        wc_gpu = wc_sites_gpu(dev, args.training_ratio)
        extra["wc_sites_gpu"] = {"value": round(64.0 / wc_gpu, 1), "unit": "images/sec (WC sites of one G+D step only)",
                                 "ms": round(wc_gpu * 1e3, 3)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.training_ratio)
        out = {
            "metric": "images/sec G+D step, CIFAR-10 ResNet-SN+WC, batch 64",
            "value": round(64.0 * world * args.steps / dt, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CIFAR-10 ResNet SN uncond + WC (scripts/cifar10_resnet_sn_uncond.sh), batch 64/GPU, "
                                   f"training_ratio {args.training_ratio}, generator_batch_multiple 2",
                       "parallelism": f"dp{world}", "wc_statistics": "sync" if args.sync_wc else "per-replica",
                       "launch": launch_mode},
            "roofline": roof, "cpu_baseline": cpu,
        }
        out.update(extra)
        print(json.dumps(out), flush=True)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
