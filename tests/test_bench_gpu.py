"""The bench line's contract on a GPU: `python bench.py` (short run, no CPU baseline) prints ONE JSON line with the driver's keys, the
roofline object of the dominant kernel and a self-consistent achieved / frac."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "images/sec" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["config"]["launch"] in ("hipgraph", "eager", "segments")
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 64.0) < 0.5          # images/sec x s/step = the batch of 64
    r_ = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launch_us", "algorithmic_bytes", "isolated_copy_GBs", "loop_copy_GBs", "frac_of_isolated_copy", "frac_of_loop_copy",
              "k3_kernels", "site_stages", "forward_site_us"):
        assert k in r_, k
    assert r_["bound"] == "hbm" and r_["unit"] == "GB/s" and r_["peak"] == 8000.0
    assert abs(r_["achieved"] - r_["algorithmic_bytes"] / r_["launch_us"] / 1e3) < 0.02 * r_["achieved"]
    assert abs(r_["frac"] - r_["achieved"] / r_["peak"]) < 1e-3
    assert 0.3 < r_["frac_of_loop_copy"] <= r_["frac_of_isolated_copy"] * 1.05 < 1.05 and "error" not in r_
    assert r_["traffic"] is None or 0.9 < r_["traffic"] / r_["algorithmic_bytes"] < 1.5


def test_roofline_names_a_kernel_that_the_generator_step_launches():
    """`roofline.kernel` is the K3 kernel the LAYERS run at the headline site -- the same kernel name must show up
    in a profiled generator update pass at batch 128 (CIFAR-10 uncond: Generator.BN.Final at 128 x 32 x 32 x 256), and it must be
    the entry of k3_kernels marked as run by the layers; no best-of picking."""
    import torch
    from torch.profiler import ProfilerActivity, profile
    sys.path.insert(0, ROOT)
    import bench
    from wc_gan_amd.train import CONFIGS
    from wc_gan_amd.generator import make_generator
    roof = bench.roofline_apply(torch.device("cuda", 0))
    assert roof["kernel_match"] in roof["kernel"]
    marked = [k for k, v in roof["k3_kernels"].items() if v["run_by_the_layers_at_this_site"]]
    assert len(marked) == 1 and roof["kernel_match"] in marked[0]
    assert abs(roof["k3_kernels"][marked[0]]["launch_us"] - roof["launch_us"]) < 0.15 * roof["launch_us"]       # the same kernel, timed twice
    torch.manual_seed(0)
    G = make_generator(**CONFIGS["cifar10_uncond"]["generator"]).cuda()
    z = torch.randn(128, 128, device="cuda")
    G(z).sum().backward()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        G(z).sum().backward()
        torch.cuda.synchronize()
    names = " | ".join(e.key for e in prof.key_averages())
    assert roof["kernel_match"] in names, (roof["kernel_match"], names[:3000])
