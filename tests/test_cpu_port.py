"""The CPU restatement behind the `_cpu` C ABI (oracle/wc_cpu.cpp) against the numpy oracle: the two
checkers of the HIP library agree with each other, stage by stage and end to end, and the port exports a `_cpu` twin of
every stage of include/wc_hip.h with the header's error behaviour."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import cpu_port as cp
from oracle import wc_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_every_stage_of_the_header_has_a_cpu_twin():
    hdr = open(os.path.join(ROOT, "include", "wc_hip.h")).read()
    stages = ["wc_stats_f32", "wc_factor_f64", "wc_color_f32", "wc_group_bias_f32", "wc_apply_f32", "wc_apply_act_f32",
              "wc_bwd_reduce_f32", "wc_bwd_factor_f64", "wc_bwd_apply_f32"]
    lib = cp.load()
    for name in stages:
        assert re.search(r"\b" + name + r"\(", hdr), name
        assert hasattr(lib, name + "_cpu"), name
    assert lib.wc_cpu_threads() >= 1


@pytest.mark.parametrize("shape,Kc,cond", [((6, 8, 8, 64), 1, "ill"), ((8, 4, 4, 32), 3, "ill"), ((5, 6, 6, 96), 2, "well")])
def test_cpu_port_matches_the_oracle_forward_and_backward(shape, Kc, cond):
    rng = np.random.default_rng(3)
    N, C = shape[0], shape[-1]
    x = o.synth_activation(rng, shape, cond).astype(np.float32)
    G, B = o.synth_coloring(rng, C, Kc)
    G = G.astype(np.float32); B = B.astype(np.float32)
    slot = rng.integers(0, Kc, N).astype(np.int32) if Kc > 1 else None
    gy = rng.standard_normal(shape).astype(np.float32)
    mm0 = rng.standard_normal((C, 1)).astype(np.float32) * 0.1
    mc0 = np.eye(C, dtype=np.float32) + 0.01
    mm, mc = mm0.reshape(-1).copy(), mc0.copy()
    y, dx, dg, db = cp.forward_backward(x, G, B, slot, gy, moving_mean=mm, moving_cov=mc)
    idx = slot if slot is not None else np.zeros(N, int)
    y_ref, cache = o.wc_forward(x, G, B, idx, moving_mean=mm0, moving_cov=mc0)
    dx_ref, dG_ref, dB_ref = o.wc_backward(gy, cache)
    assert rel(y, y_ref) < 2e-6 and rel(dx, dx_ref) < 2e-5
    assert rel(dg, dG_ref) < 2e-6 and rel(db, dB_ref) < 2e-6
    assert rel(mm, cache["moving_mean"]) < 1e-6 and rel(mc, cache["moving_cov"]) < 1e-6


def test_cpu_port_stages_and_groups():
    rng = np.random.default_rng(4)
    C, groups, Mg = 64, 3, 200
    X = rng.standard_normal((groups * Mg, C)).astype(np.float32) * np.exp(rng.uniform(-1, 1, C)).astype(np.float32) + 0.3
    s, xtx = cp.stats(X, groups)
    for g in range(groups):
        Xg = X[g * Mg:(g + 1) * Mg].astype(np.float64)
        assert rel(s[g], Xg.sum(0)) < 1e-12 and rel(xtx[g], Xg.T @ Xg) < 1e-12
        assert np.abs(xtx[g] - xtx[g].T).max() == 0.0
    mm = np.zeros(C, np.float32); mc = np.eye(C, dtype=np.float32)
    mu, L, W, cs = cp.factor(s, xtx, Mg, C, moving_mean=mm, moving_cov=mc, groups=groups)
    mm_ref, mc_ref = np.zeros(C), np.eye(C)
    for g in range(groups):
        mu_ref, sigma = o.moments_to_stats(s[g], xtx[g], Mg)
        L_ref, W_ref = o.whitening_matrix(sigma, 1e-3)
        assert rel(mu[g], mu_ref) < 1e-6 and rel(L[g], L_ref) < 1e-12 and rel(W[g], W_ref) < 1e-10
        mm_ref, mc_ref = o.update_moving(mm_ref, mc_ref, mu_ref, sigma, 0.99)
    assert rel(mm, mm_ref) < 1e-6 and rel(mc, mc_ref) < 1e-6
    assert np.all(np.log2(cs) == np.round(np.log2(cs)))                     # powers of two
    # evaluation mode: the moving statistics as they are
    mu_e, L_e, W_e, _ = cp.factor(None, None, Mg, C, training=False, moving_mean=mm, moving_cov=mc)
    L_ref, W_ref = o.whitening_matrix(0.5 * (mc.astype(np.float64) + mc.astype(np.float64).T), 1e-3)
    assert rel(mu_e, mm) == 0.0 and rel(L_e, L_ref) < 1e-12 and rel(W_e, W_ref) < 1e-10


def test_cpu_port_error_codes():
    lib = cp.load()
    x = np.zeros((64, 48), np.float32); s = np.zeros(48); t = np.zeros((48, 48))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.wc_stats_f32_cpu(p(x), 64, 48, 1, p(s), p(t), None, 0, None) == -3          # WC_ERR_CHANNELS
    x = np.zeros((64, 32), np.float32); s = np.zeros(32); t = np.zeros((32, 32))
    assert lib.wc_stats_f32_cpu(None, 64, 32, 1, p(s), p(t), None, 0, None) == -1           # WC_ERR_NULL
    assert lib.wc_stats_f32_cpu(p(x), 0, 32, 1, p(s), p(t), None, 0, None) == -2            # WC_ERR_SHAPE
    assert lib.wc_stats_f32_cpu(p(x), 64, 32, 1, p(s), p(t), None, 0, None) == 0


def test_cpu_port_is_clean_under_asan_and_ubsan():
    """SURVEY.md section 5 (host-side sanitizer build of the C++ restatement): `make -C oracle asan`, then this file's other
    tests again in a child process against that library with the ASan runtime preloaded.  Any AddressSanitizer report or
    UBSan runtime error fails the child (-fno-sanitize-recover=undefined; ASan aborts by default)."""
    import shutil
    import subprocess
    import sys
    if os.environ.get("WC_CPU_PORT_LIB"):
        pytest.skip("already inside the sanitizer run")
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host toolchain")
    oracle_dir = os.path.join(ROOT, "oracle")
    r = subprocess.run(["make", "-C", oracle_dir, "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", OMP_NUM_THREADS="4",
               WC_CPU_PORT_LIB=os.path.join(oracle_dir, "libwc_cpu_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
    assert r.returncode == 0, out[-3000:]
