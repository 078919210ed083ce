"""CPU tests of the C-ABI boundary: the shared library loads, exports every symbol include/wc_hip.h
declares, and its host-side argument checks and workspace sizing behave -- no kernel is launched."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wc_hip.h")


@pytest.fixture(scope="module")
def lib():
    from wc_gan_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.load()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wc_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_table_agree():
    from wc_gan_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_the_core_boundary_is_the_surveys_list_and_has_its_cpu_twin(lib):
    """include/wc_hip.h names THE boundary -- SURVEY.md section 8b's `stats / factor / apply / bwd x 3 / workspace` -- as
    WC_CORE_API; every entry is declared and exported, the compute entries have their `_cpu` conformance twin in oracle/libwc_cpu.so, and
    the list stays a list one can bind by hand (the rest of the header are extensions: fusions and data-format routes of the same stages)."""
    src = open(HEADER).read()
    m = re.search(r'#define WC_CORE_API((?:\s*"[^"]*"\s*\\?\n?)+)', src)
    assert m, "WC_CORE_API not found"
    core = " ".join(re.findall(r'"([^"]*)"', m.group(1))).split()
    assert len(core) == len(set(core)) and 8 <= len(core) <= 16
    stages = [n for n in core if not n.endswith("_workspace_bytes")]
    assert {"wc_stats_f32", "wc_factor_f64", "wc_apply_f32", "wc_bwd_reduce_f32", "wc_bwd_factor_f64", "wc_bwd_apply_f32"} <= set(stages)
    declared = set(declared_symbols())
    for name in core:
        assert name in declared and hasattr(lib, name), name
    twin = os.path.join(ROOT, "oracle", "libwc_cpu.so")
    if os.path.exists(twin):
        cpu = ctypes.CDLL(twin)
        for name in stages:
            assert hasattr(cpu, name + "_cpu"), name + "_cpu"


def test_every_declared_symbol_is_exported(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_abi_version_and_error_strings(lib):
    from wc_gan_amd import _lib
    assert lib.wc_abi_version() == _lib.ABI_VERSION == 8
    assert b"multiple of 32" in lib.wc_error_string(-3)
    assert lib.wc_error_string(0) == b"ok"


def test_workspace_sizes(lib):
    assert lib.wc_stats_workspace_bytes(131072, 256, 1) > 256 * 256 * 8
    assert lib.wc_stats_workspace_bytes(131072, 48, 1) == 0          # C not a multiple of 32
    assert lib.wc_stats_workspace_bytes(0, 64, 1) == 0
    assert lib.wc_factor_workspace_bytes(256, 1) >= 256 * 256 * 8
    assert lib.wc_bwd_factor_workspace_bytes(128, 10) >= 3 * 128 * 128 * 8
    per_sample = lib.wc_bwd_reduce_workspace_bytes(128, 1024, 128, 10, 1)
    whole = lib.wc_bwd_reduce_workspace_bytes(128, 1024, 128, 1, 0)
    assert per_sample >= 128 * 128 * 128 * 8 and whole > 0


def test_argument_checks_return_codes_without_touching_the_gpu(lib):
    one = ctypes.c_void_p(16)      # never dereferenced: every call below is rejected first
    assert lib.wc_apply_f32(None, None, None, None, None, 1, 1, 32, 1, None, None, None, 0, None) == -1
    assert lib.wc_apply_f32(one, None, one, None, None, 0, 1, 32, 1, one, None, None, 0, None) == -2
    assert lib.wc_apply_f32(one, None, one, None, None, 1, 1, 40, 1, one, None, None, 0, None) == -3
    assert lib.wc_stats_f32(one, 64, 64, 1, one, one, one, 16, None) == -4
    assert lib.wc_factor_f64(one, one, 64, 64, 1, 0.0, 0.99, 1, 1, None, None, one, None, one, one, one, 1 << 30, None) == -5
    assert lib.wc_factor_f64(one, one, 1, 64, 1, 1e-3, 0.99, 1, 1, None, None, one, None, one, one, one, 1 << 30, None) == -2
    assert lib.wc_color_f32(one, None, 2, 64, 1, 0, one, None, None, None, None, 0, None) == -2
    assert lib.wc_bwd_reduce_f32(one, None, one, None, 4, 16, 64, 3, one, one, one, 1 << 30, None) == -2
    assert lib.wc_stream_copy_f32(one, one, 6, None) == -2


def test_factor_mix_entry_points_check_their_arguments(lib):
    """ABI 6 (the soft-assignment coloring's dictionary mix): support predicate, sizes and rejections, no kernel launched."""
    one = ctypes.c_void_p(16)
    assert lib.wc_factor_mix_supported(15, 128) == 1 and lib.wc_factor_mix_supported(32, 256) == 1
    assert lib.wc_factor_mix_supported(33, 128) == 0 and lib.wc_factor_mix_supported(0, 128) == 0 and lib.wc_factor_mix_supported(4, 30) == 0
    assert lib.wc_factor_mix_bwd_workspace_bytes(15, 128) >= 15 * 128 * 8 and lib.wc_factor_mix_bwd_workspace_bytes(0, 128) == 0
    assert lib.wc_factor_mix_f32(None, one, None, None, 4, 32, 10, 10, one, None) == -1
    assert lib.wc_factor_mix_f32(one, one, None, None, 4, 32, 10, 7, one, None) == -2          # no idx: one table per class
    assert lib.wc_factor_mix_f32(one, one, one, None, 40, 32, 10, 7, one, None) == -2          # E beyond 32
    assert lib.wc_factor_mix_bwd_f32(one, one, one, None, 4, 32, 10, 7, one, one, None, None, 0, None) == -1
    assert lib.wc_factor_mix_bwd_f32(one, one, one, one, 4, 32, 10, 7, one, one, None, one, 8, None) == -4


def test_split_entry_points_check_their_arguments(lib):
    """ABI 4 (pre-split activations): sizes, shape support and rejections, no kernel launched."""
    one = ctypes.c_void_p(16)
    assert lib.wc_split_bytes(1024, 256) == 1024 * 256 * 4 and lib.wc_split_bytes(1024, 40) == 0
    assert lib.wc_apply_split_supported(128, 1024, 256) == 1 and lib.wc_apply_split_supported(128, 16, 128) == 1
    assert lib.wc_apply_split_supported(128, 1024, 64) == 0 and lib.wc_apply_split_supported(3, 5, 256) == 0
    assert lib.wc_stats_split_supported(131072, 256, 1) == 1 and lib.wc_stats_split_supported(8192, 256, 1) == 0
    assert lib.wc_stats_split_supported(327680, 256, 5) == 1 and lib.wc_stats_split_supported(131072, 64, 1) == 0
    assert lib.wc_stats_split_workspace_bytes(131072, 256, 1) > 36 * 32 * 32 * 8 and lib.wc_stats_split_workspace_bytes(8192, 256, 1) == 0
    assert lib.wc_apply_split_workspace_bytes(256, 10) > 10 * 256 * 4
    assert lib.wc_split_f32(None, None, one, 64, 64, 0, one, None, None) == -1
    assert lib.wc_split_f32(one, None, one, 64, 64, 2, one, None, None) == -5
    assert lib.wc_split_scales_f32(one, 64, 48, one, one, one, None) == -3
    assert lib.wc_unsplit_f32(one, None, one, 0, 64, one, None) == -2
    assert lib.wc_apply_split_f16x2(one, None, one, None, one, None, None, 128, 1024, 64, 1, 0, one, None, one, 1 << 30, None) == -2
    assert lib.wc_apply_split_f16x2(one, None, one, None, one, None, None, 128, 1024, 256, 1, 0, one, None, one, 16, None) == -4
    assert lib.wc_stats_split_f16x2(one, one, one, 8192, 256, 1, one, one, one, 1 << 30, None) == -2
    assert lib.wc_split_bias_f32(None, None, None, None, 1, 64, one, None) == -1


def test_producer_entry_points_check_their_arguments(lib):
    """ABI 5 (the residual add as the producer of pre-split planes; K1 + K2 and the K3 epilogues on planes): no kernel launched."""
    one = ctypes.c_void_p(16)
    assert lib.wc_resadd_split_supported(128, 32, 32, 256) == 1 and lib.wc_resadd_split_supported(128, 32, 32, 64) == 0
    assert lib.wc_resadd_f32(None, None, 4, 8, 8, 64, 0, one, None) == -1
    assert lib.wc_resadd_f32(one, one, 4, 7, 8, 64, 1, one, None) == -2             # up: even output planes only
    assert lib.wc_resadd_f32(one, one, 4, 8, 8, 40, 0, one, None) == -3
    assert lib.wc_resadd_f32(one, one, 4, 8, 8, 64, 2, one, None) == -5
    assert lib.wc_resadd_split_f32(one, one, 4, 8, 8, 64, 0, one, one, one, one, None, None) == -2      # planes: C = 128 | 256
    assert lib.wc_resadd_split_f32(one, one, 4, 8, 8, 128, 0, None, one, one, one, None, None) == -1
    assert lib.wc_patch_sum_f32(None, 4, 4, 4, 64, one, None) == -1 and lib.wc_patch_sum_f32(one, 4, 0, 4, 64, one, None) == -2
    assert lib.wc_fold_channel_scale_f32(one, 256, 1, 256, 256, None, None, one, one, one, None) == -1
    assert lib.wc_unfold_channel_scale_f32(one, one, 256, 1, 0, 256, one, one, one, None) == -2
    assert lib.wc_whiten_split_workspace_bytes(131072, 256, 1) > lib.wc_stats_split_workspace_bytes(131072, 256, 1) > 0
    assert lib.wc_whiten_split_workspace_bytes(8192, 256, 1) == 0
    assert lib.wc_whiten_split_error_offset(131072, 256, 1) > 0 and lib.wc_whiten_split_error_offset(131072, 64, 1) == 0
    args = (one, one, one, 131072, 256, 1, 1e-3, 0.99, 1, None, None, one, one, one, one, 1 << 30, None)
    assert lib.wc_whiten_split_f16x2(*args[:6], 0.0, *args[7:]) == -5
    assert lib.wc_whiten_split_f16x2(one, one, one, 8192, 256, 1, 1e-3, 0.99, 1, None, None, one, one, one, one, 1 << 30, None) == -2
    assert lib.wc_whiten_split_f16x2(*args[:15], 16, None) == -4
    ex = lambda y, mask, planes, osc, relu=1, hw=1024: lib.wc_apply_split_ex_f16x2(one, None, one, None, one, None, None, 128, hw, 256, 1, relu,
                                                                                    y, mask, planes, osc, None, one, 1 << 30, None)
    assert ex(None, None, None, None) == -1 and ex(one, None, one, one) == -1          # exactly one destination
    assert ex(None, None, one, None) == -1                                              # planes need their scale record
    assert ex(one, one, None, None, relu=0) == -5                                       # a mask belongs to a ReLU
    assert lib.wc_apply_split_ex_f16x2(one, None, one, None, one, None, None, 3, 5, 256, 1, 1, one, one, None, None, None, one, 1 << 30, None) == -2


def test_host_wrappers_refuse_cpu_tensors():
    import torch
    from wc_gan_amd import _lib, ops
    with pytest.raises(_lib.WcHipError):
        ops.stats(torch.zeros(64, 32))


def test_fast_apply_takes_the_grouped_and_conditional_sites(lib):
    """wc_apply_workspace_bytes > 256 <=> the split-fp16 kernel takes the shape.  The grouped / conditional sites of the
    STL-10 and CIFAR-10-cond recipes (HW = 144 or 64 rows per sample, not a multiple of the 2 x 8192/C-row register
    tile) used to fall back to the f32-MFMA kernel."""
    assert lib.wc_apply_workspace_bytes(320, 144, 256, 5) > 256       # STL-10 12x12, five statistic groups
    assert lib.wc_apply_workspace_bytes(320, 64, 128, 50) > 256       # CIFAR-10 cond 8x8, 5 groups x 10 classes
    assert lib.wc_apply_workspace_bytes(128, 144, 256, 7) > 256       # per-class tables, tiles straddle samples
    assert lib.wc_apply_workspace_bytes(128, 16, 256, 1) > 256        # 2048 rows (the 4x4 site): one planned launch since round 2
    assert lib.wc_apply_workspace_bytes(8, 36, 256, 1) == 256         # 288 rows: below the fast path's minimum


def test_round5_harness_entry_points_check_their_arguments(lib):
    """ABI 7, the harness side (one-launch split with a history record; the narrow-side layers): support predicates, sizes, rejections."""
    one = ctypes.c_void_p(16)
    assert lib.wc_conv_split_hist_f32(None, 64, 0, one, one, one, None, 0, one, 1, None) == -5          # (wc_conv.hip's own codes: ARG)
    assert lib.wc_conv_split_hist_f32(one, 62, 0, one, one, one, None, 0, one, 1, None) == -5           # n % 4
    assert lib.wc_conv_split_hist_f32(one, 64, 0, one, one, one, None, 0, None, 0, None) == -5          # no record
    assert lib.wc_conv_split_hist_f32(one, 64, 0, one, one, one, one, 48, one, 0, None) == -2           # column sums: 256 % (C / 4) != 0
    assert lib.wc_conv_wrw_narrow_supported(128, 32, 32, 3, 128, 3) == 1 and lib.wc_conv_wrw_narrow_supported(128, 16, 16, 3, 256, 1) == 1
    assert lib.wc_conv_wrw_narrow_supported(128, 32, 32, 4, 128, 3) == 0                                # 36 rows
    assert lib.wc_conv_wrw_narrow_supported(128, 32, 32, 3, 64, 3) == 0                                 # Cout % 128
    assert lib.wc_conv_wrw_narrow_supported(128, 32, 32, 3, 128, 5) == 0
    assert lib.wc_conv_wrw_narrow_workspace_bytes(128, 32, 32, 3, 128, 3) == 256 * 4096 * 4
    assert lib.wc_conv_wrw_narrow_workspace_bytes(2, 6, 10, 2, 256, 3) == 2 * 4096 * 4                  # one partial per 128 channels
    assert lib.wc_conv_wrw_narrow_workspace_bytes(128, 32, 32, 3, 64, 3) == 0
    assert lib.wc_conv_wrw_narrow_f32(None, one, 128, 32, 32, 3, 128, 3, one, 1, 27, 9, 3, None, one, 1 << 30, None) == -5
    assert lib.wc_conv_wrw_narrow_f32(one, one, 128, 32, 32, 3, 64, 3, one, 1, 27, 9, 3, None, one, 1 << 30, None) == -2
    assert lib.wc_conv_wrw_narrow_f32(one, one, 128, 32, 32, 3, 128, 3, one, 1, 27, 9, 3, None, one, 16, None) == -4
    assert lib.wc_conv_fwd_narrow_f32(one, None, 1, 27, 9, 3, None, 128, 32, 32, 3, 128, 3, 0, one, None) == -5
    assert lib.wc_conv_fwd_narrow_f32(one, one, 1, 27, 9, 3, None, 128, 32, 32, 8, 128, 3, 0, one, None) == -2


def test_round5_producer_entry_points_check_their_arguments(lib):
    """ABI 7, the producer that feeds K1 (wc_resadd_stats_split_f32) and the K1 tails that read its partials: predicates, sizes, rejections."""
    one = ctypes.c_void_p(16)
    assert lib.wc_resadd_stats_supported(128, 32, 32, 256, 1, 1) == 1 and lib.wc_resadd_stats_supported(320, 32, 32, 256, 1, 5) == 1
    assert lib.wc_resadd_stats_supported(128, 32, 32, 256, 0, 1) == 0           # up = 0: no shortcut rows to add per patch
    assert lib.wc_resadd_stats_supported(128, 32, 32, 64, 1, 1) == 0            # planes: C = 128 | 256
    assert lib.wc_resadd_stats_supported(4, 8, 8, 256, 1, 1) == 0               # below the fast reduction's minimum
    nb = lib.wc_resadd_stats_workspace_bytes(128, 32, 32, 256, 1)
    assert nb > 36 * 32 * 32 * 8 and lib.wc_resadd_stats_workspace_bytes(4, 8, 8, 256, 1) == 0
    assert lib.wc_whiten_presummed_error_offset(131072, 256, 1) > 0 and lib.wc_whiten_presummed_error_offset(131072, 64, 1) == 0
    assert lib.wc_resadd_stats_split_f32(one, None, 128, 32, 32, 256, 1, 1, one, one, one, one, None, one, nb, None) == -1
    assert lib.wc_resadd_stats_split_f32(one, one, 128, 32, 32, 64, 1, 1, one, one, one, one, None, one, nb, None) == -2
    assert lib.wc_resadd_stats_split_f32(one, one, 128, 32, 32, 256, 1, 1, one, one, one, one, None, one, 16, None) == -4
    assert lib.wc_whiten_presummed_f16x2(None, 131072, 256, 1, 1e-3, 0.99, 1, None, None, one, one, one, one, nb, None) == -1
    assert lib.wc_whiten_presummed_f16x2(one, 131072, 256, 1, 0.0, 0.99, 1, None, None, one, one, one, one, nb, None) == -5
    assert lib.wc_whiten_presummed_f16x2(one, 131072, 256, 1, 1e-3, 0.99, 1, None, None, one, one, one, one, 16, None) == -4
    assert lib.wc_stats_presummed_f16x2(one, 131072, 256, 1, None, one, one, nb, None) == -1
    assert lib.wc_stats_presummed_f16x2(one, 131072, 256, 1, one, one, one, 16, None) == -4


def test_backward_on_planes_takes_both_generator_widths(lib):
    """ABI 7: wc_bwd_xsplit_supported at C = 128 (the conditional CIFAR-10 / Tiny-ImageNet generators) as at C = 256; the bit-mask forms
    of K4 / K6 stay with C = 256 (wc_bwd_bits_supported), and the C = 128 planes entries refuse a mask."""
    one = ctypes.c_void_p(16)
    assert lib.wc_bwd_xsplit_supported(128, 1024, 256, 0) == 1 and lib.wc_bwd_xsplit_supported(128, 1024, 128, 1) == 1
    assert lib.wc_bwd_xsplit_supported(64, 4096, 128, 1) == 1 and lib.wc_bwd_xsplit_supported(128, 1024, 64, 0) == 0
    assert lib.wc_bwd_xsplit_supported(4, 16, 128, 0) == 0                      # below the fast reduction's minimum
    assert lib.wc_bwd_bits_supported(128, 1024, 128, 0) == 0 and lib.wc_bwd_bits_supported(128, 1024, 256, 0) == 1
    assert lib.wc_bwd_reduce_xsplit_f32(one, one, one, one, one, one, None, 128, 1024, 128, 1, one, one, one, one, 1 << 30, None) == -2
    assert lib.wc_bwd_apply_xsplit_f32(one, one, one, one, one, one, one, one, one, None, 128, 1024, 128, 1, one, one, one, 1 << 30, None) == -2
