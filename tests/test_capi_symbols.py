"""CPU suite: the C-ABI library loads and exports every symbol include/pdepth.h declares.
No compute calls here (no GPU); argument-validation paths return before any launch."""
import ctypes
import os
import re

import pytest

import pdepth_amd
from pdepth_amd import _native

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "pdepth.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pdepth_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_native.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    for sym in _declared_symbols():
        assert hasattr(lib, sym), f"{sym} declared in include/pdepth.h but not exported"
    assert lib.pdepth_abi_version() == 6


def test_argument_validation_without_gpu():
    lib = _native.load()
    desc = _native.SweepDesc(0, 1, 1, 1, 1, 1, 0, 0, 0, 10.0, 0, 0, 1)
    cam = _native.Camera(None, None, None, None, None)
    rc = lib.pdepth_sweep_cost_f32(ctypes.byref(desc), ctypes.byref(cam), None, None, None, 1, None, 0, None)
    assert rc == 1 and b"non-positive" in lib.pdepth_last_error()
    rc = lib.pdepth_dpv_reduce_f32(None, None, 1, 1, 1, 1, None, None, None)
    assert rc == 1 and b"null input" in lib.pdepth_last_error()
    desc = _native.SweepDesc(1, 1, 4, 8, 2, 2, 0, 0, 0, 1.0, 0, 0, 16)
    cam = _native.Camera(1, 1, 1, 1, 1)
    rc = lib.pdepth_warp_feature_f32(ctypes.byref(desc), ctypes.byref(cam), 1, 1, 1, None)
    assert rc == 1 and b"C == D" in lib.pdepth_last_error()
    desc = _native.SweepDesc(1, 1, 4, 8, 2, 2, 7, 0, 0, 1.0, 16, 16, 16)
    rc = lib.pdepth_sweep_dpv_f32(ctypes.byref(desc), ctypes.byref(cam), 1, 1, 1, 1, None, None, None, 0, None)
    assert rc == 1 and b"undefined metric" in lib.pdepth_last_error()
    desc = _native.SweepDesc(1, 1, 4, 8, 2, 2, 0, 0, 5, 1.0, 16, 16, 16)
    rc = lib.pdepth_sweep_dpv_f32(ctypes.byref(desc), ctypes.byref(cam), 1, 1, 1, 1, None, None, None, 0, None)
    assert rc == 1 and b"blas_mode" in lib.pdepth_last_error()


def test_argument_validation_of_the_round2_entries_without_gpu():
    lib = _native.load()
    cam = _native.Camera(1, 1, 1, 1, 1)
    desc = _native.SweepDesc(1, 1, 4, 8, 8, 16, 0, 0, 0, 1.0, 4 * 8 * 16, 4 * 8 * 16, 4 * 8 * 16)
    lib.pdepth_sweep_workspace_bytes.restype = ctypes.c_size_t
    need = lib.pdepth_sweep_workspace_bytes(ctypes.byref(desc))
    assert need > 0
    # workspace too small / missing / misaligned
    assert lib.pdepth_pack_source_f32(ctypes.byref(desc), 1, 256, need - 1, None) == 3 and b"workspace" in lib.pdepth_last_error()
    assert lib.pdepth_pack_source_f32(ctypes.byref(desc), 1, None, need, None) == 3
    assert lib.pdepth_pack_source_f32(ctypes.byref(desc), 1, 257, need, None) == 3 and b"aligned" in lib.pdepth_last_error()
    rc = lib.pdepth_sweep_dpv_packed_f32(ctypes.byref(desc), ctypes.byref(cam), 1, 1, None, 1, 1, 256, need - 1, None)
    assert rc == 3
    # the gather kernel has no packed form of its own
    direct = _native.SweepDesc(1, 1, 4, 8, 8, 16, 0, _native.ALGO_DIRECT, 0, 1.0, 512, 512, 512)
    assert lib.pdepth_pack_source_f32(ctypes.byref(direct), 1, 256, need, None) == 1 and b"packed source" in lib.pdepth_last_error()
    rc = lib.pdepth_sweep_dpv_packed_f32(ctypes.byref(direct), ctypes.byref(cam), 1, 1, None, 1, 1, 256, need, None)
    assert rc == 1 and b"packed source" in lib.pdepth_last_error()
    # cell-list kernels: L2 only
    cells_l1 = _native.SweepDesc(1, 1, 4, 8, 8, 16, 1, _native.ALGO_CELLS, 0, 1.0, 512, 512, 512)
    rc = lib.pdepth_sweep_dpv_f32(ctypes.byref(cells_l1), ctypes.byref(cam), 1, 1, 1, None, 1, 1, 256, need, None)
    assert rc == 1 and b"ALGO_CELLS" in lib.pdepth_last_error()
    # extended reduction
    assert lib.pdepth_dpv_reduce_ex_f32(1, None, 1, 1, 4, 4, 4, None, None, None, None, None, None) == 1 and b"no output" in lib.pdepth_last_error()
    assert lib.pdepth_dpv_reduce_ex_f32(1, None, 1, 1, 4, 2, 4, 1, None, None, None, 1, None) == 1 and b"quarter" in lib.pdepth_last_error()
    assert lib.pdepth_dpv_reduce_ex_f32(16, None, 1, 1, 4, 4, 4, 1, 16, None, None, None, None) == 1 and b"alias" in lib.pdepth_last_error()
    # inverse warp: sampling mode
    rc = lib.pdepth_inverse_warp_f32(1, 1, 1, 1, 1, 3, 4, 4, 7, 1, None, None)
    assert rc == 1 and b"unknown mode" in lib.pdepth_last_error()
    rc = lib.pdepth_inverse_warp_backward_f32(1, 1, 1, 1, 1, 1, 3, 4, 4, 0, None, None, None)
    assert rc == 1 and b"null pointer" in lib.pdepth_last_error()
    # uncertainty field: workspace
    lib.pdepth_ufield_workspace_bytes.restype = ctypes.c_size_t
    un = lib.pdepth_ufield_workspace_bytes(1, 8, 16)
    args = [1, 1, 1, None, 1, 4, 8, 16, 1, ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(1), ctypes.c_float(0), 0, ctypes.c_float(0), 1, 1]
    assert lib.pdepth_ufield_f32(*args, 256, un - 1, None) == 3
    assert lib.pdepth_ufield_f32(*args, None, un, None) == 3


def test_host_blas_probe_is_decisive():
    """The probe must reproduce torch's CPU matmul exactly with one of the two documented modes."""
    import numpy as np
    import torch
    mode = _native.host_blas_mode()
    assert mode in (_native.BLAS_FMA, _native.BLAS_SEPARATE)
    A = torch.randn(3, 3) * 50
    Bm = torch.randn(3, 777)
    C = A.matmul(Bm).numpy()
    a, b = A.numpy().astype(np.float64), Bm.numpy().astype(np.float64)
    p = [a[:, k:k + 1] * b[k:k + 1, :] for k in range(3)]
    f32 = np.float32
    if mode == _native.BLAS_FMA:
        want = ((p[0].astype(f32).astype(np.float64) + p[1]).astype(f32).astype(np.float64) + p[2]).astype(f32)
    else:
        want = ((p[0].astype(f32) + p[1].astype(f32)).astype(f32) + p[2].astype(f32)).astype(f32)
    assert np.array_equal(want, C)


def test_product_path_has_no_cpu_fallback():
    import torch
    from pdepth_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.dpv_expect(torch.zeros(1, 4, 2, 2), [1.0, 2.0, 3.0, 4.0])
    src = open(os.path.join(REPO, "probabilistic-depth_amd", "_native.py")).read()
    for root, _, files in os.walk(os.path.join(REPO, "probabilistic-depth_amd")):
        for f in files:
            if f.endswith(".py"):
                body = open(os.path.join(root, f)).read()
                assert "import oracle" not in body and "from oracle" not in body, f"{f} must not use the oracle"
    assert "oracle" not in src


def test_product_library_carries_no_lab_bench():
    """the default build holds the product kernels only -- the kernels of earlier rounds and their
    experiment switches live under csrc/lab/ (make LAB=1) --, a product sweep source has at most five preprocessor
    conditionals (diagnostic builds), and the shipped library does not answer the lab selectors."""
    import re
    csrc = os.path.join(REPO, "probabilistic-depth_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = re.search(r"^SRCS = (.*)$", mk, re.M).group(1).split()
    assert not any(s.startswith("lab/") for s in srcs) and "sweep_dist.hip" in srcs and "sweep_corr.hip" not in srcs and "ifdef LAB" in mk
    for f in srcs:
        if f.startswith("sweep_"):
            n = len(re.findall(r"^\s*#\s*if", open(os.path.join(csrc, f)).read(), re.M))
            assert n <= 5, f"{f}: {n} preprocessor conditionals"
    lib = _native.load()
    names = {"pdepth_sweep_centres_source", "pdepth_sweep_dpv_packed_f32", "pdepth_sweep_source_layout"}
    for n in names:
        assert hasattr(lib, n)
    assert (_native.ALGO_CORR, _native.ALGO_DIST) == (6, 7)
    # the lab selectors (the cell-list, fp32-matrix and correlation-form kernels of rounds 2 / 3 / 4) are refused by the product
    # library, by name
    cam = _native.Camera(1, 1, 1, 1, 1)
    for algo in (_native.ALGO_CELLS, _native.ALGO_MFMA, _native.ALGO_CORR):
        desc = _native.SweepDesc(1, 1, 4, 8, 8, 16, 0, algo, 0, 1.0, 512, 512, 512)
        rc = lib.pdepth_sweep_dpv_f32(ctypes.byref(desc), ctypes.byref(cam), 1, 1, 1, None, 1, 1, 256, 1 << 20, None)
        assert rc == 1 and b"lab builds only" in lib.pdepth_last_error()
    # ... and it reads no environment: every experiment switch of the sources (PDEPTH_SWEEP_IMPL, PDEPTH_CORR_FUSE_PACK,
    # PDEPTH_NO_SPEC, PDEPTH_CORR_NO_MFMA) sits behind #ifdef PDEPTH_LAB, so their names are not in the binary
    blob = open(_native.LIB_PATH, "rb").read()
    for name in (b"PDEPTH_SWEEP_IMPL", b"PDEPTH_CORR_FUSE_PACK", b"PDEPTH_NO_SPEC", b"PDEPTH_CORR_NO_MFMA"):
        assert name not in blob, name
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp")):
            body = open(os.path.join(csrc, f)).read()
            for m in re.finditer(r"getenv\(", body):
                head = body[:m.start()]
                opened = len(re.findall(r"^\s*#\s*ifdef PDEPTH_LAB", head, re.M)) + len(re.findall(r"^\s*#\s*if defined\(PDEPTH_LAB\)", head, re.M))
                assert opened > 0 and head.rfind("PDEPTH_LAB") > head.rfind("#endif"), f"{f}: getenv outside a lab-only block"


def test_header_says_what_the_code_does():
    """the boundary's prose is checked against the stale statements it once made."""
    h = open(os.path.join(REPO, "include", "pdepth.h")).read()
    for stale in ("arithmetic type is fp32 throughout", "(ceil(C/4)+2)*H*W*16 bytes, written by a pre-pass of every call",
                  "any cameras, depth candidates, sigma, metric, reference"):
        assert stale not in h, stale
    for needed in ("v_mfma_f32_16x16x32_f16", "320 bytes per texel", "PDEPTH_LAYOUT_DIST16", "never a clamped number", "ONE launch"):
        assert needed in h, needed
    assert "#define PDEPTH_ABI_VERSION 6" in h


def test_workspace_size_is_the_one_the_header_states():
    """include/pdepth.h gives the workspace of a sweep as a formula (the packed views in the larger of the two layouts, one
    ints per 16x4 tile + the counters, 496 floats of statistics per batch item): the library's answer is that, up to the
    256-byte roundings of its parts."""
    lib = _native.load()
    h = open(os.path.join(REPO, "include", "pdepth.h")).read()
    assert "rounded up to a multiple of 8" in h and "texel-group-major" in h
    for B, V, C, D, H, W in ((4, 1, 67, 64, 256, 512), (2, 4, 67, 128, 512, 1024), (1, 2, 22, 48, 37, 53), (3, 1, 8, 16, 20, 31), (1, 1, 40, 64, 64, 128)):
        desc = _native.SweepDesc(B, V, C, D, H, W, 0, 0, 0, 10.0, C * H * W, V * C * H * W, C * H * W)
        got = lib.pdepth_sweep_workspace_bytes(ctypes.byref(desc))
        nchk = 0 if C <= 8 else (1 if C <= 40 else 2)
        wp = (W + 2 + 7) // 8 * 8
        packed = max(B * V * (8 * nchk + 4) * (H + 2) * wp * 16 + 256 * B * V, B * V * ((C + 3) // 4 + 2) * H * W * 16)
        flags = 4 * (2 * B * ((H + 3) // 4) * ((W + 15) // 16) + 64)
        stats = 496 * 4 * B
        want = packed + flags + stats
        assert want <= got <= want + 4 * 256, (B, V, C, D, H, W, got, want)


def test_graft_entry_checks_the_current_abi():
    """__graft_entry__.build() asserts the library's ABI number: it must be the header's (a stale number fails the driver's
    build check while every test is green -- it happened in round 6)."""
    h = open(os.path.join(REPO, "include", "pdepth.h")).read()
    ver = int(re.search(r"#define PDEPTH_ABI_VERSION (\d+)", h).group(1))
    entry = open(os.path.join(REPO, "__graft_entry__.py")).read()
    assert f"pdepth_abi_version() == {ver}" in entry
    assert _native.load().pdepth_abi_version() == ver
