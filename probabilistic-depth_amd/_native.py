"""ctypes binding of the C ABI in include/pdepth.h (libpdepth_hip.so, built in-tree).

The product path has NO CPU fallback: if the shared library is missing or a tensor is not
a contiguous fp32 device tensor, these wrappers raise.  torch is used only for device
memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpdepth_hip.so")

METRIC_L2, METRIC_L1 = 0, 1
ALGO_AUTO, ALGO_DIRECT, ALGO_TILED_1, ALGO_TILED_2, ALGO_CELLS, ALGO_MFMA, ALGO_CORR, ALGO_DIST = 0, 1, 2, 3, 4, 5, 6, 7
LAYOUT_NONE, LAYOUT_C4, LAYOUT_C4_CENTRED, LAYOUT_DIST16 = 0, 1, 2, 3
BLAS_FMA, BLAS_SEPARATE = 0, 1

# every symbol include/pdepth.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = (
    "pdepth_abi_version", "pdepth_last_error", "pdepth_sweep_workspace_bytes",
    "pdepth_sweep_cost_f32", "pdepth_sweep_dpv_f32", "pdepth_dpv_reduce_f32",
    "pdepth_dpv_expect_f32", "pdepth_warp_feature_f32", "pdepth_sample_coords_f32",
    "pdepth_dpv_fuse_f32", "pdepth_correlation_forward_f32", "pdepth_inverse_warp_f32", "pdepth_inverse_warp_backward_f32",
    "pdepth_dpv_moments_f32", "pdepth_correlation_backward_f32",
    "pdepth_pack_source_f32", "pdepth_sweep_dpv_packed_f32", "pdepth_dpv_reduce_ex_f32",
    "pdepth_ufield_workspace_bytes", "pdepth_ufield_f32",
    "pdepth_correlation_output_size", "pdepth_correlation_forward_f16", "pdepth_correlation_backward_f16",
    "pdepth_pack_views_f32", "pdepth_sweep_centres_source", "pdepth_sweep_source_layout",
)


class SweepDesc(Structure):
    _fields_ = [
        ("B", c_int32), ("V", c_int32), ("C", c_int32), ("D", c_int32), ("H", c_int32), ("W", c_int32),
        ("metric", c_int32), ("algo", c_int32), ("blas_mode", c_int32), ("sigma", c_float),
        ("ref_bstride", c_int64), ("src_bstride", c_int64), ("src_vstride", c_int64),
    ]


class Camera(Structure):
    _fields_ = [("K", c_void_p), ("R", c_void_p), ("t", c_void_p), ("rays", c_void_p), ("cxcy", c_void_p)]


_lib = None
_host_blas = None


def host_blas_mode():
    """Which rounding the host BLAS behind torch's CPU matmul uses (include/pdepth.h PDEPTH_BLAS_*).

    The reference's CPU path computes K@R, K@t and (K@R)@rays with torch.matmul (MKL); whether MKL
    fuses multiply-adds depends on the host CPU.  One tiny CPU matmul is compared with both
    closed forms (products/sums evaluated in float64 then rounded, which is exact for one fma)
    so the kernels reproduce the sampling positions of the reference run on THIS host.
    Override with PDEPTH_BLAS_MODE=fma|separate.
    """
    global _host_blas
    if _host_blas is not None:
        return _host_blas
    env = os.environ.get("PDEPTH_BLAS_MODE", "").lower()
    if env in ("fma", "separate"):
        _host_blas = BLAS_FMA if env == "fma" else BLAS_SEPARATE
        return _host_blas
    import numpy as np
    g = torch.Generator().manual_seed(7)
    A = (torch.randn(3, 3, generator=g) * 100).float()
    Bm = torch.randn(3, 4096, generator=g).float()
    C = A.matmul(Bm).numpy()
    a, b = A.numpy().astype(np.float64), Bm.numpy().astype(np.float64)
    f32 = np.float32
    p = [(a[:, k:k + 1] * b[k:k + 1, :]) for k in range(3)]          # exact products (float64)
    fma = ((p[0].astype(f32).astype(np.float64) + p[1]).astype(f32).astype(np.float64) + p[2]).astype(f32)
    sep = ((p[0].astype(f32) + p[1].astype(f32)).astype(f32) + p[2].astype(f32)).astype(f32)
    bad_fma, bad_sep = int((fma != C).sum()), int((sep != C).sum())
    if bad_fma == 0 or bad_fma < bad_sep:
        _host_blas = BLAS_FMA
    else:
        _host_blas = BLAS_SEPARATE
    return _host_blas


def _no_autograd(who, *tensors):
    """The ctypes kernels are invisible to autograd: refuse inputs that would silently lose (or corrupt) gradients."""
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        raise RuntimeError(f"{who}: this HIP op has no backward; call it under torch.no_grad() "
                           "(inference only, like the reference's evaluation loop)")


def load():
    """Load libpdepth_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # experiments only (tools/variants_cells.sh): PDEPTH_LIB points at a library built with other kernel knobs
    path = os.environ.get("PDEPTH_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C probabilistic-depth_amd/csrc)")
    lib = ctypes.CDLL(path)
    lib.pdepth_abi_version.restype = c_int
    lib.pdepth_last_error.restype = c_char_p
    lib.pdepth_sweep_workspace_bytes.restype = c_size_t
    lib.pdepth_sweep_workspace_bytes.argtypes = [POINTER(SweepDesc)]
    lib.pdepth_sweep_cost_f32.argtypes = [POINTER(SweepDesc), POINTER(Camera), c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_size_t, c_void_p]
    lib.pdepth_sweep_dpv_f32.argtypes = [POINTER(SweepDesc), POINTER(Camera), c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.pdepth_pack_source_f32.argtypes = [POINTER(SweepDesc), c_void_p, c_void_p, c_size_t, c_void_p]
    lib.pdepth_sweep_dpv_packed_f32.argtypes = [POINTER(SweepDesc), POINTER(Camera), c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.pdepth_dpv_reduce_f32.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                          c_void_p, c_void_p]
    lib.pdepth_dpv_reduce_ex_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.pdepth_ufield_workspace_bytes.restype = c_size_t
    lib.pdepth_ufield_workspace_bytes.argtypes = [c_int32, c_int32, c_int32]
    lib.pdepth_ufield_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                      c_float, c_float, c_float, c_float, c_int32, c_float, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_void_p]
    lib.pdepth_dpv_expect_f32.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                          c_void_p, c_void_p]
    lib.pdepth_dpv_moments_f32.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                           c_void_p, c_void_p, c_void_p]
    lib.pdepth_warp_feature_f32.argtypes = [POINTER(SweepDesc), POINTER(Camera), c_void_p, c_void_p, c_void_p,
                                            c_void_p]
    lib.pdepth_sample_coords_f32.argtypes = [POINTER(SweepDesc), POINTER(Camera), c_void_p, c_void_p, c_void_p,
                                             c_void_p]
    lib.pdepth_dpv_fuse_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                        c_float, c_float, c_void_p, c_void_p, c_void_p]
    lib.pdepth_correlation_forward_f32.argtypes = [c_void_p, c_void_p] + [c_int32] * 10 + [c_void_p, c_void_p]
    lib.pdepth_correlation_backward_f32.argtypes = [c_void_p] * 3 + [c_int32] * 10 + [c_void_p] * 3
    lib.pdepth_correlation_forward_f16.argtypes = [c_void_p, c_void_p] + [c_int32] * 10 + [c_void_p, c_void_p]
    lib.pdepth_correlation_backward_f16.argtypes = [c_void_p] * 3 + [c_int32] * 10 + [c_void_p] * 3
    lib.pdepth_correlation_output_size.argtypes = [c_int32] * 7 + [POINTER(c_int32)] * 3
    lib.pdepth_inverse_warp_f32.argtypes = [c_void_p] * 4 + [c_int32] * 5 + [c_void_p] * 3
    lib.pdepth_inverse_warp_backward_f32.argtypes = [c_void_p] * 5 + [c_int32] * 5 + [c_void_p] * 3
    for fn in ("pdepth_sweep_cost_f32", "pdepth_sweep_dpv_f32", "pdepth_dpv_reduce_f32",
               "pdepth_dpv_expect_f32", "pdepth_warp_feature_f32", "pdepth_sample_coords_f32",
               "pdepth_dpv_fuse_f32", "pdepth_correlation_forward_f32", "pdepth_inverse_warp_f32",
               "pdepth_dpv_moments_f32", "pdepth_correlation_backward_f32", "pdepth_correlation_output_size",
               "pdepth_correlation_forward_f16", "pdepth_correlation_backward_f16"):
        getattr(lib, fn).restype = c_int
    lib.pdepth_sweep_centres_source.restype = c_int
    lib.pdepth_sweep_centres_source.argtypes = [POINTER(SweepDesc)]
    lib.pdepth_sweep_source_layout.restype = c_int
    lib.pdepth_sweep_source_layout.argtypes = [POINTER(SweepDesc)]
    if lib.pdepth_abi_version() != 6:
        raise RuntimeError("libpdepth_hip.so ABI version mismatch")
    _lib = lib
    return lib


class UnsupportedShape(RuntimeError):
    """The packed entry points do not take this shape: callers use the plain NCHW path instead (every other failure of a
    native call -- launch errors, out of memory -- is a plain RuntimeError and must propagate)."""


def _check(rc, lib):
    if rc != 0:
        msg = lib.pdepth_last_error().decode()
        if rc == 1 and "does not run on a packed source" in msg:   # PDEPTH_E_ARG of the packing entry points for such a shape
            raise UnsupportedShape(msg)
        raise RuntimeError(msg)


def _dev(t: torch.Tensor, name: str) -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the HIP path needs a device tensor (got {t.device}); there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev) -> int:
    """hipStream_t of torch's current stream on `dev` (the raw getter where this torch has it: the Stream object costs 4 us)."""
    if _raw_stream is not None and dev.index is not None:
        return _raw_stream(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


class _on_device:
    """torch.cuda.device(dev) only when another device is current (the context manager costs more than a small sweep's launch)."""

    def __init__(self, dev):
        self.ctx = None if dev.index is None or torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


_desc_cache = {}   # (B, V, C, D, H, W, metric, algo) -> (workspace bytes, staging layout): two library calls saved per sweep


def _workspace_and_layout(lib, desc):
    key = (desc.B, desc.V, desc.C, desc.D, desc.H, desc.W, desc.metric, desc.algo)
    hit = _desc_cache.get(key)
    if hit is None:
        hit = (lib.pdepth_sweep_workspace_bytes(ctypes.byref(desc)), lib.pdepth_sweep_source_layout(ctypes.byref(desc)))
        if len(_desc_cache) < 4096:
            _desc_cache[key] = hit
    return hit


def _inner_contiguous(t: torch.Tensor, n_inner: int) -> bool:
    """True if the last n_inner dims are laid out densely (row-major)."""
    exp = 1
    for size, stride in zip(reversed(t.shape[-n_inner:]), reversed(t.stride()[-n_inner:])):
        if size != 1 and stride != exp:
            return False
        exp *= size
    return True


def _camera(K, R, t, rays, cxcy, B, V, HW):
    for nm, x, shape in (("K", K, (B, 3, 3)), ("R", R, (B, V, 3, 3)), ("t", t, (B, V, 3)),
                         ("rays", rays, (B, 3, HW)), ("cxcy", cxcy, (B, 2))):
        if tuple(x.shape) != shape:
            raise RuntimeError(f"{nm}: expected shape {shape}, got {tuple(x.shape)}")
    K, R, t, rays, cxcy = (x.contiguous() for x in (K, R, t, rays, cxcy))
    cam = Camera(_dev(K, "K"), _dev(R, "R"), _dev(t, "t"), _dev(rays, "rays"), _dev(cxcy, "cxcy"))
    return cam, (K, R, t, rays, cxcy)  # keep the contiguous copies alive


def _blas(blas_mode):
    return host_blas_mode() if blas_mode is None else int(blas_mode)


def selected_kernel(B, V, C, D, H, W, metric=METRIC_L2, algo=ALGO_AUTO):
    """Name of the sweep kernel family a descriptor selects ('dist' | 'corr' | 'tiled' | 'direct'): pdepth_sweep_source_layout."""
    lib = load()
    desc = SweepDesc(B, V, C, D, H, W, int(metric), int(algo), BLAS_FMA, 1.0, C * H * W, V * C * H * W, C * H * W)
    return {LAYOUT_DIST16: "dist", LAYOUT_C4_CENTRED: "corr", LAYOUT_C4: "tiled"}.get(lib.pdepth_sweep_source_layout(ctypes.byref(desc)), "direct")


class PackedSource:
    """Source views [B,V,C,H,W] in the sweep kernels' staging layout (pdepth_pack_source_f32): a workspace to hand to
    sweep() in place of src, for callers that sweep the same source features more than once or write them once per
    frame.  Owns its device memory; do not use it from two streams at once."""

    def __init__(self, ws, shape, layout=LAYOUT_C4):
        self.ws, self.shape = ws, tuple(shape)   # shape = (B, V, C, H, W)
        self.layout = int(layout)                # LAYOUT_* : the kernel family it was packed for (pdepth_sweep_source_layout)

    @property
    def centred(self):
        """The channel means were subtracted (pdepth_sweep_centres_source)."""
        return self.layout in (LAYOUT_C4_CENTRED, LAYOUT_DIST16)


def pack_source(src, n_planes=64, algo=ALGO_AUTO, metric=METRIC_L2):
    """src [B,V,C,H,W] fp32 device tensor -> PackedSource (n_planes and algo only select the kernel that will sweep it, like
    desc.D and desc.algo: the layout the correlation-form kernel takes is mean-centred, the LDS-tiled kernel's is not)."""
    lib = load()
    _no_autograd("pack_source", src)
    _dev(src, "src")
    if src.dim() != 5:
        raise RuntimeError("pack_source: src must be [B,V,C,H,W]")
    B, V, C, H, W = src.shape
    if not _inner_contiguous(src, 3) or (V > 1 and src.stride(1) < C * H * W):
        src = src.contiguous()
    desc = SweepDesc(B, V, C, int(n_planes), H, W, int(metric), int(algo), BLAS_FMA, 1.0, C * H * W,
                     src.stride(0) if B > 1 else V * C * H * W, src.stride(1) if V > 1 else C * H * W)
    ws_bytes = lib.pdepth_sweep_workspace_bytes(ctypes.byref(desc))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=src.device)
    with torch.cuda.device(src.device):
        rc = lib.pdepth_pack_source_f32(ctypes.byref(desc), _dev(src, "src"), ws.data_ptr(), ws_bytes, _stream(src.device))
    _check(rc, lib)
    return PackedSource(ws, (B, V, C, H, W), lib.pdepth_sweep_source_layout(ctypes.byref(desc)))


def pack_views(feat, rgb, n_views, n_planes=64):
    """The encoder epilogue in one pass (pdepth_pack_views_f32): feat [B*V1, Cf, h, w] (encoder output, view V1-1 of every
    item = the reference view), rgb [B*V1, 3, H, W] -> (PackedSource of the V1-1 source views with Cf+3 channels -- the
    pooled image appended like models/models.py:518-520 --, reference-view features [B, Cf+3, h, w]).  Raises for shapes
    the packed sweep does not take (callers fall back to cat + sweep)."""
    lib = load()
    _no_autograd("pack_views", feat, rgb)
    _dev(feat, "feat"), _dev(rgb, "rgb")
    if feat.dim() != 4 or rgb.dim() != 4 or feat.shape[0] != rgb.shape[0] or rgb.shape[1] != 3 or feat.shape[0] % n_views:
        raise RuntimeError("pack_views: feat [B*V1,Cf,h,w] and rgb [B*V1,3,H,W] expected")
    feat, rgb = feat.contiguous(), rgb.contiguous()
    N, Cf, h, w = feat.shape
    rate = rgb.shape[3] // w if w else 0
    if rate < 1 or rgb.shape[2] != h * rate or rgb.shape[3] != w * rate or n_views < 2:
        raise UnsupportedShape("pack_views: the image must be an exact integral multiple of the feature map "
                               "(got image %dx%d for a %dx%d map), with at least one source view"
                               % (rgb.shape[2], rgb.shape[3], h, w))
    B, V, C = N // n_views, n_views - 1, Cf + 3
    desc = SweepDesc(B, V, C, int(n_planes), h, w, METRIC_L2, ALGO_AUTO, BLAS_FMA, 1.0, C * h * w, V * C * h * w, C * h * w)
    ws_bytes = lib.pdepth_sweep_workspace_bytes(ctypes.byref(desc))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=feat.device)
    ref = torch.empty((B, C, h, w), dtype=torch.float32, device=feat.device)
    lib.pdepth_pack_views_f32.argtypes = [POINTER(SweepDesc), c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.pdepth_pack_views_f32.restype = c_int
    with torch.cuda.device(feat.device):
        rc = lib.pdepth_pack_views_f32(ctypes.byref(desc), feat.data_ptr(), rgb.data_ptr(), rate, ref.data_ptr(), ws.data_ptr(),
                                       ws_bytes, _stream(feat.device))
    _check(rc, lib)
    return PackedSource(ws, (B, V, C, h, w), lib.pdepth_sweep_source_layout(ctypes.byref(desc))), ref


def sweep(ref, src, K, R, t, rays, cxcy, d_candi, sigma, metric=METRIC_L2, algo=ALGO_AUTO,
          want_cost=True, want_logp=False, want_depth=False, blas_mode=None):
    """Batched plane sweep (+ optional fused DPV reduction).

    ref [B,C,H,W], src [B,V,C,H,W] (batch/view strides free, inner C,H,W dense) or a PackedSource, K [B,3,3],
    R [B,V,3,3], t [B,V,3], rays [B,3,HW], cxcy [B,2], d_candi [D] -- fp32 device tensors.
    Returns (cost | None, logp | None, depth | None).
    """
    lib = load()
    packed = src if isinstance(src, PackedSource) else None
    _no_autograd("sweep", ref, None if packed else src)
    _dev(ref, "ref")
    if ref.dim() != 4:
        raise RuntimeError("sweep: ref must be [B,C,H,W] and src [B,V,C,H,W]")
    B, C, H, W = ref.shape
    if packed:
        if (packed.shape[0], packed.shape[2], packed.shape[3], packed.shape[4]) != (B, C, H, W):
            raise RuntimeError(f"sweep: packed source {packed.shape} does not match ref {tuple(ref.shape)}")
        if packed.ws.device != ref.device:
            raise RuntimeError("sweep: packed source lives on another device")
        V = packed.shape[1]
    else:
        _dev(src, "src")
        if src.dim() != 5:
            raise RuntimeError("sweep: ref must be [B,C,H,W] and src [B,V,C,H,W]")
        V = src.shape[1]
        if tuple(src.shape) != (B, V, C, H, W):
            raise RuntimeError(f"sweep: src shape {tuple(src.shape)} does not match ref {tuple(ref.shape)}")
        if not _inner_contiguous(src, 3) or (V > 1 and src.stride(1) < C * H * W) or (B > 1 and src.stride(0) < 0):
            src = src.contiguous()   # (e.g. an expanded view: view stride 0)
    if not _inner_contiguous(ref, 3):
        ref = ref.contiguous()
    d_candi = d_candi.contiguous()
    D = d_candi.numel()
    dev = ref.device
    cam, keep = _camera(K, R, t, rays, cxcy, B, V, H * W)
    desc = SweepDesc(B, V, C, D, H, W, int(metric), int(algo), _blas(blas_mode), float(sigma),
                     ref.stride(0) if B > 1 else C * H * W,
                     (src.stride(0) if B > 1 else V * C * H * W) if not packed else V * C * H * W,
                     (src.stride(1) if V > 1 else C * H * W) if not packed else C * H * W)
    ws_bytes, layout = _workspace_and_layout(lib, desc)
    if packed:
        if ws_bytes == 0 or packed.ws.numel() < ws_bytes:
            raise RuntimeError("sweep: this shape / algorithm does not run on a packed source")
        if layout != packed.layout:
            raise RuntimeError("sweep: the source was packed for another kernel family (layout %d, centred: %s); pack it with "
                               "the algo / n_planes / metric it will be swept with" % (packed.layout, packed.centred))
        ws = packed.ws
    else:
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev) if ws_bytes else None
    cost = torch.empty((B, D, H, W), dtype=torch.float32, device=dev) if want_cost else None
    logp = torch.empty((B, D, H, W), dtype=torch.float32, device=dev) if want_logp else None
    depth = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_depth else None
    with _on_device(dev):
        if packed:
            rc = lib.pdepth_sweep_dpv_packed_f32(
                ctypes.byref(desc), ctypes.byref(cam), _dev(ref, "ref"), _dev(d_candi, "d_candi"),
                cost.data_ptr() if want_cost else None, logp.data_ptr() if want_logp else None,
                depth.data_ptr() if want_depth else None, ws.data_ptr(), ws_bytes, _stream(dev))
        else:
            rc = lib.pdepth_sweep_dpv_f32(
                ctypes.byref(desc), ctypes.byref(cam), _dev(ref, "ref"), _dev(src, "src"), _dev(d_candi, "d_candi"),
                cost.data_ptr() if want_cost else None, logp.data_ptr() if want_logp else None,
                depth.data_ptr() if want_depth else None, ws.data_ptr() if ws is not None else None, ws_bytes,
                _stream(dev))
    _check(rc, lib)
    del keep
    global _last_workspace
    _last_workspace = ws  # diagnostics only: tile flags of the last sweep (see fallback_tiles())
    return cost, logp, depth


_last_workspace = None


def _queue_slot(B, H, W, slot):
    n = B * ((W + 15) // 16) * ((H + 3) // 4)
    flag_only = (4 * n + 255) & ~255
    return int(_last_workspace[flag_only + 4 * slot: flag_only + 4 * slot + 4].view(torch.int32).item())


def noncentred_guard(B, H, W):
    """Diagnostics: did the pre-pass of the last sweep on a NOT centred source (LDS-tiled kernel) find channel offsets larger
    than the spread of the features -- the tiled kernel then evaluated every plane directly (csrc/sweep_pack.hip)."""
    return None if _last_workspace is None else _queue_slot(B, H, W, 51) != 0


def fallback_tiles(B, H, W, gather_flag=1):
    """Diagnostics: how much of the last sweep left the fast path.  After the distance-form kernel (ALGO_AUTO / 'dist' on its
    shapes) or the correlation-form kernel ('corr'): passes over a block of 16 pixels they evaluated directly (the kernel
    that ran zeroes its own count per call; the other's count stays at what its last call left); after the LDS-tiled
    kernel: 16x4 tiles left to the gather kernel (flag 1; lab builds: the cell-list path flags 1 = redone by its generic
    kernel, 2 = gather kernel)."""
    if _last_workspace is None:
        return 0
    n = B * ((W + 15) // 16) * ((H + 3) // 4)
    layout = _queue_slot(B, H, W, 56)
    if layout == LAYOUT_DIST16:   # (nonce << 20) | count; a count tagged by another call's nonce is stale (kernels.hpp: DIST_NONCE_SLOT)
        tagged, nonce = _queue_slot(B, H, W, 59), _queue_slot(B, H, W, 60)
        direct = (tagged & 0xFFFFF) if nonce != 0 and (tagged >> 20) == nonce else 0
    else:
        direct = _queue_slot(B, H, W, 54) if layout == LAYOUT_C4_CENTRED else 0
    return int((_last_workspace[: 4 * n].view(torch.int32) == gather_flag).sum().item()) + direct


def dpv_reduce(logits, d_candi, want_logp=True, want_depth=True, inplace=False):
    """logits [B,D,H,W] -> (logp [B,D,H,W] | None, depth [B,H,W] | None)."""
    lib = load()
    _no_autograd("dpv_reduce", logits)   # (in place it would also overwrite a tensor that carries a grad_fn)
    _dev(logits, "logits")
    if logits.dim() != 4:
        raise RuntimeError("dpv_reduce: logits must be [B,D,H,W]")
    logits = logits.contiguous()
    B, D, H, W = logits.shape
    d_candi = d_candi.contiguous()
    if d_candi.numel() != D:
        raise RuntimeError(f"dpv_reduce: d_candi has {d_candi.numel()} entries, volume has D={D}")
    dev = logits.device
    logp = (logits if inplace else torch.empty_like(logits)) if want_logp else None
    depth = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_depth else None
    with torch.cuda.device(dev):
        rc = lib.pdepth_dpv_reduce_f32(_dev(logits, "logits"), _dev(d_candi, "d_candi"), B, D, H, W,
                                       logp.data_ptr() if want_logp else None,
                                       depth.data_ptr() if want_depth else None, _stream(dev))
    _check(rc, lib)
    return logp, depth


def dpv_reduce_ex(logits, d_candi, addend=None, want_logp=True, want_prob=False, want_depth=False, want_var=False,
                  want_quarter=False, inplace=False):
    """(logits [+ addend]) [B,D,H,W] -> dict of the requested outputs: logp, prob = exp(logp) [B,D,H,W], depth, var
    [B,H,W], quarter = nearest quarter-resolution logp [B,D,H//4,W//4] -- one pass (pdepth_dpv_reduce_ex_f32)."""
    lib = load()
    _no_autograd("dpv_reduce_ex", logits, addend)
    _dev(logits, "logits")
    if logits.dim() != 4:
        raise RuntimeError("dpv_reduce_ex: logits must be [B,D,H,W]")
    logits = logits.contiguous()
    B, D, H, W = logits.shape
    if addend is not None:
        _dev(addend, "addend")
        if tuple(addend.shape) != (B, D, H, W):
            raise RuntimeError(f"dpv_reduce_ex: addend shape {tuple(addend.shape)} != logits {tuple(logits.shape)}")
        addend = addend.contiguous()
    d_candi = d_candi.contiguous()
    if d_candi.numel() != D:
        raise RuntimeError(f"dpv_reduce_ex: d_candi has {d_candi.numel()} entries, volume has D={D}")
    dev = logits.device
    out = {}
    if want_logp:
        out["logp"] = logits if inplace else torch.empty_like(logits)
    if want_prob:
        out["prob"] = torch.empty_like(logits)
    if want_depth:
        out["depth"] = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    if want_var:
        out["var"] = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    if want_quarter:
        out["quarter"] = torch.empty((B, D, H // 4, W // 4), dtype=torch.float32, device=dev)
    ptr = lambda k: out[k].data_ptr() if k in out else None
    with torch.cuda.device(dev):
        rc = lib.pdepth_dpv_reduce_ex_f32(_dev(logits, "logits"), addend.data_ptr() if addend is not None else None,
                                          _dev(d_candi, "d_candi"), B, D, H, W, ptr("logp"), ptr("prob"), ptr("depth"),
                                          ptr("var"), ptr("quarter"), _stream(dev))
    _check(rc, lib)
    return out


def ufield(dpv, d_candi, intr, mask, bv_log, unc_ang, z_start, z_end, min_depth, quash, d_sum=None):
    """dpv [B,D,H,W], intr [B,3,3], mask [B,H,W] | None -> (plane [B,D,W], depth_zero [B,H,W])  (pdepth_ufield_f32)."""
    lib = load()
    _no_autograd("ufield", dpv)
    _dev(dpv, "dpv"), _dev(intr, "intr")
    if dpv.dim() != 4:
        raise RuntimeError("ufield: dpv must be [B,D,H,W]")
    dpv = dpv.contiguous()
    B, D, H, W = dpv.shape
    intr = intr.contiguous().float()
    if tuple(intr.shape) != (B, 3, 3):
        raise RuntimeError(f"ufield: intr must be [{B},3,3], got {tuple(intr.shape)}")
    if mask is not None:
        _dev(mask, "mask")
        if tuple(mask.shape) != (B, H, W):
            raise RuntimeError(f"ufield: mask must be [{B},{H},{W}], got {tuple(mask.shape)}")
        mask = mask.contiguous().float()
    d_candi = d_candi.contiguous()
    # depth of rows shifted in from outside the image: dpv_to_depthmap of the zero padding (exp(0) = 1 per plane).  d_sum = that
    # sum formed by the caller on the host (ops.ufield); a caller that only has the candidates on the device pays one read-back.
    oob = (float(d_sum) if d_sum is not None else float(d_candi.sum().item())) if bv_log else 0.0
    dev = dpv.device
    ws_bytes = lib.pdepth_ufield_workspace_bytes(B, H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    plane = torch.empty((B, D, W), dtype=torch.float32, device=dev)
    dzero = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pdepth_ufield_f32(_dev(dpv, "dpv"), _dev(d_candi, "d_candi"), intr.data_ptr(),
                                   mask.data_ptr() if mask is not None else None, B, D, H, W, 1 if bv_log else 0,
                                   float(unc_ang), float(np.float32(z_start)), float(np.float32(z_end)), float(min_depth),
                                   1 if quash else 0, oob, plane.data_ptr(), dzero.data_ptr(), ws.data_ptr(), ws_bytes,
                                   _stream(dev))
    _check(rc, lib)
    return plane, dzero


def dpv_expect(dpv, d_candi, bv_log):
    """dpv [B,D,H,W] -> depth [B,H,W] = sum_k d_k * (exp(dpv) if bv_log else dpv)."""
    lib = load()
    _no_autograd("dpv_expect", dpv)
    _dev(dpv, "dpv")
    if dpv.dim() != 4:
        raise RuntimeError("dpv_expect: dpv must be [B,D,H,W]")
    dpv = dpv.contiguous()
    B, D, H, W = dpv.shape
    d_candi = d_candi.contiguous()
    if d_candi.numel() != D:
        raise RuntimeError(f"dpv_expect: d_candi has {d_candi.numel()} entries, volume has D={D}")
    depth = torch.empty((B, H, W), dtype=torch.float32, device=dpv.device)
    with torch.cuda.device(dpv.device):
        rc = lib.pdepth_dpv_expect_f32(_dev(dpv, "dpv"), _dev(d_candi, "d_candi"), B, D, H, W, int(bool(bv_log)),
                                       depth.data_ptr(), _stream(dpv.device))
    _check(rc, lib)
    return depth


def dpv_moments(dpv, d_candi, bv_log=True):
    """dpv [B,D,H,W] -> (mean [B,H,W], variance [B,H,W]) of the depth distribution."""
    lib = load()
    _no_autograd("dpv_moments", dpv)
    _dev(dpv, "dpv")
    if dpv.dim() != 4:
        raise RuntimeError("dpv_moments: dpv must be [B,D,H,W]")
    dpv = dpv.contiguous()
    B, D, H, W = dpv.shape
    d_candi = d_candi.contiguous()
    if d_candi.numel() != D:
        raise RuntimeError(f"dpv_moments: d_candi has {d_candi.numel()} entries, volume has D={D}")
    mean = torch.empty((B, H, W), dtype=torch.float32, device=dpv.device)
    var = torch.empty_like(mean)
    with torch.cuda.device(dpv.device):
        rc = lib.pdepth_dpv_moments_f32(_dev(dpv, "dpv"), _dev(d_candi, "d_candi"), B, D, H, W, int(bool(bv_log)),
                                        mean.data_ptr(), var.data_ptr(), _stream(dpv.device))
    _check(rc, lib)
    return mean, var


def warp_feature(src, K, R, t, rays, cxcy, d_candi, blas_mode=None):
    """src [B,V,D,H,W] -> out [B,V,D,H,W], channel i warped with depth plane i."""
    lib = load()
    _no_autograd("warp_feature", src)
    _dev(src, "src")
    if src.dim() != 5:
        raise RuntimeError("warp_feature: src must be [B,V,D,H,W]")
    B, V, C, H, W = src.shape
    if not _inner_contiguous(src, 3) or (V > 1 and src.stride(1) < C * H * W) or (B > 1 and src.stride(0) < 0):
        src = src.contiguous()   # (e.g. an expanded view: view stride 0)
    d_candi = d_candi.contiguous()
    D = d_candi.numel()
    cam, keep = _camera(K, R, t, rays, cxcy, B, V, H * W)
    desc = SweepDesc(B, V, C, D, H, W, 0, 0, _blas(blas_mode), 1.0, 0, src.stride(0) if B > 1 else V * C * H * W,
                     src.stride(1) if V > 1 else C * H * W)
    out = torch.empty((B, V, D, H, W), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        rc = lib.pdepth_warp_feature_f32(ctypes.byref(desc), ctypes.byref(cam), _dev(src, "src"),
                                         _dev(d_candi, "d_candi"), out.data_ptr(), _stream(src.device))
    _check(rc, lib)
    del keep
    return out


def sample_coords(K, R, t, rays, cxcy, d_candi, H, W, blas_mode=None, algo=ALGO_AUTO):
    """Diagnostic: (ix, iy) [B,V,D,H,W] sample positions handed to the bilinear sampler.

    algo selects the arithmetic of the sweep kernel of the same name: ALGO_AUTO = the shared-reciprocal divide
    chain of the LDS-tiled kernel, ALGO_DIRECT = the compiler's IEEE divides of the gather kernel."""
    lib = load()
    _dev(K, "K")
    B, V = R.shape[0], R.shape[1]
    d_candi = d_candi.contiguous()
    D = d_candi.numel()
    cam, keep = _camera(K, R, t, rays, cxcy, B, V, H * W)
    desc = SweepDesc(B, V, 1, D, H, W, 0, int(algo), _blas(blas_mode), 1.0, 0, 0, H * W)
    ix = torch.empty((B, V, D, H, W), dtype=torch.float32, device=K.device)
    iy = torch.empty_like(ix)
    with torch.cuda.device(K.device):
        rc = lib.pdepth_sample_coords_f32(ctypes.byref(desc), ctypes.byref(cam), _dev(d_candi, "d_candi"),
                                          ix.data_ptr(), iy.data_ptr(), _stream(K.device))
    _check(rc, lib)
    del keep
    return ix, iy


def dpv_fuse(logp, dmaps, masks, d_candi, var, eps, want_fused=True, want_log=True):
    """logp [B,D,H,W], dmaps [B,H,W], masks [B,H,W] -> (fused | None, log fused | None)."""
    lib = load()
    _no_autograd("dpv_fuse", logp)
    _dev(logp, "logp")
    logp, dmaps, masks, d_candi = (t.contiguous() for t in (logp, dmaps, masks, d_candi))
    B, D, H, W = logp.shape
    if tuple(dmaps.shape) != (B, H, W) or tuple(masks.shape) != (B, H, W) or d_candi.numel() != D:
        raise RuntimeError("dpv_fuse: dmaps/masks must be [B,H,W] and d_candi [D]")
    fused = torch.empty_like(logp) if want_fused else None
    logf = torch.empty_like(logp) if want_log else None
    with torch.cuda.device(logp.device):
        rc = lib.pdepth_dpv_fuse_f32(_dev(logp, "logp"), _dev(dmaps, "dmaps"), _dev(masks, "masks"),
                                     _dev(d_candi, "d_candi"), B, D, H, W, float(var), float(eps),
                                     fused.data_ptr() if want_fused else None, logf.data_ptr() if want_log else None,
                                     _stream(logp.device))
    _check(rc, lib)
    return fused, logf


def _corr_out_shape(lib, H, W, pad_size, kernel_size, max_displacement, stride1, stride2):
    oc, oh, ow = c_int32(), c_int32(), c_int32()
    rc = lib.pdepth_correlation_output_size(H, W, int(pad_size), int(kernel_size), int(max_displacement), int(stride1), int(stride2),
                                            ctypes.byref(oc), ctypes.byref(oh), ctypes.byref(ow))
    _check(rc, lib)
    return oc.value, oh.value, ow.value


def _corr_dtype(x1, x2):
    if x1.dtype != x2.dtype or x1.dtype not in (torch.float32, torch.float16):
        raise RuntimeError("correlation: inputs must both be float32 or both float16 (got %s, %s)" % (x1.dtype, x2.dtype))
    return x1.dtype == torch.float16


def _dev_any(t, name):
    """Device + contiguity check for tensors that may be fp16 (the fp32-only _dev() guards everything else)."""
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise RuntimeError("%s must be a CUDA/HIP tensor (the HIP path has no CPU fallback)" % name)
    if torch.is_grad_enabled() and t.requires_grad:
        raise RuntimeError("%s requires grad: the HIP kernels are invisible to autograd (use ops.correlation / torch.no_grad())" % name)
    return t


def correlation_forward(x1, x2, pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply=1):
    """x1, x2 [B,C,H,W] fp32 or fp16 -> [B,(2*(d/s2)+1)^2,oH,oW] of the same dtype (mean over the kernel window and C of
    shifted products; every configuration of the reference's kernel: include/pdepth.h)."""
    lib = load()
    _dev_any(x1, "input1"), _dev_any(x2, "input2")
    if x1.shape != x2.shape or x1.dim() != 4:
        raise RuntimeError("correlation: inputs must be two [B,C,H,W] tensors of the same shape")
    half = _corr_dtype(x1, x2)
    x1, x2 = x1.contiguous(), x2.contiguous()
    B, C, H, W = x1.shape
    oc, oh, ow = _corr_out_shape(lib, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    out = torch.empty((B, oc, oh, ow), dtype=x1.dtype, device=x1.device)
    fn = lib.pdepth_correlation_forward_f16 if half else lib.pdepth_correlation_forward_f32
    with torch.cuda.device(x1.device):
        rc = fn(x1.data_ptr(), x2.data_ptr(), B, C, H, W, int(pad_size), int(kernel_size), int(max_displacement), int(stride1),
                int(stride2), int(corr_multiply), out.data_ptr(), _stream(x1.device))
    _check(rc, lib)
    return out


def correlation_backward(x1, x2, grad_out, pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply=1,
                         want1=True, want2=True):
    """x1, x2 [B,C,H,W], grad_out [B,(2r+1)^2,oH,oW] (fp32 or fp16, all alike) -> (grad_x1 | None, grad_x2 | None)."""
    lib = load()
    _dev_any(x1, "input1"), _dev_any(x2, "input2"), _dev_any(grad_out, "grad_output")
    half = _corr_dtype(x1, x2)
    x1, x2, grad_out = x1.contiguous(), x2.contiguous(), grad_out.contiguous().to(x1.dtype)
    B, C, H, W = x1.shape
    oc, oh, ow = _corr_out_shape(lib, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    if x2.shape != x1.shape or tuple(grad_out.shape) != (B, oc, oh, ow):
        raise RuntimeError("correlation_backward: shape mismatch")
    g1 = torch.empty_like(x1) if want1 else None
    g2 = torch.empty_like(x2) if want2 else None
    fn = lib.pdepth_correlation_backward_f16 if half else lib.pdepth_correlation_backward_f32
    with torch.cuda.device(x1.device):
        rc = fn(x1.data_ptr(), x2.data_ptr(), grad_out.data_ptr(), B, C, H, W, int(pad_size), int(kernel_size),
                int(max_displacement), int(stride1), int(stride2), int(corr_multiply),
                g1.data_ptr() if want1 else None, g2.data_ptr() if want2 else None, _stream(x1.device))
    _check(rc, lib)
    return g1, g2


SAMPLE_MODES = {"bilinear": 0, "nearest": 1}


def inverse_warp(img, depth, Kinv, proj, mode="bilinear"):
    """img [B,C,H,W], depth [B,H,W], Kinv [B,3,3], proj [B,3,4] -> (warped [B,C,H,W], valid bool [B,H,W])."""
    lib = load()
    _dev(img, "img")
    img, depth, Kinv, proj = (t.contiguous() for t in (img, depth, Kinv, proj))
    B, C, H, W = img.shape
    if tuple(depth.shape) != (B, H, W) or tuple(Kinv.shape) != (B, 3, 3) or tuple(proj.shape) != (B, 3, 4):
        raise RuntimeError("inverse_warp: expected depth [B,H,W], Kinv [B,3,3], proj [B,3,4]")
    out = torch.empty_like(img)
    valid = torch.empty((B, H, W), dtype=torch.uint8, device=img.device)
    with torch.cuda.device(img.device):
        rc = lib.pdepth_inverse_warp_f32(_dev(img, "img"), _dev(depth, "depth"), _dev(Kinv, "Kinv"), _dev(proj, "proj"),
                                         B, C, H, W, SAMPLE_MODES[mode], out.data_ptr(), valid.data_ptr(), _stream(img.device))
    _check(rc, lib)
    return out, valid.bool()


def inverse_warp_backward(img, depth, Kinv, proj, grad_out, mode="bilinear", want_img=True, want_point=True):
    """-> (grad_img [B,C,H,W] | None, grad_point [B,3,H,W] | None): pdepth_inverse_warp_backward_f32."""
    lib = load()
    img, depth, Kinv, proj, grad_out = (t.contiguous() for t in (img, depth, Kinv, proj, grad_out))
    B, C, H, W = img.shape
    if tuple(grad_out.shape) != (B, C, H, W):
        raise RuntimeError("inverse_warp_backward: grad_out must have the shape of img")
    g_img = torch.empty_like(img) if want_img else None
    g_pt = torch.empty((B, 3, H, W), dtype=torch.float32, device=img.device) if want_point else None
    with torch.cuda.device(img.device):
        rc = lib.pdepth_inverse_warp_backward_f32(_dev(img, "img"), _dev(depth, "depth"), _dev(Kinv, "Kinv"), _dev(proj, "proj"),
                                                  _dev(grad_out, "grad_out"), B, C, H, W, SAMPLE_MODES[mode],
                                                  g_img.data_ptr() if want_img else None,
                                                  g_pt.data_ptr() if want_point else None, _stream(img.device))
    _check(rc, lib)
    return g_img, g_pt
