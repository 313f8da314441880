"""ctypes loader for oracle/sweep_ref.c (fp64 "truth at the fp32 sample positions").

Test infrastructure only -- see oracle/ref_cpu.py for who may import this.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpdepth_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def sweep_dpv_f64(ref, src, K, R, t, rays, cx, cy, d_candi, sigma, metric="L2", blas_separate=0):
    """numpy fp32 inputs of ONE item -> (cost [D,H,W], logp [D,H,W], depth [H,W]) float64."""
    lib = load()
    ref = np.ascontiguousarray(ref, np.float32)
    src = np.ascontiguousarray(src, np.float32)
    C, H, W = ref.shape
    V = src.shape[0]
    d32 = np.ascontiguousarray(np.asarray(d_candi).astype(np.float32))
    D = d32.shape[0]
    cost = np.empty((D, H, W)); logp = np.empty((D, H, W)); depth = np.empty((H, W)); scratch = np.empty(D)
    K = np.ascontiguousarray(K, np.float32); R = np.ascontiguousarray(R, np.float32)
    t = np.ascontiguousarray(t, np.float32); rays = np.ascontiguousarray(rays, np.float32)
    lib.pdo_sweep_dpv_f64.restype = None
    lib.pdo_sweep_dpv_f64(_fp(ref), _fp(src), _fp(K), _fp(R), _fp(t), _fp(rays), ctypes.c_float(float(cx)),
                          ctypes.c_float(float(cy)), _fp(d32), V, C, D, H, W, ctypes.c_double(float(sigma)),
                          0 if metric == "L2" else 1, int(blas_separate), _fp(cost), _fp(logp), _fp(depth), _fp(scratch))
    return cost, logp, depth
