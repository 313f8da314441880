"""Shared helpers for the parity tests (oracle = test infrastructure, see oracle/ref_cpu.py)."""
import os

import numpy as np
import torch

from oracle import ref_cpu as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star: depth maps within 1e-4 abs of the reference PyTorch CPU path
DEPTH_ATOL = 1e-4


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def same_cpu_as_golden(g):
    """True when this host runs the same torch build on the same CPU family as the fixture's host
    (then the oracle must be BIT-identical; elsewhere MKL/vector-width drift is tolerated)."""
    from util_host import cpu_vendor
    return (str(g["meta_cpu_capability"]) == torch.backends.cpu.get_cpu_capability()
            and str(g["meta_torch"]) == torch.__version__ and str(g["meta_cpu_vendor"]) == cpu_vendor())


def golden_blas(g):
    """'fma' | 'separate': BLAS rounding of the host that generated fixture g."""
    return "separate" if int(g["meta_blas_mode"]) == 1 else "fma"


def oracle_item(it, metric="L2", sigma=10.0):
    """cost, logp, depth of one synth item via the CPU oracle."""
    K = it["K"]
    return O.sweep_dpv(it["ref"][None], it["src"][None], it["d_candi"], it["R"], it["t"], K, it["rays"],
                       K.numpy()[0, 2], K.numpy()[1, 2], sigma, metric)


def oracle_batch(batch, metric="L2", sigma=10.0):
    outs = []
    for b in range(batch["ref"].shape[0]):
        it = {k: (v[b] if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        outs.append(oracle_item(it, metric, sigma))
    return tuple(torch.cat([o[i] for o in outs], dim=0) for i in range(3))


def to_dev(batch, dev="cuda"):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def exact_item(it, sigma=10.0):
    """float64 (cost, depth, kappa) of one synth item at the reference's float32 sample positions (O.sweep_dpv_exact64)."""
    K = it["K"]
    return O.sweep_dpv_exact64(it["ref"][None], it["src"][None], it["d_candi"], it["R"], it["t"], K, it["rays"],
                               K.numpy()[0, 2], K.numpy()[1, 2], sigma)


def exact_batch(batch, sigma=10.0):
    outs = []
    for b in range(batch["ref"].shape[0]):
        it = {k: (v[b] if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        outs.append(exact_item(it, sigma))
    return tuple(torch.cat([o[i] for o in outs], dim=0) for i in range(3))


# (Diagnostics since round 6: no test relaxes the north-star bound any more -- the default kernel routes ill-conditioned items to
#  the gather kernel, tests/test_soak_regressions.py asserts the plain 1e-4 m and records these figures next to it.)
# How a kernel that does NOT copy the reference's summation order compares with the reference where the depth is
# ill-conditioned (features with means of several sigma: costs of hundreds, float32 rounding noise of 1e-4 in them, and a
# softmax that turns a unit of cost into up to (d_max - d_min) / 2 metres).  Measured on the soak's worst cases
# (profiles/r05_soak_summary.txt): the float32 reference itself is up to 3.2e-4 m from the exact evaluation of its own
# formula -- no implementation can be within 1e-4 m of it there unless it rounds like it.  So, with the exact volume X:
#   noise:      max |cost - X| / max |oracle - X|  and  the ratio of the rms errors (measured for the distance form: 2.2 / 1.5)
#   explained:  per pixel |depth - oracle depth| <= DEPTH_ATOL + kappa (max_k |cost - X| + max_k |oracle - X|): the
#               north-star bound plus what the two cost volumes' own errors at that pixel account for, to first order --
#               nothing is left for the softmax / expectation of the kernel to have added


def noise_and_explained(cost, depth, ocost, odepth, xcost, xkappa):
    """-> dict(noise_max_ratio, noise_rms_ratio, unexplained_m): see above; tensors on the CPU, [B,D,h,w] / [B,h,w]."""
    fin = torch.isfinite(xcost)
    assert torch.equal(torch.isfinite(cost), fin) and torch.equal(torch.isfinite(ocost), fin)
    ea = torch.where(fin, (cost.double() - xcost).abs(), torch.zeros_like(xcost))
    eo = torch.where(fin, (ocost.double() - xcost).abs(), torch.zeros_like(xcost))
    n = max(int(fin.sum()), 1)
    out = {"noise_max_ratio": float(ea.max() / eo.max().clamp_min(1e-30)),
           "noise_rms_ratio": float(((ea ** 2).sum() / n).sqrt() / ((eo ** 2).sum() / n).sqrt().clamp_min(1e-30)),
           "oracle_cost_noise_max": float(eo.max()), "cost_noise_max": float(ea.max())}
    dfin = torch.isfinite(odepth)
    allowed = DEPTH_ATOL + xkappa * (ea.max(1).values + eo.max(1).values)
    over = torch.where(dfin, (depth.double() - odepth.double()).abs() - allowed, torch.full_like(allowed, -1.0))
    out["unexplained_m"] = float(over.max().clamp_min(0.0))
    out["kappa_max"] = float(xkappa[dfin].max()) if bool(dfin.any()) else 0.0
    return out
