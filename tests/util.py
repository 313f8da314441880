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
