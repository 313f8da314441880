"""Round-3 golden fixture g19: the reference's uncertainty-field metric glue, compute_unc_field and compute_unc_rmse
(utils/img_utils.py:178-202; caller trainer/default_trainer.py:243-247), run on seeded volumes through both dataset
branches of gen_ufield (cfg.data.dataset_path naming kitti / ilim).  Build container only (imports /root/reference):

    python tests/golden/make_golden_r3b.py

Data only: inputs (log-DPV, validity mask, intrinsics, depth candidates) and the reference's outputs.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _import_reference  # noqa: E402,F401
from make_golden_r2 import peaked_logdpv  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    homo, view, img_utils = _import_reference()
    import pdepth_amd
    from pdepth_amd import _native, synth as S
    from util_host import cpu_vendor
    meta = dict(meta_torch=torch.__version__, meta_cpu_capability=torch.backends.cpu.get_cpu_capability(),
                meta_cpu_vendor=cpu_vendor(), meta_blas_mode=np.int32(_native.host_blas_mode()))
    g = torch.Generator().manual_seed(1919)
    D, H, W = 32, 48, 64
    d_candi = img_utils.powerf(5.0, 40.0, D, 1.0)
    intr = torch.tensor([[[90.0, 0.0, 31.3], [0.0, 85.0, 18.6], [0.0, 0.0, 1.0]]])   # [1,3,3]: the trainer's batch of one
    # (torch 2.10 rejects the stray positional arguments of Tensor.repeat([D, 1, 1], 0, 1), utils/img_utils.py:342: dropped
    #  for the duration of the calls, as in make_golden_r2.py -- the reference file is not touched)
    orig_repeat = torch.Tensor.repeat
    torch.Tensor.repeat = lambda self, *a: orig_repeat(self, a[0]) if a and isinstance(a[0], (list, tuple)) else orig_repeat(self, *a)
    out = {}
    try:
        for tag, path in (("kitti", "/data/kitti/raw"), ("ilim", "/data/ilim/set1")):
            cfg = S.Cfg({"data": {"dataset_path": path}})
            pred = peaked_logdpv(g, D, H, W, d_candi)                     # the network's refined output: a log-DPV
            truth = torch.exp(peaked_logdpv(g, D, H, W, d_candi, spread=0.8))   # the ground-truth DPV: probabilities
            mask = (torch.rand(1, H, W, generator=g) > 0.25).float()
            uf_t, uf_p, dbg = img_utils.compute_unc_field(pred, truth, d_candi, intr, mask, cfg)
            err = img_utils.compute_unc_rmse(uf_t.clone(), uf_p.clone(), d_candi)
            out.update({f"{tag}_pred_logdpv": pred.numpy(), f"{tag}_truth_dpv": truth.numpy(), f"{tag}_mask": mask.numpy(),
                        f"{tag}_field_truth": uf_t.numpy(), f"{tag}_field_pred": uf_p.numpy(), f"{tag}_debugmap": dbg.numpy(),
                        f"{tag}_rmse": np.float64(float(err)), f"{tag}_path": path})
            cols = int((~torch.isnan(uf_t).any(1) & ~torch.isnan(uf_p).any(1)).sum())
            assert cols > W // 3, f"degenerate fixture {tag}: {cols} usable columns"
            print(tag, "usable columns", cols, "error", float(err))
    finally:
        torch.Tensor.repeat = orig_repeat
    np.savez_compressed(os.path.join(HERE, "g19_unc_field.npz"), d_candi=d_candi, intr=intr.numpy(), **out, **meta)
    print("g19_unc_field.npz", os.path.getsize(os.path.join(HERE, "g19_unc_field.npz")), "bytes")


if __name__ == "__main__":
    main()
