import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pdepth_amd
from pdepth_amd.warping import homography
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from util import golden, golden_blas
from test_hip_parity import _cam
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["g1_rot_trans"]:
    g = golden(name + ".npz")
    cost = homography.est_swp_volume_v4(torch.from_numpy(g["ref"]).to(dev), torch.from_numpy(g["src"]).to(dev),
                                        g["d_candi"], torch.from_numpy(g["R"]).to(dev), torch.from_numpy(g["t"]).to(dev),
                                        _cam(g, dev), float(g["sigma"]), feat_dist="L2", blas=golden_blas(g))
    c = cost.cpu().numpy()[0]; e = g["cost_L2"][0]
    bad = np.abs(c - e) > 2e-4 + 2e-5 * np.abs(e)
    print(name, "shape", c.shape, "bad", bad.sum(), "max", np.abs(c - e).max())
    for k in range(c.shape[0]):
        print("plane", k, "bad", bad[k].sum())
        for y in range(c.shape[1]):
            print("  ", "".join("X" if b else "." for b in bad[k, y]))
