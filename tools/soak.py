"""Soak test on the GPU box: random shapes / poses / depth candidates / metrics, every implementation forced in turn
(the distance-form and the correlation-form kernels where they apply -- L2, D <= 128, C <= 72, V <= 8 --, both builds of the
tiled kernel; with SOAK_DPV every fourth case the `auto` selection), against the gather kernel, which evaluates in the
reference's op order (channel sums in ATen's cascade order since round 5).

    python tools/soak.py <seed> <cases> [case,case,...] [algo]

    SOAK_DPV=1   also the fused outputs (log-DPV, expected depth) and, every fifth case, the packed-source entry
    SOAK_OFFSET=1  every other case with per-channel offsets of up to 8 standard deviations on reference and source features
    SOAK_ORACLE=1  a leg against the CPU ORACLE (oracle.sweep_cost_at, its per-pixel form) at 160 sampled pixels per case:
                   cost, and the depth of the fused output, against the north-star bound (1e-4 m, scaled with the candidates)
    SOAK_SPEC=1  C = 67, D = 64, V = 1 (the compile-time specialised instantiation) at random sizes up to 300 x 560
    third / fourth argument: replay only these case numbers of the seed (the generator is advanced through the others
    without building them), optionally with one implementation forced; a failing case then prints where it differs
    PDEPTH_LIB + PDEPTH_LAX=1: run against the library of an older commit (bisecting)

About 1 500 cases per minute.  Round 2: ~45 000 cases, worst cost difference 6e-7 of the largest cost of a volume.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("PDEPTH_LAX"):   # tolerate missing symbols / an older ABI number of a library built from an older commit
    import ctypes
    _CDLL = ctypes.CDLL

    class _Dummy:
        argtypes = None
        restype = None

        def __call__(self, *a):
            return 2

    class _Lax:
        def __init__(self, *a, **k):
            object.__setattr__(self, "_l", _CDLL(*a, **k))

        def __getattr__(self, n):
            if n == "pdepth_abi_version":
                return _Dummy()
            try:
                return getattr(self._l, n)
            except AttributeError:
                return _Dummy()
    ctypes.CDLL = _Lax
import numpy as np  # noqa: E402
import torch  # noqa: E402
import pdepth_amd  # noqa: E402,F401
from pdepth_amd import _native, ops, synth  # noqa: E402

DEV = torch.device("cuda")
SPEC = bool(os.environ.get("SOAK_SPEC"))
DPV = bool(os.environ.get("SOAK_DPV"))
OFFSET = bool(os.environ.get("SOAK_OFFSET"))
ORACLE = bool(os.environ.get("SOAK_ORACLE"))
POSES = ("mono", "stereo", "wide", "identity")


def draw_case(rng, case, build):
    """The random draws of one case, always in the same order (a seed names its cases).  build=False consumes them only."""
    H, W = int(rng.integers(2, 200)), int(rng.integers(2, 400))
    C, D, V = int(rng.integers(1, 72)), int(rng.integers(1, 161)), int(rng.integers(1, 4))
    B = int(rng.integers(1, 3))
    if SPEC:
        C, D, V = 67, 64, 1
        H, W = int(rng.integers(40, 300)), int(rng.integers(100, 560))
    pose = POSES[int(rng.integers(0, 4))]
    cxo, cyo = float(rng.uniform(-3, 3)), float(rng.uniform(-2, 2))
    k = int(rng.integers(0, 6))   # 1: rolled + shifted view 0, 2: view 0 up to 30 m away, 3 / 4: unordered / descending candidates
    roll = shift = far = cand = None
    if k == 1:
        roll, shift = rng.uniform(-0.3, 0.3), rng.uniform(-2.5, 2.5, size=3)
    elif k == 2:
        far = rng.uniform(-30, 30, size=3)
    elif k in (3, 4):
        cand = rng.uniform(0.5, 60.0, size=D)
    shape = dict(H=H, W=W, C=C, D=D, V=V, B=B, pose=pose, k=k)
    if not build:
        return shape, None
    b = synth.make_batch(5000 + case, B, C=C, D=D, H=H, W=W, V=V, pose=pose, cx_off=cxo, cy_off=cyo)
    if k == 1:
        cz, sz = np.cos(roll), np.sin(roll)
        b["R"][0, 0] = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=torch.float32) @ b["R"][0, 0]
        b["t"][0, 0] = torch.from_numpy(shift.astype(np.float32))
    elif k == 2:
        b["t"][0, 0] = torch.from_numpy(far.astype(np.float32))
    elif k == 3:
        b["d_candi"] = cand
    elif k == 4:
        b["d_candi"] = np.sort(cand)[::-1].copy()
    if OFFSET and case % 2 == 0:   # (an own generator: the draws of a seed's cases stay what they were)
        g = torch.Generator().manual_seed(9000 + case)
        mu = (torch.rand(C, generator=g) * 2 - 1) * 8.0
        b["ref"] = b["ref"] + mu[None, :, None, None]
        b["src"] = b["src"] + mu[None, None, :, None, None]
    return shape, b


def replay_case(seed, case, spec=False, offset=True):
    """Case `case` of `seed` as a soak run with SOAK_SPEC=spec, SOAK_OFFSET=offset drew it (the generator is advanced
    through the earlier cases): tests/test_soak_regressions.py pins the worst cases of earlier rounds' soaks with this."""
    global SPEC, OFFSET
    keep = SPEC, OFFSET
    SPEC, OFFSET = spec, offset
    try:
        rng = np.random.default_rng(seed)
        for c in range(case):
            draw_case(rng, c, build=False)
        return draw_case(rng, case, build=True)
    finally:
        SPEC, OFFSET = keep


def describe(case, algo, s, metric):
    return f"case {case} {algo} {s['pose']} {s['H']}x{s['W']} C={s['C']} D={s['D']} V={s['V']} B={s['B']} k={s['k']} {metric}"


def where(ca, cd, fin):
    """Replay mode: which planes / tiles differ."""
    bad = (np.abs(ca - cd) > 1e-5 * max(1.0, float(np.abs(cd[fin]).max()))) & fin
    bb, kk, yy, xx = np.nonzero(bad)
    print("  bad elements", bad.sum(), "of", bad.size, "planes", np.unique(kk)[:40], "rows", yy.min(), yy.max(), "cols", xx.min(), xx.max())
    tiles = sorted(set(zip((yy // 4).tolist(), (xx // 16).tolist())))
    print("  16x4 tiles touched", len(tiles), tiles[:24])
    for (ty, tx) in tiles[:3]:
        sub = bad[0, :, ty * 4:ty * 4 + 4, tx * 16:tx * 16 + 16]
        print("   tile", ty, tx, "bad per plane", sub.reshape(sub.shape[0], -1).sum(1).tolist())


def oracle_leg(b, d, args, algo, tag, rng):
    """Item 0 of the case at 160 sampled pixels through the oracle's per-pixel form: cost and fused depth of `algo`."""
    from oracle import ref_cpu as O
    H, W, D = b["ref"].shape[2], b["ref"].shape[3], len(b["d_candi"])
    idx = torch.from_numpy(np.unique(rng.integers(0, H * W, 160))).long()
    K = b["K"][0]
    ocost = O.sweep_cost_at(b["ref"][0:1], b["src"][0:1], b["d_candi"], b["R"][0], b["t"][0], K, b["rays"][0], K.numpy()[0, 2], K.numpy()[1, 2],
                            8.0, idx)                                                     # [1, D, n]
    odepth = O.dpv_to_depthmap(O.log_dpv(ocost.reshape(1, D, 1, -1)), b["d_candi"], BV_log=True).reshape(-1)
    cost, _, depth = ops.sweep_dpv(*args, feat_dist="L2", algo=algo, want_cost=True)
    c_at = cost[0].reshape(D, H * W)[:, idx.to(DEV)].cpu()
    d_at = depth[0].reshape(H * W)[idx.to(DEV)].cpu()
    fin = torch.isfinite(ocost[0])
    if not torch.equal(torch.isfinite(c_at), fin):
        print(tag, "ORACLE: finiteness of the cost differs")
        return 0.0
    cerr = float(((c_at - ocost[0]).abs() - 2e-5 * ocost[0].abs())[fin].max()) if bool(fin.any()) else 0.0
    if cerr > 2e-4:
        print(tag, "ORACLE: cost differs by", cerr, "beyond 2e-5 relative")
    dfin = torch.isfinite(odepth)
    if not bool(dfin.any()):
        return 0.0
    scale = max(1.0, float(np.max(np.abs(b["d_candi"]))) / 40.0)
    derr = float((d_at - odepth)[dfin].abs().max()) / scale
    if derr > 1e-4:
        print(tag, "ORACLE: depth differs by", derr * scale, "(candidates up to", scale * 40.0, "m)")
    return derr


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    only = set(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[3] else None
    force = sys.argv[4] if len(sys.argv) > 4 else None
    worst = worst_d = worst_o = 0.0
    n = fb = 0
    for case in range(cases):
        s, b = draw_case(rng, case, build=only is None or case in only)
        if case % 500 == 499 and only is None:
            print("... case", case + 1, "worst so far", worst, flush=True)   # (a GPU run that stays silent for seven minutes is taken to be hung)
        if b is None:
            continue
        d = {kk: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for kk, v in b.items()}
        metric = "L1" if case % 7 == 3 else "L2"
        algo = ("tiled1", "dist", "direct", "auto", "tiled2")[case % 5]   # (period 5: every kernel meets offsets -- even cases -- and the oracle leg -- every third)
        if algo in ("corr", "dist") and (metric == "L1" or s["D"] > 128 or s["C"] > 72):
            algo = "tiled1"
        if algo == "tiled2" and s["D"] > 64:
            algo = "tiled1"
        if algo == "auto" and metric == "L1":
            algo = "tiled1"
        if force:
            algo = force
            if force in ("corr", "dist") and (metric == "L1" or s["D"] > 128 or s["C"] > 72):
                algo = "tiled1"   # (shapes the correlation-form kernel is not built for)
        args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 8.0)
        tag = describe(case, algo, s, metric)
        if DPV:
            _, _, da = ops.sweep_dpv(*args, feat_dist=metric, algo=algo, want_cost=True)
            _, _, dd = ops.sweep_dpv(*args, feat_dist=metric, algo="direct", want_cost=True)
            if bool((torch.isfinite(dd) != torch.isfinite(da)).any()):
                print(tag, "depth finiteness differs")
            else:
                fd = torch.isfinite(dd)
                if bool(fd.any()):
                    scale = max(1.0, float(np.max(np.abs(b["d_candi"]))) / 40.0)
                    de = float((da - dd)[fd].abs().max())
                    worst_d = max(worst_d, de / scale)
                    if de > 3e-4 * scale:
                        print(tag, "depth differs by", de, "(candidates up to", scale * 40.0, "m)")
            if case % 5 == 0 and algo in ("auto", "tiled1", "tiled2") and metric == "L2" and s["C"] <= 68:
                try:
                    ps = ops.pack_source(d["src"], s["D"], "auto", metric)
                    cp, _, dp = ops.sweep_dpv(d["ref"], ps, *args[2:], feat_dist=metric, algo="auto", want_cost=True)
                    ca2, _, da2 = ops.sweep_dpv(*args, feat_dist=metric, algo="auto", want_cost=True)
                    # (round 6: the NCHW entry of the distance-form kernel takes its statistics over the source views and the
                    #  reference view, the packed entry over the source views: the two agree to rounding, not bit for bit)
                    close = lambda x, y: torch.allclose(x.nan_to_num(), y.nan_to_num(), rtol=3e-5, atol=3e-4 * max(1.0, float(np.max(np.abs(b["d_candi"]))) / 40.0))
                    if not (close(cp, ca2) and close(dp, da2) and torch.equal(torch.isnan(cp), torch.isnan(ca2))):
                        print(tag, "packed entry differs from the plain entry")
                except RuntimeError as e:   # shapes the packed entry declines
                    if "packed" not in str(e):
                        raise
        if ORACLE and metric == "L2" and case % 3 == 0:
            worst_o = max(worst_o, oracle_leg(b, d, args, algo, tag, np.random.default_rng(77 + case)))
        ca_dev = ops.sweep_cost(*args, feat_dist=metric, algo=algo)
        fb += _native.fallback_tiles(s["B"], s["H"], s["W"])
        if case % 4 == 1:   # repeatability: the same call again, bit for bit (LDS min / max tables, queues, stealing: no order dependence)
            cb_dev = ops.sweep_cost(*args, feat_dist=metric, algo=algo)
            if not torch.equal(ca_dev.nan_to_num(), cb_dev.nan_to_num()):
                print(tag, "a second call gives different bits")
        ca = ca_dev.cpu().numpy()
        cd = ops.sweep_cost(*args, feat_dist=metric, algo="direct").cpu().numpy()
        n += 1
        if not np.array_equal(np.isnan(ca), np.isnan(cd)):
            print(tag, "NaN pattern differs")
            continue
        fin = np.isfinite(cd)
        if fin.any():
            err = float(np.abs(ca - cd)[fin].max()) / max(1.0, float(np.abs(cd[fin]).max()))
            worst = max(worst, err)
            if err > 2e-6:
                print(tag, "err", err)
                if only is not None:
                    where(ca, cd, fin)
    print("cases", n, "worst", worst, "fallback tiles", fb, ("worst depth difference (scaled) %.3e" % worst_d) if DPV else "",
          ("worst depth difference from the ORACLE at the sampled pixels (scaled) %.3e" % worst_o) if ORACLE else "")


if __name__ == "__main__":
    main()
