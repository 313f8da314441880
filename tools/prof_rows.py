"""One op of the path, N times, for a rocprofv3 kernel trace (tools/prof_rows.sh): python3 tools/prof_rows.py <case> [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth

case, N = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20


def sweep(B, D, H, W, V, pose, algo="auto"):
    b = synth.make_batch(2, B, C=67, D=D, H=H, W=W, V=V, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    return lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)


B, D, H, W = 4, 64, 256, 512
x = torch.randn(B, D, H, W, device="cuda")
dc = ops.d_candi_tensor(synth.powerf(5, 40, D, 1.0), "cuda")
if case == "cfg5":
    f = sweep(2, 128, 512, 1024, 4, "mono")
elif case == "cfg5_packed":   # config 5 with the source views already in the staging layout (the encoder epilogue writes it)
    b_ = synth.make_batch(2, 2, C=67, D=128, H=512, W=1024, V=4, pose="mono")
    d_ = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b_.items()}
    ps_ = ops.pack_source(d_["src"], 128)
    dc_ = ops.d_candi_tensor(d_["d_candi"], "cuda")
    f = lambda: ops.sweep_dpv(d_["ref"], ps_, d_["K"], d_["R"], d_["t"], d_["rays"], d_["cxcy"], dc_, 10.0)
elif case == "cfg3":
    f = sweep(4, 64, 256, 512, 1, "stereo")
elif case == "cfg2_tiled":
    f = sweep(4, 64, 256, 512, 1, "mono", "tiled2")
elif case == "model_real_packed":   # BASELINE config 1 as the host model runs it: the sources already in the staging layout
    b_ = synth.make_batch(2, 1, C=67, D=64, H=64, W=128, V=1, pose="mono")
    d_ = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b_.items()}
    ps_ = ops.pack_source(d_["src"], 64)
    dc_ = ops.d_candi_tensor(d_["d_candi"], "cuda")
    f = lambda: ops.sweep_dpv(d_["ref"], ps_, d_["K"], d_["R"], d_["t"], d_["rays"], d_["cxcy"], dc_, 10.0)
elif case == "model_real":
    f = sweep(4, 64, 64, 128, 1, "mono")
elif case == "reduce_ex":
    y = torch.randn(B, D, H, W, device="cuda")
    f = lambda: (ops.dpv_reduce_ex(x, dc, want_logp=True, want_depth=True, want_var=True, want_quarter=True),
                 ops.dpv_reduce_ex(x, dc, addend=y, want_logp=True, want_prob=True))
elif case == "dpv_fuse":
    lp = torch.log_softmax(x, dim=1)
    mk = (torch.rand(B, 1, H, W, device="cuda") > 0.6).float()
    dm = (torch.rand(B, H, W, device="cuda") * 30 + 6) * mk[:, 0]
    f = lambda: ops.dpv_fuse(lp, dm, mk, dc, 0.3)
elif case == "ufield":
    lp = torch.log_softmax(x, dim=1)
    intr = torch.tensor([[0.58 * W, 0, W / 2.0, 0, 0.58 * W, H / 2.0, 0, 0, 1]], device="cuda").repeat(B, 1).reshape(B, 3, 3)
    f = lambda: ops.ufield(lp, dc, intr, None, BV_log=True)
elif case == "correlation":
    x1, x2 = torch.randn(4, 64, 64, 128, device="cuda"), torch.randn(4, 64, 64, 128, device="cuda")
    go = torch.randn(4, 81, 64, 128, device="cuda")
    f = lambda: (pdepth_amd._native.correlation_forward(x1, x2, 4, 1, 4, 1, 1, 1), pdepth_amd._native.correlation_backward(x1, x2, go, 4, 1, 4, 1, 1, 1))
elif case == "correlation_general":
    x1, x2 = torch.randn(4, 64, 64, 128, device="cuda"), torch.randn(4, 64, 64, 128, device="cuda")
    f = lambda: (pdepth_amd._native.correlation_forward(x1, x2, 5, 3, 4, 2, 3, 1), pdepth_amd._native.correlation_forward(x1.half(), x2.half(), 4, 1, 4, 1, 1, 1))
elif case == "pack_views_small":   # the encoder epilogue at the model-real size (B=1: one source view + the reference view of a 256x512 frame)
    feat, rgb = torch.randn(2, 64, 64, 128, device="cuda"), torch.rand(2, 3, 256, 512, device="cuda")
    f = lambda: ops.pack_views(feat, rgb, 2, 64)
elif case == "pack_views":
    feat, rgb = torch.randn(8, 64, 256, 512, device="cuda"), torch.rand(8, 3, 1024, 2048, device="cuda")
    f = lambda: ops.pack_views(feat, rgb, 2, 64)
else:
    raise SystemExit("unknown case " + case)
for _ in range(3):
    f()
torch.cuda.synchronize()
for _ in range(N):
    f()
torch.cuda.synchronize()
