"""CPU suite: bench.py's N > 1 reporting path.  Two gloo ranks gather their metric vectors exactly as bench.py does
and rank 0 assembles the JSON line with the function bench.py uses (pdist.assemble_bench_line)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pdepth_amd  # noqa: F401
from pdepth_amd import dist as pdist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank(rank, world, port, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = pdist.init_from_env(backend="gloo")
    lo, hi = pdist.shard_range(4 * w, r, w)                 # weak scaling: 4 volumes per rank
    pdist.barrier()
    wall = pdist.max_over_ranks(0.010 * (1 + r), torch.device("cpu"))     # rank 1 is the slower one: 20 ms
    vec = torch.tensor([hi - lo, 0.5 + 0.25 * r, 21.5, 1.0], dtype=torch.float32)
    allm = pdist.gather_metrics(vec)
    if r == 0:
        line = pdist.assemble_bench_line(allm, wall, steps=10, warmup=2, batch_per_gpu=4, world=w, metric="m", unit="u",
                                         workload="w", bytes_per_volume=1000, hbm_peak_gbs=8000.0,
                                         extras={"roofline": {"traffic": 123}, "gather_fallback_tiles": 0})
        json.dump(line, open(out_path, "w"))
    pdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_bench_line(tmp_path):
    out = str(tmp_path / "line.json")
    mp.spawn(_rank, args=(2, _free_port(), out), nprocs=2, join=True)
    line = json.load(open(out))
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["config"]["parallelism"] == "dp2"
    assert line["value"] == pytest.approx(8 * 10 / 0.020)          # all ranks' volumes / slowest rank's wall
    assert line["ms_per_step"] == pytest.approx(2.0)
    assert line["per_rank_kernel_ms"] == pytest.approx([0.5, 0.75])
    assert line["roofline"]["achieved"] == pytest.approx(4 * 1000 / 0.5e-3 / 1e9)   # rank 0's launch
    assert line["roofline"]["traffic"] == 123 and line["scaling"] == "weak" and line["vs_baseline"] is None
    json.dumps(line)


def test_assemble_rejects_a_wrong_world_size():
    with pytest.raises(ValueError):
        pdist.assemble_bench_line(torch.zeros(1, 4), 1.0, steps=1, warmup=0, batch_per_gpu=4, world=2, metric="m",
                                  unit="u", workload="w", bytes_per_volume=1, hbm_peak_gbs=1.0)


def test_self_launch_starts_ranks_before_touching_the_gpu():
    """`python bench.py --gpus 2` with no launcher must start the ranks itself and fail non-zero when they fail: here
    (no GPU) every rank dies on the GPU assertion, after torch.distributed.run has started them with WORLD_SIZE=2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=280)
    if torch.cuda.device_count() >= 2:   # a multi-GPU host: the two ranks run and rank 0 prints the line
        assert p.returncode == 0, p.stderr[-2000:]
        assert json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 2
        return
    assert p.returncode != 0
    if not torch.cuda.is_available():
        assert "bench.py needs a GPU" in p.stderr


def _rank8(rank, world, port, out_path, global_batch, tag):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = pdist.init_from_env(backend="gloo")
    lo, hi = pdist.shard_range(global_batch, r, w)          # the slice of the global batch this rank owns
    pdist.barrier()
    wall = pdist.max_over_ranks(0.004 + 0.0001 * r, torch.device("cpu"))
    vec = torch.tensor([hi - lo, 0.40 + 0.01 * r, float(lo), 1.0], dtype=torch.float32)
    allm = pdist.gather_metrics(vec)
    if r == 0:
        line = pdist.assemble_bench_line(allm, wall, steps=10, warmup=5, batch_per_gpu=global_batch // w, world=w, metric="m", unit="u",
                                         workload=tag, bytes_per_volume=104333312, hbm_peak_gbs=8000.0)
        line["first_items"] = [float(x) for x in allm[:, 2]]
        json.dump(line, open(out_path, "w"))
    pdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("global_batch,tag", [(32, "configs[2]: default_stereo, batch 32 over 8 GPUs"),
                                               (16, "configs[4]: D=128 512x1024 V=4, batch 16 over 8 GPUs")])
def test_eight_rank_shards_of_configs_3_and_5(tmp_path, global_batch, tag):
    """the partition and the arithmetic of the line at the world size the driver's scaling run uses
    (`python bench.py --gpus 8 --batch 4 --pose stereo`, `--gpus 8 --batch 2 --planes 128 --height 512 --width 1024 --views 4`;
    train.py:88-95 is the reference's one-process-per-GPU spawn).  gloo on the CPU: 8 ranks, contiguous slices, one all_gather."""
    out = str(tmp_path / "line8.json")
    mp.spawn(_rank8, args=(8, _free_port(), out, global_batch, tag), nprocs=8, join=True)
    line = json.load(open(out))
    per = global_batch // 8
    assert line["n_gpus"] == 8 and line["config"]["global_batch"] == global_batch and line["config"]["per_gpu_batch"] == per
    assert line["config"]["parallelism"] == "dp8" and line["scaling"] == "weak"
    assert len(line["per_rank_kernel_ms"]) == 8 and line["per_rank_kernel_ms"][7] == pytest.approx(0.47)
    assert line["first_items"] == [float(per * r) for r in range(8)]          # contiguous, disjoint, covering
    assert line["value"] == pytest.approx(global_batch * 10 / 0.0047)         # every rank's volumes / the slowest rank's wall
    assert line["roofline"]["achieved"] == pytest.approx(per * 104333312 / 0.40e-3 / 1e9)   # rank 0's own launch
