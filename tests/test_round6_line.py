"""CPU suite: the committed bench line of round 6 (profiles/r06_bench_line.json, `python bench.py --steps 20 --warmup 5` on a
gpurun box) carries what VERDICT r5 asked of it: the two multi-GPU configs at one GPU's shard, the model-real figures, the
packed entry, a dtype that says what the products are."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_of_round_6():
    line = json.loads(open(os.path.join(REPO, "profiles", "r06_bench_line.json")).read().strip().splitlines()[-1])
    assert line["metric"].startswith("depth-volumes/sec") and line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5
    assert line["roofline"]["sweep_kernel"] == "dist" and 0.12 < line["roofline"]["frac"] < 1 and line["ms_per_step"] < 0.42
    assert "fp16-pair products on MFMA" in line["dtype"] and line["config"]["workload"].startswith("BASELINE configs[1]")
    w = line["workloads"]
    for key, kernel in (("cfg3_shard", "dist"), ("cfg5_share", "dist")):
        row = w[key]
        assert row["ms_per_call"] > 0 and 0 < row["frac"] < 1 and row["sweep_kernel"] == kernel and row["volumes_per_s"] > 0, key
        assert row["max_abs_depth_diff_vs_gather"] <= 1.5e-4, key      # (two GPU kernels against each other)
    assert "configs[2]" in w["cfg3_shard"]["workload"] and "configs[4]" in w["cfg5_share"]["workload"]
    assert w["cfg5_share"]["ms_per_call"] / w["cfg5_share"]["B"] < 2.3
    # the routed workload: every pixel block of every item left to the gather kernel, whose answer it then is
    sm = w["cfg2_smooth"]   # smooth features at unit variance: the fast form, within the north star of the gather kernel
    assert sm["direct_passes"] == 0 and sm["max_abs_depth_diff_vs_gather"] <= 1e-4 and sm["ms_per_call"] < 1.3 * line["ms_per_step"]
    r = w["cfg2_routed"]
    assert r["direct_passes"] == r["B"] * 256 * 512 // 16 and r["max_abs_depth_diff_vs_gather"] == 0.0 and r["ms_per_call"] > line["ms_per_step"]
    for key in ("B1_nchw", "B1_packed", "B4_nchw", "B4_packed"):
        row = line["model_real"][key]
        assert row["us_per_call"] > 0 and row["launches"] in (1, 4) and 0 < row["frac"] < 1, key
    assert line["packed_entry"]["kernel_ms"] < 0.34 and line["packed_entry"]["max_abs_depth_diff_vs_headline"] <= 1e-4
    assert line["preflight"]["max_abs_depth_diff_vs_gather_kernel"] <= 1e-4
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["max_abs_depth_diff_gpu_vs_port_item0"] <= 1e-4
    assert line["roofline"]["traffic"] is None or line["roofline"]["traffic"] > line["roofline"]["algorithmic_bytes_per_launch"]
