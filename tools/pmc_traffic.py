"""profiles/pmc_traffic.json from the summaries of tools/prof.sh (the counters bench.py attaches to its roofline block).

    python tools/pmc_traffic.py <mono summary.txt> <stereo summary.txt> <tag of the committed copies, e.g. r05> <collected: commit, date> [kernel: dist]
"""
import json, os, re, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    """{kernel name prefix: {counter: mean per dispatch}} of the '== counters' sections"""
    out, kern = {}, None
    for line in open(path):
        m = re.match(r"\s+kernel (.*)$", line)
        if m:
            kern = m.group(1).strip()
            continue
        m = re.match(r"\s+(\w+)\s+mean/dispatch\s+([0-9.e+-]+)", line)
        if m and kern:
            out.setdefault(kern, {})[m.group(1)] = float(m.group(2))
    return out


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


def record(path, pose, chosen, source, collected):
    c = parse(path)
    kernels, tot = {}, {"hbm": 0.0, "valu": 0.0, "salu": 0.0, "lds": 0.0}
    for k, v in c.items():
        if not k.startswith(("pdepth::", "void pdepth::")):
            continue
        n = short(k)
        kernels[n] = {"FETCH_SIZE_KB": v.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB": v.get("WRITE_SIZE", 0.0),
                      "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU", 0.0), "SQ_INSTS_SALU": v.get("SQ_INSTS_SALU", 0.0),
                      "SQ_INSTS_LDS": v.get("SQ_INSTS_LDS", 0.0)}
        for extra in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_BUSY_CU_CYCLES",
                      "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU"):
            if extra in v:
                kernels[n][extra] = v[extra]
        if "sweep_direct" in n:
            # bench.py's one preflight run of the gather kernel (the cross-check in front of the timed region): not part of a step
            kernels[n]["note"] = ("the mean over bench.py's ONE preflight run of the whole gather kernel (the cross-check, not part of a step) and the routed launch "
                                  "of every NCHW call, whose blocks read one flag and leave (4 us, no traffic) when no item is routed: not in the totals")
            continue
        tot["hbm"] += (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
        tot["valu"] += v.get("SQ_INSTS_VALU", 0.0)
        tot["salu"] += v.get("SQ_INSTS_SALU", 0.0)
        tot["lds"] += v.get("SQ_INSTS_LDS", 0.0)
    return {"workload": {"batch": 4, "C": 67, "D": 64, "H": 256, "W": 512, "V": 1, "pose": pose, "kernel": chosen},
            "source": source, "collected": collected, "kernels": kernels, "hbm_bytes_per_launch": tot["hbm"],
            "valu_wave_instr_per_launch": tot["valu"], "salu_wave_instr_per_launch": tot["salu"], "lds_wave_instr_per_launch": tot["lds"]}


if __name__ == "__main__":
    mono, stereo, tag, collected = sys.argv[1:5]
    chosen = sys.argv[5] if len(sys.argv) > 5 else "dist"
    doc = {"correction": "gfx950: FETCH_SIZE x2 (calibrated in profiles/r01_fetch_size_calibration.txt), WRITE_SIZE exact; KB = 1024 B",
           "note": "one launch = one pdepth_sweep_dpv_f32 call (NCHW entry) = feature_stats_kernel + pack_dist_kernel (pre-pass: channel means and "
                   "scale, centred fp16 hi/lo planes + squared neighbour differences) + sweep_dist_kernel + the gather kernel's launch for routed items (empty on this workload). bench.py attaches a record only when its workload AND the kernel "
                   "that ran match; 'collected' says on which commit and when the counters were taken (another box than any later bench run).",
           "workloads": [
               record(mono, "mono", chosen, "profiles/%s_auto_mono.rocprofv3.txt (tools/prof.sh: rocprofv3 --pmc passes of `python3 bench.py --steps 3 "
                      "--warmup 1 --no-cpu-baseline`; SQ_* in one pass, FETCH_SIZE and WRITE_SIZE in separate passes)" % tag, collected),
               record(stereo, "stereo", chosen, "profiles/%s_auto_stereo.rocprofv3.txt (tools/prof.sh ... --pose stereo)" % tag, collected)]}
    json.dump(doc, open(os.path.join(REPO, "profiles", "pmc_traffic.json"), "w"), indent=1)
    for w in doc["workloads"]:
        print(w["workload"]["pose"], w["workload"]["kernel"], "HBM MB/launch %.1f" % (w["hbm_bytes_per_launch"] / 1e6), "VALU %.3e" % w["valu_wave_instr_per_launch"])
