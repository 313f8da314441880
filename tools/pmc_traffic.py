"""profiles/pmc_traffic.json from the summaries of tools/prof.sh (the counters bench.py attaches to its roofline block).

    python tools/pmc_traffic.py <mono summary.txt> <stereo summary.txt> <tag of the committed copies, e.g. r03>
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


def record(path, pose, chosen, source):
    c = parse(path)
    kernels, tot = {}, {"hbm": 0.0, "valu": 0.0, "salu": 0.0, "lds": 0.0}
    for k, v in c.items():
        if not k.startswith(("pdepth::", "void pdepth::")):
            continue
        n = short(k)
        if "sweep_direct" in n:
            n = "pdepth::sweep_direct_kernel<0, 68, false> (no tile handed over)"
            if n in kernels:
                continue   # (one instantiation per entry: the headline call launches the first)
        elif ("sweep_mfma" in n) != (chosen == "mfma") and ("sweep_mfma" in n or "sweep_tiled" in n):
            n += " (not chosen: leaves at once)"
        kernels[n] = {"FETCH_SIZE_KB": v.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB": v.get("WRITE_SIZE", 0.0),
                      "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU", 0.0), "SQ_INSTS_SALU": v.get("SQ_INSTS_SALU", 0.0),
                      "SQ_INSTS_LDS": v.get("SQ_INSTS_LDS", 0.0)}
        if "clear_and_pick" in n:
            continue   # (packed-source entry only)
        if "sweep_direct" in n:
            # the mean per dispatch mixes the empty launches of a step (no tile handed over: ~10 KB) with bench.py's one preflight run
            # of the whole gather kernel: not part of a step, left out of the per-launch totals
            kernels[n]["note"] = "mean over the steps' empty launches AND bench.py's preflight run of the full kernel; not in the totals"
            continue
        tot["hbm"] += (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
        tot["valu"] += v.get("SQ_INSTS_VALU", 0.0)
        tot["salu"] += v.get("SQ_INSTS_SALU", 0.0)
        tot["lds"] += v.get("SQ_INSTS_LDS", 0.0)
    return {"workload": {"batch": 4, "C": 67, "D": 64, "H": 256, "W": 512, "V": 1, "pose": pose, "kernel": chosen},
            "source": source, "kernels": kernels, "hbm_bytes_per_launch": tot["hbm"], "valu_wave_instr_per_launch": tot["valu"],
            "salu_wave_instr_per_launch": tot["salu"], "lds_wave_instr_per_launch": tot["lds"]}


if __name__ == "__main__":
    mono, stereo, tag = sys.argv[1:4]
    doc = {"correction": "gfx950: FETCH_SIZE x2 (calibrated in profiles/r01_fetch_size_calibration.txt), WRITE_SIZE exact; KB = 1024 B",
           "note": "one launch = one pdepth_sweep_dpv_f32 call = pack pre-pass (which also picks the sweep kernel on the device for this "
                   "shape class) + both sweep kernels (the one not chosen leaves at once) + the gather kernel over the (here: zero) flagged "
                   "tiles. bench.py attaches a record only when its workload AND the kernel that ran match.",
           "workloads": [
               record(mono, "mono", "tiled", "profiles/%s_auto_mono.rocprofv3.txt (tools/prof.sh: rocprofv3 --pmc passes of `python3 bench.py --steps 3 "
                      "--warmup 1 --no-cpu-baseline`; SQ_* in one pass, FETCH_SIZE and WRITE_SIZE in separate passes)" % tag),
               record(stereo, "stereo", "mfma", "profiles/%s_auto_stereo.rocprofv3.txt (tools/prof.sh ... --pose stereo)" % tag)]}
    json.dump(doc, open(os.path.join(REPO, "profiles", "pmc_traffic.json"), "w"), indent=1)
    for w in doc["workloads"]:
        print(w["workload"]["pose"], w["workload"]["kernel"], "HBM MB/launch %.1f" % (w["hbm_bytes_per_launch"] / 1e6), "VALU %.3e" % w["valu_wave_instr_per_launch"])
