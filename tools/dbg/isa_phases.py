"""Static instruction census of the matrix-pipe sweep kernel per phase (build with -DMFMA_MARKS, ISA listing on stdin)."""
import sys, re, collections
lines = sys.stdin.read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN6pdepth12_GLOBAL__N_117sweep_mfma_kernelILi17ELi1EE"))
phase = "pre"; cnt = collections.defaultdict(collections.Counter)
for l in lines[start:]:
    t = l.strip()
    if t.startswith("s_endpgm"): break
    m = re.match(r"; MARK (\d+)", t)
    if m: phase = "after mark " + m.group(1); continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
    op = t.split()[0]
    kind = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
            "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "scratch_", "flat_")) else "other")
    if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"): kind = "lanexfer"
    if op.startswith("s_cbranch") or op == "s_branch": kind = "branch"
    if op in ("s_waitcnt", "s_nop"): kind = op
    cnt[phase][kind] += 1
kinds = ["valu", "salu", "lanexfer", "branch", "s_waitcnt", "s_nop", "lds", "vmem", "mfma", "other"]
print("%-16s" % "phase" + "".join("%10s" % k for k in kinds))
for ph, c in cnt.items(): print("%-16s" % ph + "".join("%10d" % c[k] for k in kinds))
