"""Static instruction census of one kernel of an ISA listing, per phase mark (build with -DDIST_MARKS):
   hipcc ... -DDIST_MARKS -S --cuda-device-only sweep_dist.hip -o x.s ; python tools/dbg/isa_census.py x.s <kernel substring>"""
import sys, re, collections
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if key in l and re.match(r"^\S+:\s*(;.*)?$", l) and not l.startswith(" ") and not l.startswith("\t"))
phase = "pre"; cnt = collections.defaultdict(collections.Counter); order = []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"): break
    m = re.match(r"; MARK (\d+)", t)
    if m:
        phase = "after mark " + m.group(1)
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
    op = t.split()[0]
    kind = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
            "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "scratch_", "flat_")) else "other")
    if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"): kind = "lanexfer"
    if op.startswith("s_cbranch") or op == "s_branch": kind = "branch"
    if op in ("s_waitcnt", "s_nop"): kind = op
    if phase not in order: order.append(phase)
    cnt[phase][kind] += 1
kinds = ["valu", "salu", "lanexfer", "branch", "s_waitcnt", "s_nop", "lds", "vmem", "mfma", "other"]
print("%-16s" % "phase" + "".join("%10s" % k for k in kinds))
tot = collections.Counter()
for ph in order:
    c = cnt[ph]; tot.update(c)
    print("%-16s" % ph + "".join("%10d" % c[k] for k in kinds))
print("%-16s" % "total" + "".join("%10d" % tot[k] for k in kinds))
