"""Static instruction account of K3' (k_score_lds<1, 4, true>) from the compiler's ISA listing: what a wave issues per chunk outside its
steps, per full step, per last step of a class, and behind the chunk loop -- by unit (SALU / VALU / LDS / VMEM / MFMA / SMEM / other).
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -o k.s slam_toolbox_amd/csrc/matcher_kernels.hip
  python tools/k3_isa_account.py k.s > profiles/r6_k_score_isa_account.txt
Dynamic counts per launch (SQ_INSTS_* of profiles/r6_k_score_pmc.json) are set against it at the end."""
import json, os, re, sys
src = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(src) if l.startswith("_ZN2kh11k_score_ldsILi1ELi4ELb1EEEvPKhmiii:"))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end + 1]

def unit(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "SMEM"
    if op.startswith("v_"): return "VALU"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop") or op.startswith("s_setprio") or op.startswith("s_cbranch") or op.startswith("s_branch"): return "ctrl"
    if op.startswith("s_"): return "SALU"
    return None

def count(lines):
    c = {}
    for l in lines:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        u = unit(op)
        if u:
            c[u] = c.get(u, 0) + 1
    return c

def fmt(c):
    tot = sum(c.values())
    return "%4d  (" % tot + ", ".join("%s %d" % (k, c[k]) for k in ("SALU", "VALU", "LDS", "MFMA", "VMEM", "SMEM", "ctrl") if k in c) + ")"

labels = {i: l.split(":")[0] for i, l in enumerate(body) if l.startswith(".LBB")}
idx = sorted(labels)
# the chunk loop: from its header (the first label carrying "Loop Header: Depth=1" in front of the first s_barrier) to the s_setprio 0 behind the classes
hdr = next(i for i in idx if "Loop Header: Depth=1" in body[i] and any("s_barrier" in body[j] for j in range(i, i + 15)))
prio1 = next(i for i in range(hdr, len(body)) if "s_setprio 1" in body[i])
prio0 = next(i for i in range(prio1, len(body)) if "s_setprio 0" in body[i])
back = next(i for i in range(prio0, len(body)) if "s_cbranch" in body[i])
inner = [i for i in idx if prio1 < i < prio0 and "Inner Loop Header: Depth=2" in body[i + 1] + body[i]]
print("K3' k_score_lds<1, 4, true>: static instruction account of the compiled kernel (gfx950), instructions a wave issues\n")
print("in front of the chunk loop (set-up, first DMA)        ", fmt(count(body[:hdr])))
print("per chunk, head: barrier, DMA of the next region, its window offsets, the descriptor after the next", fmt(count(body[hdr:prio1])))
steps, tails, heads = [], [], []
cur = prio1
for k, i in enumerate(inner):
    loop_end = next(j for j in range(i, prio0) if "s_cbranch_scc" in body[j] and ".LBB" in body[j] and labels[i] in body[j])
    nxt = inner[k + 1] if k + 1 < len(inner) else prio0
    heads.append(count(body[cur:i]))
    steps.append(count(body[i:loop_end + 1]))
    # the tail step of the class lies between the loop's end and the next class' loop header (minus that class' own head of ~5 instructions)
    tails.append(count(body[loop_end + 1:nxt]))
    cur = nxt
for c in range(len(inner)):
    print("  class %d: in front of its steps %s" % (c, fmt(heads[c])))
    print("           one full step (4 windows) %s" % fmt(steps[c]))
    print("           its last step (1-3 windows, with the next class' set-up) %s" % fmt(tails[c]))
print("per chunk, behind the classes (rotate descriptors, loop) ", fmt(count(body[prio0:back + 1])))
print("behind the chunk loop (class merge, pose loop, maxima)  ", fmt(count(body[back + 1:])))
per_chunk_fixed = sum(count(body[hdr:prio1]).values()) + sum(sum(h.values()) for h in heads) + sum(count(body[prio0:back + 1]).values())
step_n = sum(steps[0].values())
tail_n = sum(sum(t.values()) for t in tails) / max(1, len(tails))
print("\nper chunk outside the steps: %d instructions; a full step: %d; a class' last step with its bookkeeping: %.0f on average" % (per_chunk_fixed, step_n, tail_n))
pmc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r6_k_score_pmc.json")
if os.path.exists(pmc):
    d = json.load(open(pmc))
    c = d.get("counters", d)
    def get(k):
        v = c.get(k)
        return v.get("per_launch", v) if isinstance(v, dict) else v
    keys = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
    vals = {k: get(k) for k in keys if get(k) is not None}
    if vals:
        print("\ndynamic, per launch of 51.2 matches (profiles/r6_k_score_pmc.json):", ", ".join("%s %.3g" % (k[9:], v) for k, v in vals.items()))
