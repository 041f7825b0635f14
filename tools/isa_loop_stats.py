"""Instruction mix of a kernel's basic blocks from `hipcc -S` output: python tools/isa_loop_stats.py file.s <mangled-prefix>
Prints per basic block (label): counts of MFMA / VALU / transcendental / LDS / VMEM / SALU / waitcnt / barrier instructions."""
import re, sys
src = open(sys.argv[1]).read().splitlines()
pref = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith(pref) and l.rstrip().endswith(":") or (l.startswith(pref) and ":" in l))
blocks, cur = [], ["entry", {}]
TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")
for l in src[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end") or t.startswith("s_endpgm"):
        blocks.append(cur); break
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append(cur); cur = [t.split(":")[0], {}]; continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("v_mfma"): k = "mfma"
    elif op.startswith(TRANS): k = "trans"
    elif op.startswith("v_"): k = "valu"
    elif op.startswith("ds_"): k = "lds_r" if "read" in op or "load" in op else "lds_w"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): k = "vmem"
    elif op.startswith("s_waitcnt"): k = "wait"
    elif op.startswith("s_barrier"): k = "barrier"
    elif op.startswith("s_"): k = "salu"
    else: k = "other"
    cur[1][k] = cur[1].get(k, 0) + 1
    if k == "valu":
        cur[1].setdefault("_ops", {})
        cur[1]["_ops"][op] = cur[1]["_ops"].get(op, 0) + 1
for name, c in blocks:
    tot = sum(v for k, v in c.items() if not k.startswith("_"))
    if tot < 20: continue
    print(name, {k: v for k, v in c.items() if not k.startswith("_")})
    if "-v" in sys.argv and "_ops" in c:
        print("   ", sorted(c["_ops"].items(), key=lambda kv: -kv[1])[:25])
