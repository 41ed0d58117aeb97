"""Instruction mix per basic block of a kernel's asm dump: python tools/asm_blocks.py FILE FIRST_LINE LAST_LINE"""
import re, sys
f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lines = open(f).read().splitlines()[a - 1:b]
cnt = dict(valu=0, salu=0, lds=0, vmem=0, wait=0)
def flush(tag):
    n = sum(cnt.values())
    if n:
        print(f"   {n:4d}  " + " ".join(f"{k} {v}" for k, v in cnt.items() if v) + (f"   {tag}" if tag else ""))
    for k in cnt: cnt[k] = 0
for i, ln in enumerate(lines):
    t = ln.strip()
    if not t or t.startswith(";"): continue
    if t.startswith(".LBB"):
        flush(""); print(f"{a + i}: {t.split()[0]}"); continue
    op = t.split()[0]
    if op.startswith(("s_cbranch", "s_branch")):
        cnt["salu"] += 1; flush("-> " + t); continue
    if op.startswith("v_"): cnt["valu"] += 1
    elif op.startswith(("s_waitcnt", "s_nop")): cnt["wait"] += 1
    elif op.startswith("s_"): cnt["salu"] += 1
    elif op.startswith("ds_"): cnt["lds"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): cnt["vmem"] += 1
flush("")
