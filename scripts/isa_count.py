"""Instruction counts of one kernel in a gfx950 assembly listing (hipcc --cuda-device-only -S), whole kernel and main loop:
vector ALU, scalar ALU, LDS, vector memory, branches, waits.  usage: isa_count.py <listing.s> <mangled-name substring>..."""
import re, sys
text = open(sys.argv[1]).read().splitlines()
for pat in sys.argv[2:]:
    start = next(i for i, l in enumerate(text) if pat in l and l.startswith("_Z") and ": " in l and "@" in l)
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    body = text[start:end]
    # main loop: from the first label marked "Loop Header: Depth=1" to the last line that says it is in that loop
    hdr = next((i for i, l in enumerate(body) if "Loop Header: Depth=1" in l), None)
    last = max((i for i, l in enumerate(body) if "in Loop: Header=" in l or "Parent Loop" in l or "Inner Loop Header" in l), default=len(body) - 1)
    def count(lines):
        c = dict(valu=0, salu=0, lds=0, vmem=0, smem=0, branch=0, wait=0, other=0)
        for l in lines:
            m = re.match(r"\s+([a-z_0-9]+)", l)
            if not m: continue
            op = m.group(1)
            if op.startswith(("v_",)): c["valu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): c["vmem"] += 1
            elif op.startswith("s_load") or op.startswith("s_buffer"): c["smem"] += 1
            elif op.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
            elif op.startswith(("s_waitcnt", "s_nop")): c["wait"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            else: c["other"] += 1
        return c
    print(text[start][:80])
    print("  whole kernel:", count(body))
    if hdr is not None:
        print("  main loop   :", count(body[hdr:last + 1]))
