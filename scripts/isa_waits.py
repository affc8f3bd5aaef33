"""List the s_waitcnt vmcnt / barriers / LDS-DMA / MFMA groups of one kernel in a gfx950 assembly dump.
   python scripts/isa_waits.py file.s <substring of the mangled kernel name>"""
import re
import sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ": ;" in l]
for st in start:
    end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    print(lines[st], end - st, "lines")
    nm = 0
    for i in range(st, end):
        l = lines[i].strip()
        if "v_mfma" in l:
            nm += 1
            continue
        tag = None
        if l.startswith("s_waitcnt") and ("vmcnt" in l):
            tag = l.split(";")[0]
        elif l.startswith("s_barrier"):
            tag = "BARRIER"
        elif "buffer_load" in l and " lds" in l:
            tag = "DMA"
        elif l.startswith("s_cbranch") or (l.endswith(":") and l.startswith(".LBB")):
            tag = l
        elif l.startswith("global_store") or l.startswith("global_load") or l.startswith("buffer_store") or l.startswith("scratch_"):
            tag = l.split()[0]
        if tag:
            if nm:
                print("      [%d mfma]" % nm)
                nm = 0
            print("   %5d %s" % (i - st, tag))
