"""Instruction-class picture of a kernel's MFMA-holding basic blocks from a hipcc -S / --save-temps device listing:
M = MFMA, v = other VALU, r / w = LDS read / write, G = buffer or global access, | = barrier, _ = s_waitcnt, B = branch, s = scalar.
Usage: python tools/asm_loop_shape.py listing.s 'substring of the demangled kernel name' [max_chars]"""
import re, subprocess, sys
lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 900
starts = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
names = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True).stdout.split("\n")
for (i, _), d in zip(starts, names):
    if want not in d:
        continue
    j = i
    out, k = "", j
    while not lines[k].startswith(".Lfunc_end"):
        k += 1
        t = lines[k].split(";")[0].strip()
        if not t:
            continue
        op = t.split()[0]
        if op.startswith(".LBB"):
            out += "\n" + op + " "
        elif "mfma" in op: out += "M"
        elif op.startswith("v_"): out += "v"
        elif op.startswith("ds_read"): out += "r"
        elif op.startswith("ds_write"): out += "w"
        elif op.startswith(("buffer", "global", "flat")): out += "G"
        elif "barrier" in op: out += "|"
        elif "waitcnt" in op: out += "_"
        elif "branch" in op: out += "B"
        elif op.startswith("s_"): out += "s"
    print(d[:160])
    shown = 0
    for blk in out.split("\n"):               # only the blocks that hold MFMAs (the main loops), up to `limit` characters
        if "M" in blk and shown < limit:
            print(blk); shown += len(blk)
    break
