"""Static instruction statistics of a gfx950 kernel from hipcc's assembly (-save-temps / -S):

    python tools/isa_stats.py file.s kernel_substring [min_block_size]

Per basic block: vector-ALU, scalar-ALU, scalar-memory, vector-memory, LDS, s_waitcnt, s_nop, scratch (spill) counts and
the DPP-carrying adds; then totals for the kernel.  Used to compare builds of the fused chain kernels without a GPU
(registers, spills, instructions per row step)."""
import re
import sys
from collections import OrderedDict


def classify(op):
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_"):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].startswith("_Z") and ":" in l:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = {"hdr": ""}
    tot = {}
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", s)
        if m:
            cur = m.group(1)
            blocks[cur] = {"hdr": (m.group(2) or "")}
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        c = classify(op)
        b = blocks[cur]
        b[c] = b.get(c, 0) + 1
        tot[c] = tot.get(c, 0) + 1
        if "_dpp" in op:
            b["dpp"] = b.get("dpp", 0) + 1
            tot["dpp"] = tot.get("dpp", 0) + 1
        if op.startswith("v_mov_b32") or op.startswith("v_accvgpr"):
            b["mov"] = b.get("mov", 0) + 1
            tot["mov"] = tot.get("mov", 0) + 1
        if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
            b["lane"] = b.get("lane", 0) + 1
            tot["lane"] = tot.get("lane", 0) + 1
    keys = ["valu", "salu", "smem", "vmem", "lds", "wait", "nop", "scratch", "dpp", "mov", "lane"]
    print("%-12s " % "block" + " ".join("%7s" % k for k in keys))
    for name, b in blocks.items():
        n = sum(v for k, v in b.items() if k not in ("hdr", "dpp", "mov", "lane"))
        if n >= minsz:
            print("%-12s " % name + " ".join("%7d" % b.get(k, 0) for k in keys) + "  " + b["hdr"][:60])
    print("%-12s " % "TOTAL" + " ".join("%7d" % tot.get(k, 0) for k in keys))


if __name__ == "__main__":
    main()
