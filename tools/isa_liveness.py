"""Approximate VGPR liveness of a gfx950 kernel from hipcc's assembly: where is the register pressure?

    python tools/isa_liveness.py file.s kernel_substring [top_n]

Builds the control-flow graph from labels and branches, runs a backward dataflow over v0..v511 and prints, per basic
block, the maximum number of live vector registers and the instruction at which it occurs.  Approximate: every first
vector operand of a v_* / *_load* instruction is taken as a full definition (partial writes and sub-dword destinations
are treated as full ones), everything else as uses."""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
NO_DEF = ("v_cmp", "v_cmpx", "ds_write", "buffer_store", "global_store", "scratch_store", "flat_store", "v_nop", "ds_bpermute_dummy",
          "buffer_wbl2", "buffer_inv", "global_atomic", "buffer_atomic")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def main():
    path, key = sys.argv[1], sys.argv[2]
    topn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or
                 (l.startswith("_Z") and key in l and ":" in l))
    blocks, order, cur = {}, [], "entry"
    blocks[cur] = []
    order.append(cur)
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        s = s.split(";")[0].strip()
        if s:
            blocks[cur].append(s)
    # successors
    succ = {}
    for i, b in enumerate(order):
        ss, fall = [], True
        for ins in blocks[b]:
            op = ins.split()[0]
            if op == "s_branch":
                ss.append(ins.split()[1])
                fall = False
            elif op.startswith("s_cbranch"):
                ss.append(ins.split()[1])
            elif op in ("s_endpgm", "s_setpc_b64"):
                fall = False
        if fall and i + 1 < len(order):
            ss.append(order[i + 1])
        succ[b] = [x for x in ss if x in blocks]

    def defs_uses(ins):
        parts = ins.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if not ops:
            return [], []
        writes = (op.startswith("v_") or "_load" in op or op.startswith("ds_read") or op.startswith("ds_bpermute") or
                  op.startswith("ds_swizzle")) and not op.startswith(NO_DEF)
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            return [], regs(" ".join(ops[1:]))
        if writes:
            d = regs(ops[0])
            u = regs(" ".join(ops[1:]))
            if op.startswith("v_writelane") or "_dpp" in op or "sdwa" in op or op.startswith("v_fmac") or op.startswith("v_mac") or \
                    op.startswith("v_pk_fmac"):
                u = u + d  # destination is also read
            return d, u
        return [], regs(" ".join(ops))

    info = {b: [defs_uses(i) for i in blocks[b]] for b in order}
    live_in = {b: set() for b in order}
    changed = True
    while changed:
        changed = False
        for b in reversed(order):
            live = set()
            for s_ in succ[b]:
                live |= live_in[s_]
            for d, u in reversed(info[b]):
                live -= set(d)
                live |= set(u)
            if live != live_in[b]:
                live_in[b] = live
                changed = True
    rows = []
    for b in order:
        live = set()
        for s_ in succ[b]:
            live |= live_in[s_]
        best, at = len(live), len(blocks[b])
        for k in range(len(blocks[b]) - 1, -1, -1):
            d, u = info[b][k]
            live -= set(d)
            live |= set(u)
            if len(live) > best:
                best, at = len(live), k
        rows.append((best, b, at, len(blocks[b])))
    rows.sort(reverse=True)
    for best, b, at, n in rows[:topn]:
        ins = blocks[b][at] if at < n else "(block end)"
        print(f"{b:12s} max live {best:4d} at {at:4d}/{n:4d}  {ins[:90]}")


if __name__ == "__main__":
    main()
