#!/usr/bin/env python3
"""VGPR liveness of one kernel from its device assembly (tools/isa.py output): which registers are live THROUGH a given basic
block without being touched in it -- what the machine scheduler sees as pressure it cannot move.
  python tools/isa_liveness.py /tmp/lr_f64_p8.s _ZN2lr13k_chain_mixedILi13E [.LBB40_40]
Without a block: every block with its instruction count, registers touched, live-through count."""
import re, sys

path, kern = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else None
s = open(path).read()
i = s.index("\n" + kern)
j = s.index("s_endpgm", i)
lines = s[i:j].split("\n")[1:]

blocks, order = {}, []
cur = "entry"
blocks[cur] = []
order.append(cur)
for l in lines:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        cur = m.group(1)
        blocks[cur] = []
        order.append(cur)
        continue
    t = l.split(";")[0].strip()
    if not t or t.startswith("."):
        continue
    blocks[cur].append(t)


def regs(tok):
    out = []
    for m in re.finditer(r"(?<![\w.])v\[(\d+):(\d+)\]|(?<![\w.\[:])v(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


NO_DEF = ("v_cmp", "v_cmpx", "ds_write", "global_store", "buffer_store", "scratch_store", "v_readlane", "v_readfirstlane", "s_", "v_accvgpr_write",
          "global_atomic", "ds_add", "flat_store")
RMW = ("v_writelane", "v_fmac", "v_pk_fmac", "v_swap", "v_permlane", "v_mac")


def def_use(ins):
    op = ins.split()[0]
    ops = ins[len(op):].split(",")
    ops = [o.strip() for o in ops]
    if op.startswith(NO_DEF):
        return [], [r for o in ops for r in regs(o)]
    d = regs(ops[0]) if ops else []
    u = [r for o in ops[1:] for r in regs(o)]
    partial = "bank_mask:0x" in ins and "bank_mask:0xf" not in ins
    if op.startswith(RMW) or partial or "_dpp" in op and "bound_ctrl" not in ins:
        u += d
    if op.startswith(("v_swap", "v_permlane")):
        d += regs(ops[1])
    return d, u


succ = {}
for k, b in enumerate(order):
    ins = blocks[b]
    nxt = order[k + 1] if k + 1 < len(order) else None
    ss = []
    fall = True
    for t in ins:
        m = re.match(r"s_c?branch\S*\s+(\.LBB\S+)", t)
        if m:
            ss.append(m.group(1))
            if t.startswith("s_branch"):
                fall = False
    if fall and nxt:
        ss.append(nxt)
    succ[b] = ss

gen, kill, touched = {}, {}, {}
for b in order:
    g, k, t = set(), set(), set()
    for ins in blocks[b]:
        d, u = def_use(ins)
        for r in u:
            if r not in k:
                g.add(r)
        k.update(d)
        t.update(d)
        t.update(u)
    gen[b], kill[b], touched[b] = g, k, t
live_in = {b: set() for b in order}
live_out = {b: set() for b in order}
changed = True
while changed:
    changed = False
    for b in reversed(order):
        lo = set()
        for x in succ[b]:
            lo |= live_in.get(x, set())
        li = gen[b] | (lo - kill[b])
        if lo != live_out[b] or li != live_in[b]:
            live_out[b], live_in[b] = lo, li
            changed = True
for b in order:
    if want and b != want:
        continue
    through = (live_in[b] & live_out[b]) - touched[b]
    print(f"{b:14s} {len(blocks[b]):5d} instr  touched {len(touched[b]):3d}  live-in {len(live_in[b]):3d}  live-out {len(live_out[b]):3d}  live-through-untouched {len(through):3d}")
    if want:
        print("untouched live-through:", sorted(through))
