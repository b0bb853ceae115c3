#!/usr/bin/env python3
"""Symbolic execution of one straight-line region of gfx950 device assembly: every 32-bit register (v, a, s) holds an expression
tree over the values that were live at the region's start.  Written to locate round 4's wrong-result kernel:
the butterfly all-reduce of a float64 vector is 32 identical dataflow trees, one per coordinate -- a coordinate whose tree differs
from the others' (an operand taken from a register that was overwritten, halves of two different values glued together, a spill
slot read back after its reuse) is the miscompiled one.

    python tools/isa_symexec.py <file.s> <first line> <last line> [--sums]

--sums: print, for every v_add_f64 whose operands are the two results of a v_permlane32_swap pair (the last level of group_sum),
the set of region inputs the sum depends on and a structural hash; the odd one out is flagged."""
import hashlib
import re
import sys


class Sym:
    __slots__ = ("op", "args", "_h", "_leaves")

    def __init__(self, op, args=()):
        self.op, self.args, self._h, self._leaves = op, tuple(args), None, None

    def h(self):
        if self._h is None:
            m = hashlib.sha1(self.op.encode())
            for a in self.args:
                m.update(a.h() if isinstance(a, Sym) else str(a).encode())
            self._h = m.digest()
        return self._h

    def leaves(self):
        if self._leaves is None:
            if not self.args and self.op.startswith("in:"):
                self._leaves = frozenset([self.op])
            else:
                s = frozenset()
                for a in self.args:
                    if isinstance(a, Sym):
                        s |= a.leaves()
                self._leaves = s
        return self._leaves

    def shape(self):
        """structural hash with the leaves' NAMES erased (so that coordinates can be compared with each other)"""
        if not self.args and self.op.startswith("in:"):
            return b"L"
        m = hashlib.sha1(self.op.encode())
        for a in self.args:
            m.update(a.shape() if isinstance(a, Sym) else str(a).encode())
        return m.digest()

    def show(self, depth=3):
        if not self.args:
            return self.op
        if depth == 0:
            return self.op + "(..)"
        return self.op + "(" + ", ".join(a.show(depth - 1) if isinstance(a, Sym) else str(a) for a in self.args) + ")"


class Machine:
    def __init__(self):
        self.r = {}
        self.mem = {}  # scratch memory by byte offset (spill slots): 32-bit symbols
        self.log = []  # (line number, text, dest names, value)

    def get(self, name):
        if name not in self.r:
            self.r[name] = Sym("in:" + name)
        return self.r[name]

    def set(self, name, v):
        self.r[name] = v


def expand(tok):
    """'v[10:11]' -> ['v10', 'v11'];  'v5' -> ['v5'];  anything else -> None"""
    tok = tok.strip()
    m = re.fullmatch(r"([vas])\[(\d+):(\d+)\]", tok)
    if m:
        return [m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if re.fullmatch(r"[vas]\d+", tok):
        return [tok]
    return None


def src_val(mc, tok, width):
    """value of a source operand: a list of `width` 32-bit symbols"""
    tok = tok.strip()
    neg = tok.startswith("-")
    if neg:
        tok = tok[1:]
    ab = tok.startswith("|") and tok.endswith("|")
    if ab:
        tok = tok[1:-1]
    regs = expand(tok)
    if regs is None:
        vals = [Sym("const:" + tok)] * width
    else:
        vals = [mc.get(r) for r in regs]
        if len(vals) < width:
            vals = vals + [Sym("const:pad")] * (width - len(vals))
    if ab:
        vals = [Sym("abs", [v]) for v in vals]
    if neg:
        vals = [Sym("neg", [v]) for v in vals]
    return vals


def wide(vals):
    """a 64-bit operand from its two halves: the halves of ONE 64-bit result collapse back to it"""
    lo, hi = vals[0], vals[1]
    if lo.op == "lo" and hi.op == "hi" and lo.args[0] is hi.args[0]:
        return lo.args[0]
    return Sym("pair", [lo, hi])


def run(lines, first, mc=None, lane_agnostic_only=False):
    """lane_agnostic_only: the path of a lane whose EXEC bit is off through this region -- only the scalar unit and
    v_readlane / v_writelane (which ignore EXEC) act"""
    mc = mc or Machine()
    for off, raw in enumerate(lines):
        ln = first + off
        t = raw.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op, _, rest = t.partition(" ")
        if lane_agnostic_only and not (op.startswith("s_") or op in ("v_readlane_b32", "v_writelane_b32")):
            continue
        toks = [x.strip() for x in re.split(r",(?![^\[]*\])", rest)] if rest else []
        if op in ("s_nop", "s_waitcnt", "s_barrier", "s_cbranch_execz", "s_cbranch_execnz", "s_cbranch_vccnz", "s_cbranch_vccz", "s_branch", "s_endpgm"):
            continue
        if op.startswith("v_permlane") and "swap" in op:
            a, b = toks[0], toks[1]
            va, vb = mc.get(a), mc.get(b)
            w = "32" if "permlane32" in op else "16"
            na, nb = Sym("swapA" + w, [va, vb]), Sym("swapB" + w, [va, vb])
            mc.set(a, na)
            mc.set(b, nb)
            mc.log.append((ln, t, [a, b], [na, nb]))
            continue
        if op == "v_mov_b32_dpp":
            ctrl = " ".join(toks[1].split()[1:]) if " " in toks[1] else ""
            s = toks[1].split()[0]
            v = Sym("dpp[" + ctrl + "]", [mc.get(s)])
            mc.set(toks[0], v)
            mc.log.append((ln, t, [toks[0]], [v]))
            continue
        if op in ("v_mov_b32_e32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "s_mov_b32"):
            v = src_val(mc, toks[1], 1)[0]
            mc.set(toks[0], v)
            mc.log.append((ln, t, [toks[0]], [v]))
            continue
        if op in ("v_mov_b64_e32", "s_mov_b64"):
            d = expand(toks[0])
            if d is None:  # exec, vcc
                continue
            vals = src_val(mc, toks[1], 2)
            for r, v in zip(d, vals):
                mc.set(r, v)
            mc.log.append((ln, t, d, vals))
            continue
        if op.startswith("scratch_load") and toks[1] == "off" and toks[2].startswith("off"):
            d = expand(toks[0])
            m = re.search(r"offset:(\d+)", toks[2])
            base = int(m.group(1)) if m else 0
            vals = [mc.mem.get(base + 4 * i) or Sym("in:scratch%d" % (base + 4 * i)) for i in range(len(d))]
            for r, v in zip(d, vals):
                mc.set(r, v)
            mc.log.append((ln, t, d, vals))
            continue
        if op.startswith("scratch_store") and toks[0] == "off" and toks[2].startswith("off"):
            m = re.search(r"offset:(\d+)", toks[2])
            base = int(m.group(1)) if m else 0
            for i, r in enumerate(expand(toks[1])):
                mc.mem[base + 4 * i] = mc.get(r)
            mc.log.append((ln, t, [], []))
            continue
        if op.startswith("ds_read"):
            d = expand(toks[0])
            addr = toks[1].split()
            base = Sym("lds:" + op + " " + " ".join(addr[1:]), [mc.get(addr[0])])
            vals = [Sym("w%d" % i, [base]) for i in range(len(d))]
            for r, v in zip(d, vals):
                mc.set(r, v)
            mc.log.append((ln, t, d, vals))
            continue
        if op.startswith("scratch_load") or op.startswith("ds_read") or op.startswith("global_load") or op.startswith("s_load"):
            d = expand(toks[0])
            base = Sym("mem:" + op + " " + ", ".join(toks[1:]) + "@" + str(ln))
            vals = [Sym("w%d" % i, [base]) for i in range(len(d))]
            for r, v in zip(d, vals):
                mc.set(r, v)
            mc.log.append((ln, t, d, vals))
            continue
        if op.startswith("scratch_store") or op.startswith("global_store") or op.startswith("ds_write") or op.startswith("flat_store"):
            mc.log.append((ln, t, [], []))
            continue
        # generic: first operand is the destination
        d = expand(toks[0]) if toks else None
        if d is None:
            mc.log.append((ln, t, [], []))
            continue
        is64 = "_f64" in op or op.endswith("_b64") or "_u64" in op or "_i64" in op
        srcs = []
        for s in toks[1:]:
            if re.match(r"(row_|quad_|bank_|bound_|op_sel|neg_|clamp|mul:|div:|offset|off\b)", s):
                continue
            srcs.append(wide(src_val(mc, s, 2)) if is64 and (expand(s.lstrip("-|").rstrip("|")) or [0, 0]).__len__() == 2 else src_val(mc, s, 1)[0])
        if op.startswith("v_fmac") or op.startswith("v_mac"):
            srcs.append(wide([mc.get(r) for r in d]) if len(d) == 2 else mc.get(d[0]))
        res = Sym(op, srcs)
        if len(d) == 2:
            vals = [Sym("lo", [res]), Sym("hi", [res])]
        else:
            vals = [res] + [Sym("part%d" % i, [res]) for i in range(1, len(d))]
        for r, v in zip(d, vals):
            mc.set(r, v)
        mc.log.append((ln, t, d, vals))
    return mc


def main():
    path, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lines = open(path).read().split("\n")[first - 1:last]
    mc = run(lines, first)
    if "--sums" in sys.argv:
        rows = []
        for ln, t, d, vals in mc.log:
            if not t.startswith("v_add_f64") or not vals:
                continue
            res = vals[0].args[0]
            txt = res.show(4)
            if "swapA32" in txt and "swapB32" in txt and res.args[0].op == "pair" and res.args[1].op == "pair":
                rows.append((ln, t, res))
        from collections import Counter
        shapes = Counter(r[2].shape() for r in rows)
        common = shapes.most_common(1)[0][0] if shapes else None
        for ln, t, res in rows:
            lv = sorted(res.leaves())
            flag = "" if res.shape() == common else "   <-- ODD SHAPE"
            print("%6d  %-44s leaves(%d): %s%s" % (ln, t, len(lv), " ".join(x[3:] for x in lv), flag))
        print("%d final sums, %d distinct shapes" % (len(rows), len(shapes)))
    return mc


if __name__ == "__main__":
    main()
