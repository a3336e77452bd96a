#!/usr/bin/env python
"""Audit of hand-counted vmcnt waits in hipcc output (guide section 5.7, item 1).

Walks a kernel's instructions in program order and keeps the FIFO of outstanding vector-memory operations
(gfx9 model: loads and stores retire in order on one counter; `s_waitcnt vmcnt(N)` retires all but the N
youngest).  Flags any instruction that READS or WRITES a VGPR that is the destination of a still-outstanding
load -- e.g. a compiler `v_mov` of an inline-asm load destination before its counted wait, or a wait whose
count is one too high.

Control flow: a kernel starts with an empty FIFO; the walk carries a SET of FIFOs, one per distinct history.
A forward branch hands a copy of the current FIFOs to its target (an unconditional one ends the fall-through),
so the two sides of an if / else are judged separately and meet again behind it.  A backward branch re-runs
its loop body: `trips - 1` more times when the loop is a counted one whose trip count the code states
(`s_mov_b32 sN, imm` ... `s_add_i32 sN, sN, -1` / `s_cmp_lg_u32 sN, 0` / `s_cbranch_scc1`, the form hipcc
gives constant-trip loops), once more otherwise (the steady state of a ring loop repeats after one iteration).

Packed instructions: `v_pk_*` with op_sel / op_sel_hi reads only the halves of its 64-bit sources that the
selects name (`op_sel_hi:[1,0,1]` on `v[2:3]` reads v2 only).

usage: check_asm_waits.py file.s [kernel-substring]      (exit status 1 if anything was flagged)
"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def _sel(l, name, n, default):
    m = re.search(name + r":\[([01,]+)\]", l)
    if not m:
        return [default] * n
    v = [int(x) for x in m.group(1).split(",")]
    return (v + [default] * n)[:n]


def touched_regs(op, args, l):
    """VGPRs an ALU / LDS instruction reads or writes."""
    if op.startswith("v_pk_") and ("op_sel" in l):
        parts = [a.strip() for a in re.sub(r"\s+(op_sel|op_sel_hi|neg_lo|neg_hi|clamp)\b.*$", "", args).split(",")]
        out = regs(parts[0])
        srcs = parts[1:]
        lo, hi = _sel(l, "op_sel", len(srcs), 0), _sel(l, "op_sel_hi", len(srcs), 1)
        for i, s in enumerate(srcs):
            m = re.match(r"v\[(\d+):(\d+)\]$", s)
            if m and int(m.group(2)) == int(m.group(1)) + 1:
                base = int(m.group(1))
                out.update({base + lo[i], base + hi[i]})
            else:
                out.update(regs(s))
        return out
    return regs(args)


class Checker(object):
    """Forward walk over one kernel with a SET of vmcnt FIFOs (one per distinct control-flow history)."""
    MAX_STATES = 12

    def __init__(self, name, ins, labels, verbose=True):
        self.name, self.ins, self.labels, self.verbose = name, ins, labels, verbose
        self.bad, self.seen = 0, set()

    def flag(self, ln, what, l):
        key = (ln, what)
        if key in self.seen:
            return
        self.seen.add(key)
        self.bad += 1
        if self.verbose:
            print("%s:%d %s : %s" % (self.name, ln, what, l))

    @staticmethod
    def pending(fifo):
        out = set()
        for d in fifo:
            if d:
                out |= d
        return out

    def step(self, fifo, ln, l):
        """fifo: tuple of frozensets (load destinations; empty = a store or an LDS-DMA) -> new fifo"""
        op = l.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                if len(fifo) > n:
                    fifo = fifo[len(fifo) - n:] if n else ()
            return fifo
        args = l[len(op):]
        parts = [a.strip() for a in args.split(",")]
        if op.startswith(("buffer_load", "global_load", "flat_load", "scratch_load")):
            to_lds = re.search(r"\blds\b", l) is not None
            read = regs(",".join(parts if to_lds else parts[1:]))
            hit = read & self.pending(fifo)
            if hit:
                self.flag(ln, "reads pending %s" % sorted(hit), l)
            fifo = fifo + (frozenset() if to_lds else frozenset(regs(parts[0])),)
        elif op.startswith(("buffer_store", "global_store", "flat_store", "scratch_store", "buffer_atomic", "global_atomic")):
            hit = regs(args) & self.pending(fifo)
            if hit:
                self.flag(ln, "store touches pending %s" % sorted(hit), l)
            fifo = fifo + (frozenset(),)
        elif op.startswith(("s_", ";")):
            return fifo
        else:
            hit = touched_regs(op, args, l) & self.pending(fifo)
            if hit:
                self.flag(ln, "touches pending %s" % sorted(hit), l)
        return fifo[-64:]                                      # vmcnt holds 6 bits: older entries can never matter

    def trips(self, start, end):
        """Trip count of the loop [start, end] (indices into ins; end = the backward branch) if the code states it."""
        for j in range(end - 1, max(start, end - 8), -1):
            m = re.match(r"s_(?:add|sub)_i32\s+(s\d+),\s*\1,\s*(-?1)\b", self.ins[j][1])
            if m:
                cnt = m.group(1)
                for k in range(start - 1, max(-1, start - 40), -1):
                    mi = re.match(r"s_mov(?:k_i32|_b32)\s+%s,\s*(0x[0-9a-f]+|\d+)\s*$" % cnt, self.ins[k][1])
                    if mi:
                        v = int(mi.group(1), 0)
                        return v if 0 < v <= 4096 else None
                    if re.match(r"\S+\s+%s\b" % cnt, self.ins[k][1]):
                        return None
                return None
        return None

    def walk(self, lo, hi, entry, depth=0, own_back=-1):
        """Instructions lo .. hi-1 entered at lo with the states `entry`; -> states that fall out at hi.
        Forward branches hand their states to the target; a backward branch re-runs its loop body."""
        inject, cur = {}, set(entry)
        out_of_range = set()
        for i in range(lo, hi):
            cur |= inject.pop(i, set())
            if not cur:
                continue
            ln, l = self.ins[i]
            cur = {self.step(f, ln, l) for f in cur}
            if len(cur) > self.MAX_STATES:
                cur = set(sorted(cur, key=len, reverse=True)[:self.MAX_STATES])
            op = l.split()[0]
            if op == "s_endpgm":
                cur = set()
                continue
            m = re.match(r"s_c?branch\S*\s+(\S+)", l)
            if not m or m.group(1) not in self.labels:
                continue
            tgt = self.labels[m.group(1)]
            if tgt > i:                                          # forward
                (inject.setdefault(tgt, set()) if tgt < hi else out_of_range).update(cur)
                if op == "s_branch":
                    cur = set()
            elif i == own_back:                                  # the back edge of the body being re-walked
                pass
            elif tgt >= lo and depth < 4:                        # loop [tgt, i]
                t = self.trips(tgt, i)
                again = set(cur)
                for _ in range(min((t - 1) if t else 1, 63)):
                    again = self.walk(tgt, i + 1, again, depth + 1, own_back=i)
                    if not again:
                        break
                # a loop with a stated trip count leaves only after its last trip; any other may leave after each
                cur = set() if op == "s_branch" else (again if t else (cur | again))
        for k, v in inject.items():                              # targets past hi
            out_of_range |= v
        return cur | out_of_range

    def run(self):
        self.walk(0, len(self.ins), {()})


def check_text(text, want="", verbose=True):
    """-> (kernels checked, violations)"""
    total = nk = 0
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@", text, re.M):
        name = m.group(1)
        if want not in name:
            continue
        end = text.find(".end_amdhsa_kernel", m.end())
        if end < 0:
            end = text.find("s_endpgm", m.end())
        ins, labels = [], {}
        for i, raw in enumerate(text[m.end():end].split("\n")):
            l = raw.split(";")[0].strip()
            if not l or l.startswith("."):
                if l.startswith(".LBB") and l.endswith(":"):
                    labels[l[:-1]] = len(ins)
                continue
            if l.endswith(":"):
                labels[l[:-1]] = len(ins)
                continue
            ins.append((i, l))
        c = Checker(name[:70], ins, labels, verbose)
        c.run()
        total += c.bad
        nk += 1
    return nk, total


def main():
    nk, total = check_text(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "")
    print("kernels:", nk, "violations:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
