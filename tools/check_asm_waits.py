#!/usr/bin/env python
"""Audit of hand-counted vmcnt waits in hipcc output (guide section 5.7, item 1).

Walks a kernel's instructions in program order and keeps the FIFO of outstanding vector-memory operations
(gfx9 model: loads and stores retire in order on one counter; `s_waitcnt vmcnt(N)` retires all but the N
youngest).  Flags any instruction that READS or WRITES a VGPR that is the destination of a still-outstanding
load -- e.g. a compiler `v_mov` of an inline-asm load destination before its counted wait, or a wait whose
count is one too high.

Control flow: the kernel is cut into basic blocks; dominators give its natural loops (a backward jump whose target
does not dominate it -- hipcc parks the rare arm of an if / else behind the main code and jumps back -- is an ordinary
edge).  Blocks are visited in reverse post-order with a SET of FIFOs per block, one per distinct history (a kernel
starts with an empty FIFO), so the two sides of an if / else are judged separately and meet again behind it.  A loop is
walked as a unit when its header is reached: `trips` passes when it is a counted one whose trip count the code states
(`s_mov_b32 sN, imm` ... `s_add_i32 sN, sN, -1` / `s_cmp_lg_u32 sN, 0` / `s_cbranch_scc1`, the form hipcc gives
constant-trip loops; it is left through its latch after the last pass only), two passes otherwise (the steady state of
a ring loop repeats after one iteration).

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

    def __init__(self, name, ins, labels, verbose=True, min_store_run=0):
        self.name, self.ins, self.labels, self.verbose = name, ins, labels, verbose
        self.bad, self.seen = 0, set()
        self.min_store_run = min_store_run

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
        """Trip count of the loop [start, end] (indices into ins; end = the backward branch) if the code states it:
        a scalar counter with ONE writer in the loop (`s_add_i32 sN, sN, STEP`), a constant start (`s_mov_b32 sN, INIT`
        in front of the loop) and a compare against a constant at the latch (`s_cmp_lg_u32 sN, LIMIT` + scc1, or
        `s_cmp_eq_u32 sN, LIMIT` + scc0) -- counting down to 0 or up to a limit, as hipcc emits constant-trip loops."""
        br = self.ins[end][1]
        cmp_ = None
        for j in range(end - 1, max(start - 1, end - 16), -1):
            m = re.match(r"s_cmp_(eq|lg)_u32\s+(s\d+),\s*(0x[0-9a-f]+|\d+)\s*$", self.ins[j][1])
            if m:
                cmp_ = m
                break
            if re.match(r"s_cmp|v_cmp|s_and_b64\s+vcc|s_andn2_b64\s+vcc", self.ins[j][1]):
                return None                                      # the branch tests something else
        if cmp_ is None:
            return None
        if not ((cmp_.group(1) == "lg" and br.startswith("s_cbranch_scc1")) or
                (cmp_.group(1) == "eq" and br.startswith("s_cbranch_scc0"))):
            return None
        cnt, limit = cmp_.group(2), int(cmp_.group(3), 0)
        step = None
        for j in range(start, end):
            l = self.ins[j][1]
            m = re.match(r"s_(add|sub)_i32\s+%s,\s*%s,\s*(-?\d+)\s*$" % (cnt, cnt), l)
            if m:
                if step is not None:
                    return None
                step = int(m.group(2)) * (1 if m.group(1) == "add" else -1)
            elif re.match(r"s_\S+\s+%s\b" % cnt, l) and not l.startswith("s_cmp"):
                return None                                      # another writer of the counter
        if not step:
            return None
        for k in range(start - 1, max(-1, start - 60), -1):
            mi = re.match(r"s_mov(?:k_i32|_b32)\s+%s,\s*(0x[0-9a-f]+|\d+)\s*$" % cnt, self.ins[k][1])
            if mi:
                span = limit - int(mi.group(1), 0)
                if span % step == 0 and 0 < span // step <= 4096:
                    return span // step
                return None
            if re.match(r"s_\S+\s+%s\b" % cnt, self.ins[k][1]) and not self.ins[k][1].startswith("s_cmp"):
                return None
        return None

    # ---- control flow: basic blocks, dominators, natural loops; blocks are visited in reverse post-order -------
    def build_cfg(self):
        n = len(self.ins)
        tgt_of = {}
        leaders = {0}
        for i, (_, l) in enumerate(self.ins):
            op = l.split()[0]
            m = re.match(r"s_c?branch\S*\s+(\S+)", l)
            if m and m.group(1) in self.labels:
                tgt_of[i] = self.labels[m.group(1)]
                leaders.add(tgt_of[i])
                leaders.add(i + 1)
            elif op in ("s_endpgm", "s_setpc_b64"):
                leaders.add(i + 1)
        for v in self.labels.values():
            leaders.add(v)
        starts = sorted(x for x in leaders if x < n)
        self.blk_of = {}
        self.blocks = []
        for k, st in enumerate(starts):
            en = starts[k + 1] if k + 1 < len(starts) else n
            self.blocks.append((st, en))
            self.blk_of[st] = k
        self.succ = []
        for k, (st, en) in enumerate(self.blocks):
            last = self.ins[en - 1][1]
            op = last.split()[0]
            out = []
            if en - 1 in tgt_of:
                out.append(self.blk_of[tgt_of[en - 1]])
                if op != "s_branch" and en < n:
                    out.append(self.blk_of[en])
            elif op not in ("s_endpgm", "s_setpc_b64") and en < n:
                out.append(self.blk_of[en])
            self.succ.append(out)
        nb = len(self.blocks)
        # reverse post-order (iterative DFS) and dominators (Cooper / Harvey / Kennedy)
        order, seen, stack = [], {0}, [(0, iter(self.succ[0]))]
        while stack:
            v, it = stack[-1]
            nxt = next(it, None)
            if nxt is None:
                order.append(v)
                stack.pop()
            elif nxt not in seen:
                seen.add(nxt)
                stack.append((nxt, iter(self.succ[nxt])))
        self.rpo = order[::-1]
        num = {v: i for i, v in enumerate(self.rpo)}
        preds = [[] for _ in range(nb)]
        for u in self.rpo:
            for v in self.succ[u]:
                preds[v].append(u)
        idom = {0: 0}
        changed = True
        while changed:
            changed = False
            for v in self.rpo[1:]:
                cand = [p for p in preds[v] if p in idom]
                if not cand:
                    continue
                new = cand[0]
                for p in cand[1:]:
                    a, b2 = p, new
                    while a != b2:
                        while num[a] > num[b2]:
                            a = idom[a]
                        while num[b2] > num[a]:
                            b2 = idom[b2]
                    new = a
                if idom.get(v) != new:
                    idom[v] = new
                    changed = True

        def dominates(a, v):
            while True:
                if v == a:
                    return True
                if v == 0 or v not in idom:
                    return False
                v = idom[v]
        # natural loops: header -> (body, latches)
        self.loops = {}
        for u in self.rpo:
            for v in self.succ[u]:
                if dominates(v, u):
                    body, latches = self.loops.setdefault(v, ({v}, set()))
                    latches.add(u)
                    work = [u]
                    while work:
                        x = work.pop()
                        if x not in body:
                            body.add(x)
                            work.extend(preds[x])

    def cap(self, states):
        if len(states) > self.MAX_STATES:
            return set(sorted(states, key=len, reverse=True)[:self.MAX_STATES])
        return states

    def run_region(self, region, head, entry, depth):
        """One pass over the blocks of `region` (reverse post-order) entered at `head`.
        -> (exits: {(src block, dst block): states}, latch: states that come back to `head`)"""
        inbox = {head: set(entry)}
        exits, latch, skip = {}, set(), set()

        def route(src, dst, st):
            if not st:
                return
            if dst == head and src != -1 and dst in self.loops and src in self.loops[dst][1]:
                latch.update(st)
            elif dst in region and dst not in skip:
                inbox[dst] = self.cap(inbox.get(dst, set()) | st)
            else:
                exits.setdefault((src, dst), set()).update(st)

        for b in self.rpo:
            if b not in region or b in skip:
                continue
            cur = inbox.pop(b, None)
            if not cur:
                continue
            if b != head and b in self.loops and depth < 6:
                body = self.loops[b][0]
                if body <= region:
                    for (src, dst), st in self.run_loop(b, cur, depth + 1).items():
                        route(src, dst, st)
                    skip |= body
                    continue
            st_, en = self.blocks[b]
            for i in range(st_, en):
                ln, l = self.ins[i]
                cur = {self.step(f, ln, l) for f in cur}
            cur = self.cap(cur)
            for dst in self.succ[b]:
                route(b, dst, cur)
        return exits, latch

    def run_loop(self, h, entry, depth):
        body, latches = self.loops[h]
        t = None
        for u in latches:                                        # a counted loop states its trip count at its latch
            t = t or self.trips(self.blocks[h][0], self.blocks[u][1] - 1)
        if not t and self.min_store_run:
            # annotated kernels: a store-only loop whose trip count is a run-time value that the HOST guarantees
            # to issue at least `min_store_run` stores (mix_sru_fwd_kernel<.., WB>: the word loop runs L >= WB steps)
            ops = [self.ins[i][1].split()[0] for b in body for i in range(*self.blocks[b])]
            nst = sum(o.startswith(("global_store", "buffer_store", "flat_store")) for o in ops)
            if nst and not any(o.startswith(("global_load", "buffer_load", "flat_load")) for o in ops):
                t = -(-self.min_store_run // nst)
        n = min(t, 64) if t else 2                               # passes over the body
        states, out = set(entry), {}
        for trip in range(n):
            exits, latch = self.run_region(body, h, states, depth)
            for (src, dst), st in exits.items():
                if t and trip < n - 1 and src in latches:
                    continue                                     # a counted loop leaves through its latch after the LAST trip only
                out.setdefault((src, dst), set()).update(st)
            states = latch
            if not states:
                break
        return out

    def run(self):
        self.build_cfg()
        self.run_region(set(range(len(self.blocks))), 0, {()}, 0)


def check_text(text, want="", verbose=True, min_store_run=None):
    """-> (kernels checked, violations).  min_store_run: None or a function kernel name -> the number of stores the
    host guarantees a store-only run-time loop of that kernel to issue (0: no such guarantee)."""
    total = nk = 0
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@", text, re.M):
        name = m.group(1)
        if want not in name:
            continue
        end = text.find(".end_amdhsa_kernel", m.end())
        if end < 0:
            end = text.find("s_endpgm", m.end())
        items = []
        for i, raw in enumerate(text[m.end():end].split("\n")):
            l = raw.split(";")[0].strip()
            if not l or l.startswith("."):
                if l.startswith(".LBB") and l.endswith(":"):
                    items.append(("l", l[:-1]))
                continue
            if l.endswith(":"):
                items.append(("l", l[:-1]))
                continue
            items.append(("i", i, l))
        ins, labels = [], {}
        for it in items:
            if it[0] == "l":
                labels[it[1]] = len(ins)
            else:
                ins.append((it[1], it[2]))
        c = Checker(name[:70], ins, labels, verbose, min_store_run(name) if min_store_run else 0)
        c.run()
        total += c.bad
        nk += 1
    return nk, total


def mix_sru_store_run(name):
    """mix_sru_fwd_kernel<ACT, KX, LX, WB>: dmn_mix_sru_fwd picks WB <= L (csrc/mix_sru.hip), and the word loop
    issues one h store per word."""
    m = re.match(r"_ZN3dmn18mix_sru_fwd_kernelILi\d+ELi\d+ELi\d+ELi(\d+)EEE", name)
    return int(m.group(1)) if m else 0


def main():
    nk, total = check_text(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "", min_store_run=mix_sru_store_run)
    print("kernels:", nk, "violations:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
