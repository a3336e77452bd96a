#!/usr/bin/env python
"""Audit of hand-counted vmcnt waits in hipcc output (guide section 5.7, item 1).

Linear scan of a kernel's instructions: keeps the FIFO of outstanding VMEM operations;
`s_waitcnt vmcnt(N)` retires all but the N youngest; flags any instruction that READS or
WRITES a VGPR that is the destination of a still-outstanding load (e.g. a compiler v_mov
of an inline-asm load destination before its wait).  Loop bodies are scanned twice to
cover the back edge.  usage: check_asm_waits.py file.s [kernel-substring]
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


def scan(lines, name):
    fifo = []          # (kind, dest regs)
    bad = 0
    for ln, l in lines:
        op = l.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                while len(fifo) > n:
                    fifo.pop(0)
            continue
        args = l[len(op):]
        parts = [a.strip() for a in args.split(",")]
        if op.startswith(("buffer_load", "global_load")):
            touched = regs(",".join(parts[1:]))             # address operands are read
            pend = set().union(*[d for k, d in fifo if k == "L"]) if fifo else set()
            if touched & pend:
                print("%s:%d reads pending %s : %s" % (name, ln, sorted(touched & pend), l)); bad += 1
            fifo.append(("L", regs(parts[0])))
            continue
        if op.startswith(("buffer_store", "global_store")):
            touched = regs(args)
            pend = set().union(*[d for k, d in fifo if k == "L"]) if fifo else set()
            if touched & pend:
                print("%s:%d store touches pending %s : %s" % (name, ln, sorted(touched & pend), l)); bad += 1
            fifo.append(("S", set()))
            continue
        if op.startswith(("s_", ";")):
            continue
        touched = regs(args)
        pend = set().union(*[d for k, d in fifo if k == "L"]) if fifo else set()
        if touched & pend:
            print("%s:%d touches pending %s : %s" % (name, ln, sorted(touched & pend), l)); bad += 1
    return bad


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    total = 0
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@", text, re.M):
        name = m.group(1)
        if want not in name:
            continue
        end = text.find(".end_amdhsa_kernel", m.end())
        body = text[m.end():end].split("\n")
        ins = []
        for i, l in enumerate(body):
            l = l.split(";")[0].strip()
            if not l or l.startswith(".") or l.endswith(":"):
                continue
            ins.append((i, l))
        total += scan(ins + ins, name[:60])
    print("violations:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
