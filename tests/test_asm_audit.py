"""CPU test (-m "not gpu"): audit of the generated gfx950 code for the wide-store hazard found in round 2.

gfx950 reads the data registers of a vector-memory store wider than 64 bits late: a VALU instruction that
rewrites one of them directly behind the store can win (seen as corrupted grad_U at training size,
csrc/sru_bwd.hip header of store_wide).  hipcc pads global_store_dwordx3/x4 itself but not a buffer store
with an SGPR soffset, so sru_bwd.hip issues those as asm statements with `s_nop 1` attached.  This test
compiles the sources that hold wide stores to assembly (hipcc cross-compiles without a GPU) and checks
that no instruction within two wait states behind a wide store writes one of its data registers."""
import os
import re
import subprocess

import pytest

from dms_amd import build as dbuild

CSRC = os.path.join(os.path.dirname(dbuild.__file__), "csrc")
WIDE = re.compile(r"^\s*(buffer|global|flat)_store_dwordx[34]\s+(.*)$")


def _regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def _instructions(path):
    for ln, raw in enumerate(open(path), 1):
        l = raw.split(";")[0].strip()
        if not l or l.startswith((".", "//")) or l.endswith(":"):
            continue
        yield ln, l


def audit(path, wait_states=2):
    """-> (number of wide stores, list of violations)"""
    ins = list(_instructions(path))
    nstores, bad = 0, []
    for i, (ln, l) in enumerate(ins):
        m = WIDE.match(l)
        if not m:
            continue
        nstores += 1
        ops = m.group(2).split(",")
        # buffer_store: data first; global/flat_store: address first, data second
        data = _regs(ops[0] if m.group(1) == "buffer" else ops[1])
        ws = 0
        for ln2, l2 in ins[i + 1:i + 1 + wait_states + 1]:
            if ws >= wait_states:
                break
            op = l2.split()[0]
            if op == "s_nop":
                ws += int(l2.split()[1], 0) + 1
                continue
            if op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                dst = _regs(l2[len(op):].split(",")[0])
                if dst & data:
                    bad.append("%s:%d `%s` rewrites v%s %d wait state(s) behind `%s` (line %d)" % (
                        os.path.basename(path), ln2, l2, sorted(dst & data), ws, l, ln))
            ws += 1
    return nstores, bad


def test_audit_flags_the_pattern_that_corrupted_grad_U(tmp_path):
    s = tmp_path / "bad.s"
    s.write_text("k:\n\tbuffer_store_dwordx3 v[12:14], v11, s[0:3], s48 offen nt\n\tv_add_f32_e32 v12, 1.0, v36\n"
                 "\tbuffer_store_dwordx4 v[0:3], v11, s[0:3], s48 offen nt\n\ts_nop 1\n\tv_add_f32_e32 v1, 1.0, v36\n"
                 "\tglobal_store_dwordx4 v[8:9], v[4:7], off\n\ts_nop 0\n\tv_mov_b32_e32 v7, 0\n")
    n, bad = audit(str(s))
    assert n == 3 and len(bad) == 2 and "v[12]" in bad[0] and "v[7]" in bad[1]


_ASM = {}


def _asm_of(src, tmp_path_factory):
    """hipcc -S of one source with the build's own flags (cached per test session)."""
    if src not in _ASM:
        out = str(tmp_path_factory.mktemp("asm") / (src + ".s"))
        cmd = [dbuild.HIPCC] + dbuild.FLAGS + dbuild.EXTRA_FLAGS.get(src, []) + [
            "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _ASM[src] = out
    return _ASM[src]


def _sregs(tok):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def audit_sgpr(path, need=5):
    """Vector-memory instructions INSIDE inline asm are invisible to hipcc's hazard recogniser: an SGPR
    (descriptor word, soffset) written by a VALU instruction (v_readfirstlane, v_readlane, v_cmp, carry-outs)
    must be `need` wait states old before such an instruction reads it.  -> (asm VMEM count, violations)"""
    ins, inasm = [], False
    for ln, raw in enumerate(open(path), 1):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        l = raw.split(";")[0].strip()
        if not l or l.startswith((".", "//")):
            continue
        ins.append((ln, "LABEL" if l.endswith(":") else l, inasm))
    n, bad = 0, []
    for i, (ln, l, ia) in enumerate(ins):
        if not ia or not l.split()[0].startswith(("buffer_", "global_", "flat_")):
            continue
        n += 1
        use, ws, j = _sregs(l), 0, i - 1
        while j >= 0 and ws < need:
            ln2, l2, _ = ins[j]
            if l2 == "LABEL":
                break
            op2 = l2.split()[0]
            if op2 == "s_nop":
                ws += int(l2.split()[1], 0) + 1
            else:
                if op2.startswith("v_"):
                    ops = l2[len(op2):].split(",")
                    two = "_co_" in op2 or op2.startswith(("v_cmp", "v_div_scale", "v_mad_u64", "v_mad_i64"))
                    dst = _sregs(",".join(ops[:2]) if two else ops[0])
                    if dst & use:
                        bad.append("%s:%d `%s` writes s%s %d wait state(s) before asm `%s` (line %d)" % (
                            os.path.basename(path), ln2, l2, sorted(dst & use), ws, l, ln))
                ws += 1
            j -= 1
    return n, bad


def test_sgpr_audit_flags_a_descriptor_written_by_valu_right_before_an_asm_load(tmp_path):
    s = tmp_path / "bad.s"
    s.write_text("k:\n\tv_readfirstlane_b32 s4, v1\n\ts_mov_b32 s6, 0\n\t;;#ASMSTART\n"
                 "\tbuffer_load_dword v0, v2, s[4:7], s9 offen nt\n\t;;#ASMEND\n"
                 "\tv_readfirstlane_b32 s9, v1\n\ts_nop 4\n\t;;#ASMSTART\n"
                 "\tbuffer_load_dword v0, v2, s[4:7], s9 offen nt\n\t;;#ASMEND\n")
    n, bad = audit_sgpr(str(s))
    assert n == 2 and len(bad) == 1 and "s[4]" in bad[0]


@pytest.mark.parametrize("src", ["sru_bwd.hip", "sru_fwd_k2.hip", "mix_sru.hip"])
def test_asm_memory_instructions_never_read_an_sgpr_a_valu_wrote_less_than_five_wait_states_ago(src, tmp_path_factory):
    if not os.path.exists(dbuild.HIPCC):
        pytest.skip("hipcc not installed")
    n, bad = audit_sgpr(_asm_of(src, tmp_path_factory))
    assert n > 0, "no inline-asm memory instructions found in %s" % src
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("src", ["sru_bwd.hip", "tapconv.hip"])
def test_no_valu_write_of_a_wide_stores_data_within_two_wait_states(src, tmp_path_factory):
    if not os.path.exists(dbuild.HIPCC):
        pytest.skip("hipcc not installed")
    out = _asm_of(src, tmp_path_factory)
    n, bad = audit(out)
    assert n > 0, "no wide stores found in %s: the audit is looking at the wrong thing" % src
    assert not bad, "\n".join(bad[:20])
