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
_ASM_SOURCES = ["sru_bwd.hip", "sru_fwd_k2.hip", "sru_fwd_k3.hip", "sru_fwd_k4.hip", "mix_sru.hip", "mix.hip", "tapconv.hip", "dynfilter.hip"]


def _asm_of(src, tmp_path_factory):
    """hipcc -S of one source with the build's own flags.  The first call compiles every audited source at once
    (a minute each, in parallel); the results are cached for the test session."""
    if src not in _ASM:
        d = tmp_path_factory.mktemp("asm")
        todo = [s for s in dict.fromkeys(_ASM_SOURCES + [src]) if s not in _ASM]
        procs = []
        for s in todo:
            out = str(d / (s + ".s"))
            cmd = [dbuild.HIPCC] + dbuild.FLAGS + dbuild.EXTRA_FLAGS.get(s, []) + [
                "-S", "--cuda-device-only", os.path.join(CSRC, s), "-o", out]
            procs.append((s, out, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
        for s, out, p in procs:
            assert p.wait() == 0, "hipcc -S failed on %s" % s
            _ASM[s] = out
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


@pytest.mark.parametrize("src", ["sru_bwd.hip", "sru_fwd_k2.hip", "sru_fwd_k3.hip", "sru_fwd_k4.hip", "mix_sru.hip", "mix.hip", "dynfilter.hip"])
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


# ------------------------------------------------------------------------------------------------------------
# counted waits of the register rings (tools/check_asm_waits.py)
# ------------------------------------------------------------------------------------------------------------
def _waits_checker():
    import importlib.util
    path = os.path.join(os.path.dirname(CSRC), "..", "tools", "check_asm_waits.py")
    spec = importlib.util.spec_from_file_location("check_asm_waits", os.path.abspath(path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_wait_checker_flags_a_ring_register_touched_before_its_counted_wait_and_honours_branches_and_op_sel():
    chk = _waits_checker()
    head = "_Z1kv: ; @_Z1kv\n"
    tail = "\ts_endpgm\n.end_amdhsa_kernel\n"
    # v_mov of a load destination while the load is in flight (what hipcc did with "=v" slots)
    bad = head + "\tbuffer_load_dword v4, v1, s[0:3], s8 offen\n\tv_mov_b32_e32 v9, v4\n\ts_waitcnt vmcnt(0)\n" + tail
    assert chk.check_text(bad, verbose=False) == (1, 1)
    # a wait that leaves exactly the younger load outstanding is fine; one count too high is not
    ok = head + "\tbuffer_load_dword v4, v1, s[0:3], s8 offen\n\tbuffer_load_dword v5, v1, s[0:3], s9 offen\n" \
         "\ts_waitcnt vmcnt(1)\n\tv_add_f32_e32 v6, v4, v4\n\ts_waitcnt vmcnt(0)\n" + tail
    assert chk.check_text(ok, verbose=False) == (1, 0)
    assert chk.check_text(ok.replace("vmcnt(1)", "vmcnt(2)"), verbose=False) == (1, 1)
    # the two sides of an if / else are exclusive: a load on one side does not poison a write on the other
    br = head + "\ts_cbranch_scc1 .LBB0_2\n\tglobal_load_dword v4, v[2:3], off\n\ts_branch .LBB0_3\n.LBB0_2:\n" \
         "\tv_mov_b32_e32 v4, 0\n.LBB0_3:\n\ts_waitcnt vmcnt(0)\n\tv_add_f32_e32 v6, v4, v4\n" + tail
    assert chk.check_text(br, verbose=False) == (1, 0)
    # op_sel_hi:[1,0,1] reads only the LOW half of the second source pair
    pk = head + "\tbuffer_load_dword v3, v1, s[0:3], s8 offen\n\tv_pk_fma_f32 v[4:5], v[6:7], v[2:3], v[8:9] op_sel_hi:[1,0,1]\n" \
         "\ts_waitcnt vmcnt(0)\n" + tail
    assert chk.check_text(pk, verbose=False) == (1, 0)
    assert chk.check_text(pk.replace(" op_sel_hi:[1,0,1]", ""), verbose=False) == (1, 1)
    # a counted loop is expanded to its stated trip count: 20 stores behind the load, vmcnt(20) retires it
    loop = head + "\tbuffer_load_dword v4, v1, s[0:3], s8 offen\n\ts_mov_b32 s5, 20\n.LBB0_1:\n" \
           "\tbuffer_store_dword v7, v1, s[0:3], s9 offen\n\ts_add_i32 s5, s5, -1\n\ts_cmp_lg_u32 s5, 0\n" \
           "\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt vmcnt(20)\n\tv_add_f32_e32 v6, v4, v4\n" + tail
    assert chk.check_text(loop, verbose=False) == (1, 0)
    assert chk.check_text(loop.replace("vmcnt(20)", "vmcnt(21)"), verbose=False) == (1, 1)
    # ... also when it counts UP in steps of 2 to a limit (what hipcc makes of `#pragma unroll 2` over a constant range)
    up = head + "\tbuffer_load_dword v4, v1, s[0:3], s8 offen\n\ts_mov_b32 s5, 0\n.LBB0_1:\n" \
         "\tbuffer_store_dword v7, v1, s[0:3], s9 offen\n\tbuffer_store_dword v8, v1, s[0:3], s9 offen\n" \
         "\ts_add_i32 s5, s5, 2\n\ts_cmp_eq_u32 s5, 8\n\ts_cbranch_scc0 .LBB0_1\n\ts_waitcnt vmcnt(8)\n\tv_add_f32_e32 v6, v4, v4\n" + tail
    assert chk.check_text(up, verbose=False) == (1, 0)
    assert chk.check_text(up.replace("vmcnt(8)", "vmcnt(9)"), verbose=False) == (1, 1)
    # the rare arm of an if / else parked behind the main code jumps BACK: an ordinary edge, not a loop -- the load on
    # the main path (younger than the jump's target) must not be seen by the parked block, and vice versa
    cold = head + "\ts_cbranch_scc1 .LBB0_9\n.LBB0_1:\n\tv_mov_b32_e32 v5, 0\n\tglobal_load_dword v4, v[2:3], off\n" \
           "\ts_waitcnt vmcnt(0)\n\tv_add_f32_e32 v6, v4, v5\n\ts_endpgm\n.LBB0_9:\n\tv_mov_b32_e32 v4, 0\n\ts_branch .LBB0_1\n" + tail
    assert chk.check_text(cold, verbose=False) == (1, 0)
    # a store-only loop with a run-time trip count: only the host's promise (>= 8 stores) makes vmcnt(8) enough
    rt = head + "\tbuffer_load_dword v4, v1, s[0:3], s8 offen\n.LBB0_1:\n\tbuffer_store_dword v7, v1, s[0:3], s9 offen\n" \
         "\ts_add_i32 s5, s5, 1\n\ts_cmp_lt_i32 s5, s6\n\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt vmcnt(8)\n\tv_add_f32_e32 v6, v4, v4\n" + tail
    assert chk.check_text(rt, verbose=False) == (1, 1)
    assert chk.check_text(rt, verbose=False, min_store_run=lambda name: 8) == (1, 0)
    assert chk.check_text(rt, verbose=False, min_store_run=lambda name: 7) == (1, 1)


@pytest.mark.parametrize("src,kernels", [("sru_bwd.hip", ""), ("sru_fwd_k2.hip", ""), ("sru_fwd_k3.hip", ""),
                                         ("sru_fwd_k4.hip", ""), ("mix.hip", "mix_bwd_one_kernel"), ("mix_sru.hip", "")])
def test_no_ring_register_is_touched_before_its_counted_wait(src, kernels, tmp_path_factory):
    """The scans' register rings are loaded by inline asm that hipcc does not count; every slot is waited for with
    a hand-counted s_waitcnt vmcnt(N).  The invariant -- no instruction reads or writes a slot register while its
    load is in flight -- depends on hipcc's register allocation and on -fno-slp-vectorize for sru_bwd.hip, so it
    is checked on the generated code of every instantiation -- the fused mix + scan kernels included: in round 3 the
    checker learned their control flow (out-of-line arms, count-up loops) and found hipcc moving the prefetched A
    registers of mix_sru_fwd_kernel<.., 10, 0, 8|20> in front of their counted wait (fixed in the source: ordered asm
    moves).  One promise comes from the host, not from the code: mix_sru_fwd_kernel<.., WB> runs at least WB words
    (`mix_sru_store_run`)."""
    if not os.path.exists(dbuild.HIPCC):
        pytest.skip("hipcc not installed")
    chk = _waits_checker()
    nk, bad = chk.check_text(open(_asm_of(src, tmp_path_factory)).read(), kernels, verbose=False,
                             min_store_run=chk.mix_sru_store_run)
    assert nk > 0 and bad == 0, "%s: %d violation(s) in %d kernels (run tools/check_asm_waits.py on the .s)" % (src, bad, nk)


def test_hipcc_itself_counts_a_younger_store_as_still_outstanding_behind_an_older_load(tmp_path):
    """The vmcnt policy of the library (csrc/sru_ring.h): loads AND stores retire in order on the one counter.
    That is hipcc's own model for gfx9-family targets -- for load; store; use-of-load it emits vmcnt(1), which
    would be wrong if the store could retire first -- and every compiler-scheduled kernel depends on it."""
    if not os.path.exists(dbuild.HIPCC):
        pytest.skip("hipcc not installed")
    src = os.path.join(os.path.dirname(CSRC), "..", "tools", "probe", "vmcnt_order.hip")
    out = str(tmp_path / "v.s")
    subprocess.run([dbuild.HIPCC, "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", os.path.abspath(src), "-o", out],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ins = [l for _, l in _instructions(out)]
    i_load = next(i for i, l in enumerate(ins) if l.startswith("global_load_dword"))
    i_store = next(i for i, l in enumerate(ins) if l.startswith("global_store_dword"))
    waits = [l for l in ins[i_store:] if l.startswith("s_waitcnt") and "vmcnt" in l]
    assert i_load < i_store and waits and "vmcnt(1)" in waits[0]
