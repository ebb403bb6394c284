"""The built libraries must pass tools/check_isa.py: no instruction form known to miscompute on gfx950, and the hand-scheduled
inline-asm sequences in the shape their correctness depends on.

PK-OPSEL: packed-fp32 ops whose low lane selects a source's high dword, inside a kernel with MFMAs, returned 0 in lanes 48-63
under two waves per SIMD -- the root cause of round 1's "fp16 conversion" failure.  LDS-DMA-M0: every `global_load_lds` right
behind its own `s_mov_b32 m0` (the compiler does not know the asm writes m0).  COUNTED-WAIT: between two marked asm
`global_load_dword` and their hand-counted `s_waitcnt vmcnt(N)` exactly N vector loads, no store, no touch of the destination
registers (round 2's pass_c_mfma_kernel had such a sequence; round 3 replaced it by plain loads, the rule stays).  The compiler creates / could break all three by itself, so the check is on the shipped code objects,
not the source -- and each rule is shown here to fire on a patched listing."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_isa():
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _need_objdump(ci):
    if ci.OBJDUMP is None or os.environ.get("MTMC_SKIP_ISA_LINT"):
        pytest.skip("llvm-objdump not available / lint switched off")


def test_shipped_library_passes_isa_lint():
    from mtmc_mpn import _lib
    ci = _check_isa()
    _need_objdump(ci)
    assert os.path.exists(_lib.LIB_PATH), "library not built"
    kernels = list(ci.kernels_of(_lib.LIB_PATH))
    assert len(kernels) > 40 and any("gemm_bn_f16x3" in n for n, _ in kernels)
    assert ci.violations_in(kernels) == []
    # the LDS-DMA rule had something to look at: the two pre-split GEMMs.  (Round 3 rewrote pass_c_mfma_kernel as a software
    # pipeline of plain loads, so the product has no hand-counted wait left; the COUNTED-WAIT rule stays for any kernel
    # that marks one, and is exercised on patched listings below.)
    dma = {n for n, b in kernels for i in b if i.startswith("global_load_lds_")}
    assert any("gemm_f16p_m16" in n for n in dma) and any("gemm_staged" in n for n in dma)
    assert not [n for n, b in kernels for i in b if ci.MARK_PAIR in i or ci.MARK_WAIT in i]


def test_product_library_ships_no_laboratory_kernels_and_no_scratch():
    """libmtmc_mpn.so holds product kernels only: the pre-split GEMM's A/B variants and the timing experiments with wrong
    results live in libmtmc_lab.so; and no product kernel spills to scratch memory."""
    from mtmc_mpn import _lib
    ci = _check_isa()
    _need_objdump(ci)
    names = [n for n, _ in ci.kernels_of(_lib.LIB_PATH)]
    for lab_only in ("gemm_f16p_kernel", "gemm_f16p_mid_kernel", "gemm_f16p_pp_kernel"):
        assert not any(lab_only in n for n in names), lab_only
    assert any("gemm_f16p_m16_kernel" in n for n in names)
    import ctypes
    lib = ctypes.CDLL(_lib.LIB_PATH)
    assert not hasattr(lib, "mtmc_lab_linear_presplit_raw")
    for n, body in ci.kernels_of(_lib.LIB_PATH):
        assert not any(i.startswith(("scratch_", "buffer_store_dword v", "buffer_load_dword v")) and "off" in i and "s[0:3]" in i
                       for i in body), f"{n} uses scratch memory"
    import lab_lib                                        # (the laboratory's loader lives with the tests, not in the package)
    assert not hasattr(_lib, "load_lab") and not hasattr(_lib, "LAB_PATH")
    if os.path.exists(lab_lib.LAB_PATH):
        lab = [n for n, _ in ci.kernels_of(lab_lib.LAB_PATH)]
        assert any("gemm_f16p_mid_kernel" in n for n in lab)
        assert ci.violations(lab_lib.LAB_PATH) == []


def test_lint_rule_matches_the_failing_form():
    ci = _check_isa()
    assert ci.PK.search("v_pk_mul_f32 v[42:43], v[8:9], v[38:39] op_sel:[0,1]")
    m = ci.OPSEL_LO.search("v_pk_fma_f32 v[38:39], v[8:9], v[38:39], v[52:53] op_sel:[0,1,0] neg_lo:[0,0,1]")
    assert m and "1" in m.group(1)
    assert ci.OPSEL_LO.search("v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[0,1]") is None   # measured safe
    body = ["v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]", "v_pk_fma_f32 v[38:39], v[8:9], v[38:39], v[52:53] op_sel:[0,1,0]"]
    assert [r for r, _, _ in ci.violations_in([("k", body)])] == ["PK-OPSEL"]
    assert ci.violations_in([("k", body[1:])]) == []                  # no MFMA in the kernel: the form is harmless


GOOD_DMA = ["s_add_i32 s17, s25, 0x1000", "s_mov_b32 m0, s17", "s_nop 0", "global_load_lds_dwordx4 v54, s[20:21]"]


def test_lds_dma_rule_on_patched_listings():
    ci = _check_isa()
    assert ci.violations_in([("k", GOOD_DMA)]) == []
    # the compiler (or an edit) put an m0 reader / writer / anything between the s_mov_b32 m0 and its DMA instruction
    for intruder in ("s_mov_b32 s5, m0", "v_movrels_b32_e32 v1, v2", "s_add_i32 s4, s4, 1", "ds_write_b32 v1, v2"):
        bad = GOOD_DMA[:3] + [intruder] + GOOD_DMA[3:]
        v = ci.violations_in([("k", bad)])
        assert [r for r, _, _ in v] == ["LDS-DMA-M0"], intruder
    # two DMA instructions sharing one m0 write
    v = ci.violations_in([("k", GOOD_DMA + ["global_load_lds_dwordx4 v55, s[20:21]"])])
    assert [r for r, _, _ in v] == ["LDS-DMA-M0"]
    assert ci.violations_in([("k", ["global_load_lds_dwordx4 v55, s[20:21]"])])        # first instruction of a function


GOOD_WAIT = ["s_mov_b32 s30, 0xc0de0001",
             "global_load_dword v92, v[2:3], off", "global_load_dword v91, v[4:5], off",
             "v_add_u32_e32 v7, 1, v7",
             "global_load_dwordx4 v[40:43], v[8:9], off", "global_load_dword v86, v[10:11], off",
             "v_fma_f32 v50, v40, v41, v42", "s_mov_b32 s30, 0xc0de0002", "s_waitcnt vmcnt(2)",
             "global_store_dwordx2 v[12:13], v[50:51], off", "v_fma_f32 v60, v92, v91, v60"]


def test_counted_wait_rule_on_patched_listings():
    ci = _check_isa()
    assert ci.violations_in([("k", GOOD_WAIT)]) == []

    def patched(at, ins, drop=0):
        return GOOD_WAIT[:at] + [ins] + GOOD_WAIT[at + drop:]
    cases = {
        "a third load behind the pair": patched(6, "global_load_dword v87, v[10:11], off"),
        "a store before the wait": patched(6, "global_store_dword v[12:13], v50, off"),
        "an atomic before the wait": patched(6, "global_atomic_add_f32 v[12:13], v50, off"),
        "a copy of a destination register": patched(3, "v_mov_b32_e32 v70, v92"),
        "a destination register inside a range operand": patched(3, "v_pk_mul_f32 v[70:71], v[90:91], v[70:71]"),
        "a destination register overwritten": patched(3, "v_mov_b32_e32 v91, 0"),
        "the wait with another count": patched(8, "s_waitcnt vmcnt(1)", drop=1),
        "a branch whose target is unknown": patched(3, "s_cbranch_vccnz 12"),
        "a load behind a branch (not on every path)": GOOD_WAIT[:4] + ["s_cbranch_scc1 2"] + GOOD_WAIT[4:],
        "a way out of the region": patched(3, "s_endpgm"),
        "a prefetch load missing": GOOD_WAIT[:5] + GOOD_WAIT[6:],
    }
    for what, body in cases.items():
        v = ci.violations_in([("k", body)])
        assert v and all(r == "COUNTED-WAIT" for r, _, _ in v), what
    # markers out of order / unpaired
    assert ci.violations_in([("k", GOOD_WAIT[7:])])
    assert ci.violations_in([("k", GOOD_WAIT[:7])])
    # with addresses (as in a real listing): branches inside the region are fine, one that leaves or enters it is not
    def listing(extra_before=(), region_branch_target=None, outside_target=None):
        lines, addr = ["0000000000001000 <k>:"], 0x1000
        def emit(text, target=None):
            nonlocal addr
            tail = f" <k+0x{target - 0x1000:X}>" if target is not None else ""
            lines.append(f"\t{text}    // {addr:012X}: BF800000{tail}")
            addr += 4
        if outside_target is not None:
            emit("s_cbranch_scc0 9", outside_target)
        for t in GOOD_WAIT[:6]:
            emit(t)
        if region_branch_target is not None:
            emit("s_cbranch_scc1 3", region_branch_target)
        for t in GOOD_WAIT[6:]:
            emit(t)
        return list(ci.functions_of("\n".join(lines)))
    assert ci.violations_in(listing()) == []
    assert ci.violations_in(listing(region_branch_target=0x1000 + 4 * 7)) == []        # skips one VALU op, stays inside
    v = ci.violations_in(listing(region_branch_target=0x1000 + 4 * 12))               # jumps past the wait
    assert v and "leaves the region" in v[0][2]
    v = ci.violations_in(listing(outside_target=0x1000 + 4 * 6))                      # lands behind the pair's loads
    assert v and "enters the region" in v[0][2]


def test_objdump_is_found_without_a_hard_coded_path(monkeypatch, tmp_path):
    ci = _check_isa()
    fake = tmp_path / "llvm-objdump"
    fake.write_text("#!/bin/sh\n")
    fake.chmod(0o755)
    monkeypatch.setenv("MTMC_OBJDUMP", str(fake))
    assert ci.find_objdump() == str(fake)
    monkeypatch.delenv("MTMC_OBJDUMP")
    root = tmp_path / "rocm-9.9"
    (root / "lib" / "llvm" / "bin").mkdir(parents=True)
    tool = root / "lib" / "llvm" / "bin" / "llvm-objdump"
    tool.write_text("#!/bin/sh\n")
    monkeypatch.setenv("ROCM_PATH", str(root))
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("PATH", "/nonexistent")
    found = ci.find_objdump()
    assert found in (str(tool), "/opt/rocm/lib/llvm/bin/llvm-objdump")      # a versioned ROCm root, or this image's default
