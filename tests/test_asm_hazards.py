"""CPU: the compiled gfx950 code of the hand-scheduled MFMA kernels never touches a register whose hand-issued
ds_read may still be in flight (tools/check_asm_hazards.py explains the failure mode).  Needs hipcc only."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(ROOT, "tools", "check_asm_hazards.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def test_scanner_flags_a_copy_of_a_pending_read():
    code = [";;#ASMSTART", "ds_read_b128 v[10:13], v1 offset:0", "ds_read_b128 v[14:17], v1 offset:1024", ";;#ASMEND",
            "s_waitcnt lgkmcnt(1)",
            "v_mov_b64_e32 v[20:21], v[10:11]",        # first read retired: fine
            "v_mov_b64_e32 v[22:23], v[14:15]",        # second read still pending: hazard
            "s_waitcnt lgkmcnt(0)", "v_mov_b64_e32 v[24:25], v[16:17]"]
    found = chk.scan_function(code)
    assert [f[0] for f in found] == [6]
    # a compiler-issued read (outside an asm block) is hipcc's own business
    assert chk.scan_function(["ds_read_b128 v[10:13], v1", "v_mov_b32_e32 v20, v10"]) == []
    # hand-issued global loads are tracked against vmcnt
    glob = [";;#ASMSTART", "global_load_dwordx4 v[4:7], v9, s[2:3]", ";;#ASMEND", "v_mov_b32_e32 v30, v5", "s_waitcnt vmcnt(0)",
            "v_mov_b32_e32 v31, v6"]
    assert [f[0] for f in chk.scan_function(glob)] == [3]


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src", chk.DEFAULT)
def test_no_inflight_register_use(src):
    res = chk.scan_file(os.path.join(chk.CSRC, src))
    assert res, "no kernels found"
    if src == "conv_gemm4.hip":      # its timing-only ablation builds (DBG != 0) replace fragment reads by dummies: not product code
        res = {k: v for k, v in res.items() if "ELi0ELi0EEE" in k or "ELi0ELi1EEE" in k}    # <STAMP, DBG = 0, TAG>
        assert res
    bad = {k: v[:3] for k, v in res.items() if v}
    assert not bad, bad


def test_asm_mfma_scanner():
    code = [";;#ASMSTART", "v_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[4:7], a[0:3]", ";;#ASMEND",
            "v_accvgpr_read_b32 v9, a0",               # one instruction after the MFMA that writes a0: stale
            "s_add_i32 s1, s1, 1", "v_mov_b32_e32 v20, v21", "v_cndmask_b32_e32 v4, v8, v9, vcc",
            ";;#ASMSTART", "v_mfma_f32_16x16x32_f16 a[4:7], v[0:3], v[4:7], a[4:7]", ";;#ASMEND", "v_accvgpr_read_b32 v10, a4"]
    assert [t.split()[0] for _, t in chk.scan_asm_mfma_region(code)] == ["v_accvgpr_read_b32", "v_cndmask_b32_e32", "v_accvgpr_read_b32"]
    assert chk.scan_asm_mfma_region(["v_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[4:7], a[0:3]", "v_accvgpr_read_b32 v9, a0"]) == []
    fenced = [";;#ASMSTART", "v_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[4:7], a[0:3]", ";;#ASMEND", "s_nop 15", "s_nop 15", "v_accvgpr_read_b32 v9, a0"]
    assert chk.scan_asm_mfma_region(fenced) == []


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_gemm4_loop_has_no_compiler_valu():
    """conv_gemm4.hip writes its MFMAs as asm: between the first and the last of them the product build must hold no
    accumulator traffic and no compiler-generated VALU instruction (both were real bugs: see the kernel's comments)."""
    res = chk.scan_file(os.path.join(chk.CSRC, "conv_gemm4.hip"), scan=chk.scan_asm_mfma_region)
    prod = {k: v for k, v in res.items() if "Lb0ELi0ELi" in k}     # <false, 0, TAG>
    assert prod, list(res)
    assert all(not v for v in prod.values()), {k: v[:4] for k, v in prod.items() if v}


def test_scanner_follows_branches_to_labels_above_and_counts_every_vmem_instruction():
    """(a) a hand-issued load pending at a branch to a label ABOVE (a loop back-edge, or hipcc placing a later block earlier in
    the text) is carried to that label on the next pass; (b) stores and LDS-DMA pieces take places in the vmcnt queue;
    (c) a 128-bit asm store needs `s_nop 1` behind it."""
    code = ["s_branch .LBB0_2",
            ".LBB0_1:", "v_mov_b32_e32 v20, v4",                       # reached from below with v[4:7] still in flight
            "s_waitcnt vmcnt(0)", "v_mov_b32_e32 v21, v5", "s_endpgm",
            ".LBB0_2:", ";;#ASMSTART", "global_load_dwordx4 v[4:7], v[8:9], off", ";;#ASMEND", "s_branch .LBB0_1"]
    assert [f[0] for f in chk.scan_function(code)] == [2]
    q = [";;#ASMSTART", "global_load_dwordx4 v[4:7], v[8:9], off", ";;#ASMEND",
         "global_load_lds_dwordx4 v10, s[2:3]",                         # an LDS-DMA piece: one place in the queue
         ";;#ASMSTART", "global_store_dwordx4 v[8:9], v[12:15], off", "s_nop 1", ";;#ASMEND",
         "s_waitcnt vmcnt(2)", "v_mov_b32_e32 v30, v4",                 # the load is the third youngest: retired
         ";;#ASMSTART", "global_load_dwordx4 v[16:19], v[8:9], off", ";;#ASMEND",
         "s_waitcnt vmcnt(1)", "v_mov_b32_e32 v31, v16"]                # youngest of all: still pending
    assert [f[0] for f in chk.scan_function(q)] == [14]
    bad_store = [";;#ASMSTART", "global_store_dwordx4 v[8:9], v[12:15], off", ";;#ASMEND", "v_mov_b32_e32 v12, v1"]
    assert len(chk.scan_function(bad_store)) == 1
