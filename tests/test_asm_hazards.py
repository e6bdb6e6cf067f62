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
    bad = {k: v[:3] for k, v in res.items() if v}
    assert not bad, bad
