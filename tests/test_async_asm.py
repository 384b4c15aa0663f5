"""Build-time guard for the hand-pipelined kernels (tools/check_async_asm.py): no instruction of the emitted gfx950 code may read
or write the destination registers of an inline-asm global load before the next `s_waitcnt vmcnt(0)` on any path.  The compiler does
not know those registers are in flight; round 3 found it (a) copying them across a control-flow merge after an unrelated edit
(garbage results, caught by the parity tests) and (b) hoisting register copies above the wait in a kernel variant nobody had seen
fail (a race that only shows when a load takes longer than an MFMA block).  Compiles csrc/sparse_conv.hip to assembly: ~1 minute."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import check_async_asm as chk  # noqa: E402


def test_checker_sees_a_copy_before_the_wait():
    bad = """
kern:
.LBB0_1:
\t;;#ASMSTART
\tglobal_load_dwordx4 v[4:7], v[2:3], off
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_3
.LBB0_2:
\tv_mov_b32_e32 v10, v4
.LBB0_3:
\ts_waitcnt vmcnt(0)
\tv_add_f32_e32 v11, v4, v5
\ts_endpgm
"""
    v = chk.check_asm(bad)
    assert len(v) == 1 and "v_mov_b32_e32 v10, v4" in v[0][2]
    good = bad.replace("\tv_mov_b32_e32 v10, v4\n", "\tv_mov_b32_e32 v10, v8\n")
    assert chk.check_asm(good) == []
    # a compiler-inserted partial wait promises nothing for loads it does not count
    partial = bad.replace("\ts_waitcnt vmcnt(0)\n", "\ts_waitcnt vmcnt(1)\n")
    assert len(chk.check_asm(partial)) == 2


def test_no_kernel_touches_an_asm_loaded_register_before_its_wait():
    import subprocess
    import tempfile
    src = os.path.join(REPO, "from-voxel-to-point_amd", "csrc", "sparse_conv.hip")
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + chk.FLAGS + ["-I" + os.path.dirname(src), src, "-o", asm], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    assert text.count(";;#ASMSTART") > 100          # the asm pipelines are in the file that was checked
    v = chk.check_asm(text)
    assert not v, "\n".join(f"{k}: line {no}: {ins} (load at {at})" for k, no, ins, at in v[:20])
