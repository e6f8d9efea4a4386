"""The build-time gate of george_amd/csrc/Makefile (check_kernels.py): kernels whose LDS reads are inline assembly with
hand-counted waits (GH_SP_READ8 / GH_SP_WAIT, gh_gemm_tile.h) must not use scratch memory, and no instruction of any
kernel may touch a register an LDS read still owes data to.  CPU only: hipcc cross-compiles gfx950 without a GPU."""
import importlib.util
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "george_amd", "csrc")

spec = importlib.util.spec_from_file_location("check_kernels", os.path.join(CSRC, "check_kernels.py"))
ck = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ck)


def test_the_makefile_links_only_behind_the_gate():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "-Rpass-analysis=kernel-resource-usage" in mk and "--save-temps" in mk
    assert "check_kernels.py $(B)" in mk
    # the library's link rule depends on the gate's stamp
    assert "$(TARGET): $(OBJS) $(B)/kernel_gate.ok" in mk


def test_the_product_build_passes_and_covers_the_inline_asm_kernels():
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)     # (no-op when the tree is built)
    rc = subprocess.call([sys.executable, os.path.join(CSRC, "check_kernels.py"), os.path.join(CSRC, "build")],
                         stdout=subprocess.DEVNULL)
    assert rc == 0
    rep = open(os.path.join(CSRC, "build", "kernel_gate.txt")).read()
    covered = [l for l in rep.splitlines() if l.startswith("#   ")]
    names = " ".join(covered)
    # the roofline kernel (both triangle modes) and the HODLR leaf products (both accumulate modes)
    assert names.count("gemm_f64_mfma_dma_sp") == 2 and names.count("hodlr_bmm_nt_kernel") == 2, covered
    for l in rep.splitlines():
        if "[inline LDS reads]" in l:
            assert " ok " in l and "scratch=0" in l, l


def test_the_gate_refuses_a_deliberately_spilled_variant(tmp_path):
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "tests", "kernel_gate", "spilled_variant.hip")
    rem = tmp_path / "spilled_variant.remarks"
    with open(rem, "w") as f:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                               "--save-temps=obj", "-c", src, "-o", str(tmp_path / "spilled_variant.o")], stderr=f, cwd=str(tmp_path))
    asm = [p for p in os.listdir(tmp_path) if p.startswith("spilled_variant-hip-amdgcn") and p.endswith(".s")]
    assert asm
    res = ck.parse_remarks(str(rem))
    (name, r), = res.items()
    assert r["scratch"] > 0 and r["vgpr_spill"] > 0               # the variant does what it was written to do
    out = subprocess.run([sys.executable, os.path.join(CSRC, "check_kernels.py"), "--asm", str(tmp_path / asm[0]), "--remarks", str(rem)],
                         capture_output=True, text=True)
    assert out.returncode == 1
    assert "uses scratch memory" in out.stdout
    # and the mechanism itself is in the assembly: a read's destination goes to scratch before the wait that guards it
    assert "scratch_store" in out.stdout and "while an LDS read into it is outstanding" in out.stdout


ASM_OK = """
\tds_read_b128 v[10:13], v2 offset:0
\tds_read_b128 v[14:17], v2 offset:2048
\tv_add_f64 v[20:21], v[22:23], v[24:25]
\ts_waitcnt lgkmcnt(1)
\tv_add_f64 v[20:21], v[10:11], v[12:13]
\ts_waitcnt lgkmcnt(0)
\tv_add_f64 v[20:21], v[14:15], v[16:17]
\ts_endpgm
"""
ASM_EARLY_USE = ASM_OK.replace("s_waitcnt lgkmcnt(1)", "s_nop 0")
ASM_LOOP_CARRIED = """
\tds_read_b64 v[4:5], v2
.LBB0_1:
\ts_waitcnt lgkmcnt(0)
\tv_add_f64 v[6:7], v[4:5], v[6:7]
\tds_read_b64 v[4:5], v2
\ts_cbranch_scc1 .LBB0_1
\tv_mov_b32_e32 v8, v4
\ts_endpgm
"""
ASM_SMEM_BLOCKS = """
\tds_read_b64 v[4:5], v2
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\ts_waitcnt lgkmcnt(1)
\tv_mov_b32_e32 v8, v4
\ts_endpgm
"""


@pytest.mark.parametrize("asm,nbad", [(ASM_OK, 0), (ASM_EARLY_USE, 1), (ASM_LOOP_CARRIED, 1), (ASM_SMEM_BLOCKS, 1)])
def test_lint_on_synthetic_streams(asm, nbad):
    """in-order retirement by lgkmcnt(k); a use before the wait; a read left outstanding on the loop's exit path; a scalar
    load in flight (out-of-order) makes lgkmcnt(1) retire nothing"""
    bad = ck.lint("k", asm.splitlines())
    assert len(bad) == nbad, bad
