"""Build-time checks that need hipcc but no GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="no hipcc")
def test_compiler_never_touches_the_landing_registers_of_the_streaming_kernels():
    """csrc/sa_bf3.inc / sa_stream.inc prefetch rows into AGPRs a0..a63 through inline asm, invisibly to the register
    allocator; tools/check_landing_regs.py compiles sa_mlp.hip to assembly and fails on any compiler-generated use."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_landing_regs.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 unsafe uses" in r.stdout


def test_landing_register_checker_flags_planted_violations():
    """The checker itself: a compiler-generated write, and a read while the load is in flight, must be reported; the asm
    blocks' own uses and a clean function must not."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_landing_regs as C
    clean = """_ZN1x22sa_mid_fwd_bf3s_kernelILi128EEEv:                  ; @k
\t;;#ASMSTART
\tglobal_load_dwordx4 a[0:3], v[2:3], off
\t;;#ASMEND
\tv_mfma_f32_32x32x16_bf16 a[32:47], v[4:7], v[8:11], a[32:47]
\t;;#ASMSTART
\ts_waitcnt vmcnt(4)
\tv_accvgpr_read_b32 v6, a[0+0]
\t;;#ASMEND
\ts_endpgm
.Lfunc_end0:
"""
    seen, bad = C.check(clean)
    assert seen == 1 and not bad, bad
    spill = clean.replace("\tv_mfma_f32", "\tv_accvgpr_write_b32 a2, v9\n\tv_mfma_f32")
    seen, bad = C.check(spill)
    assert seen == 1 and len(bad) == 1 and "writes a landing register" in bad[0][2]
    copy = clean.replace("\tv_mfma_f32", "\tv_accvgpr_mov_b32 a40, a1\n\tv_mfma_f32")
    seen, bad = C.check(copy)
    assert len(bad) == 1 and "in flight" in bad[0][2]
    other = clean.replace("sa_mid_fwd_bf3s_kernel", "some_other_kernel")
    assert C.check(other.replace("\tv_mfma_f32", "\tv_accvgpr_write_b32 a2, v9\n\tv_mfma_f32"))[0] == 0
