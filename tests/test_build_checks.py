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
