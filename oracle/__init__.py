"""CPU oracle for the SpaCap3D hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``spacap3d_amd`` (the product) never does.

Contents
--------
* ``pointnet2_oracle.c``  C restatement of the nine ``pointnet2._ext`` operators
  (reference: lib/pointnet2/_ext_src/src/*.cu); built by ``oracle/Makefile`` into
  ``oracle/_build/liboracle.so`` (canonical, single thread) and
  ``liboracle_omp.so`` (same loops, OpenMP -- multi-core CPU baseline).
* ``ext_cpu.py``          ``OracleExt``: the nine ``_ext`` entry points on CPU torch
  tensors, backed by the C library through ctypes.
* ``literal_py.py``       pure-Python thread-by-thread simulation of the CUDA blocks,
  small cases only; pins the C restatement's tie-break behaviour.
* ``attention_ref.py``    plain PyTorch fp32 restatement of ``attention()``
  (models/transformer_captioner.py:27-37) and of the relation feature
  (models/transformer_captioner.py:392-397) for the floating-point kernels.

Parity status: the reference has no golden vectors for FPS / ball_query / group /
gather / three_nn ("parity unpinned"); three_interpolate is pinned
by the inputs of lib/pointnet2/pointnet2_test.py:14-26.  There is no ``oracle/_ref``:
the reference's native path is CUDA-only and cannot be compiled in this image (no nvcc).
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
BUILD_DIR = os.path.join(_HERE, "_build")


def build(force: bool = False) -> None:
    """Compile the C restatement (gcc, seconds)."""
    libs = [os.path.join(BUILD_DIR, n) for n in ("liboracle.so", "liboracle_omp.so", "liboracle_fma.so")]
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    fresh = all(os.path.exists(p) and os.path.getmtime(p) >= os.path.getmtime(src) for p in libs)
    if fresh and not force:
        return
    subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True)


def lib_path(openmp: bool = False, fma: bool = False) -> str:
    if fma:   # nvcc-default contraction of the distance sums (OpenMP build); never the parity target
        return os.path.join(BUILD_DIR, "liboracle_fma.so")
    return os.path.join(BUILD_DIR, "liboracle_omp.so" if openmp else "liboracle.so")
