"""Thread-by-thread pure-Python simulation of the reference CUDA blocks.

TEST INFRASTRUCTURE, small cases only (seconds at n <= ~2000).  Where
``pointnet2_oracle.c`` re-orders loops for speed (it walks points in index order and
keeps one (best, besti) pair per simulated thread), this file keeps the shape of the
CUDA program -- a loop over threads, each walking its own strided points, then the
shared-memory tree -- so that the C restatement's FPS tie-break and skip rule are
checked against an independent, more literal reading of
lib/pointnet2/_ext_src/src/sampling_gpu.cu:69-173 and ball_query_gpu.cu:9-44.

All arithmetic goes through numpy float32 scalars so every operation rounds to fp32
exactly once (no FMA, no double intermediates), as the un-contracted source reads.
"""
import math

import numpy as np

f32 = np.float32


def opt_n_threads(work_size: int) -> int:
    # include/cuda_utils.h:15-19
    pow_2 = int(math.log(float(work_size)) / math.log(2.0))
    return max(min(1 << pow_2, 512), 1)


def fps_literal(xyz: np.ndarray, m: int) -> np.ndarray:
    """xyz (n,3) float32 -> (m,) int32, one scene."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    bs = opt_n_threads(n)
    temp = np.full(n, 1e10, dtype=np.float32)  # sampling.cpp:74-76
    idxs = np.zeros(m, dtype=np.int32)
    if m <= 0:
        return idxs
    old = 0
    for j in range(1, m):
        dists = np.empty(bs, dtype=np.float32)
        dists_i = np.empty(bs, dtype=np.int64)
        x1, y1, z1 = xyz[old]
        for tid in range(bs):  # every CUDA thread
            besti = 0
            best = f32(-1.0)
            for k in range(tid, n, bs):
                x2, y2, z2 = xyz[k]
                mag = f32(f32(x2 * x2) + f32(y2 * y2)) + f32(z2 * z2)
                if float(mag) <= 1e-3:  # double comparison
                    continue
                dx = f32(x2 - x1)
                dy = f32(y2 - y1)
                dz = f32(z2 - z1)
                d = f32(f32(dx * dx) + f32(dy * dy)) + f32(dz * dz)
                d2 = temp[k] if np.isnan(d) else (d if d < temp[k] else temp[k])  # fminf
                temp[k] = d2
                if d2 > best:
                    besti = k
                    best = d2
            dists[tid] = best
            dists_i[tid] = besti
        s = bs // 2
        while s >= 1:  # the `if (block_size >= 2s) { if (tid < s) __update(tid, tid+s) }` ladder
            for tid in range(s):
                v1, v2 = dists[tid], dists[tid + s]
                i1, i2 = dists_i[tid], dists_i[tid + s]
                dists[tid] = max(v1, v2)
                dists_i[tid] = i2 if v2 > v1 else i1
            s //= 2
        old = int(dists_i[0])
        idxs[j] = old
    return idxs


def ball_query_literal(new_xyz: np.ndarray, xyz: np.ndarray, radius: float, nsample: int) -> np.ndarray:
    """new_xyz (m,3), xyz (n,3) -> (m,nsample) int32, one scene."""
    new_xyz = np.ascontiguousarray(new_xyz, dtype=np.float32)
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    m, n = new_xyz.shape[0], xyz.shape[0]
    idx = np.zeros((m, nsample), dtype=np.int32)
    r = f32(radius)
    radius2 = f32(r * r)
    for j in range(m):
        nx, ny, nz = new_xyz[j]
        cnt = 0
        k = 0
        while k < n and cnt < nsample:
            x, y, z = xyz[k]
            dx = f32(nx - x)
            dy = f32(ny - y)
            dz = f32(nz - z)
            d2 = f32(f32(dx * dx) + f32(dy * dy)) + f32(dz * dz)
            if d2 < radius2:
                if cnt == 0:
                    idx[j, :] = k
                idx[j, cnt] = k
                cnt += 1
            k += 1
    return idx


def three_nn_literal(unknown: np.ndarray, known: np.ndarray):
    """unknown (n,3), known (m,3) -> dist2 (n,3) f32, idx (n,3) i32, one scene."""
    unknown = np.ascontiguousarray(unknown, dtype=np.float32)
    known = np.ascontiguousarray(known, dtype=np.float32)
    n, m = unknown.shape[0], known.shape[0]
    dist2 = np.zeros((n, 3), dtype=np.float32)
    idx = np.zeros((n, 3), dtype=np.int32)
    for j in range(n):
        ux, uy, uz = unknown[j]
        b1 = b2 = b3 = 1e40
        i1 = i2 = i3 = 0
        for k in range(m):
            x, y, z = known[k]
            dx = f32(ux - x)
            dy = f32(uy - y)
            dz = f32(uz - z)
            d = float(f32(f32(dx * dx) + f32(dy * dy)) + f32(dz * dz))
            if d < b1:
                b3, i3 = b2, i2
                b2, i2 = b1, i1
                b1, i1 = d, k
            elif d < b2:
                b3, i3 = b2, i2
                b2, i2 = d, k
            elif d < b3:
                b3, i3 = d, k
        with np.errstate(over="ignore"):
            dist2[j] = np.array([b1, b2, b3], dtype=np.float64).astype(np.float32)
        idx[j] = (i1, i2, i3)
    return dist2, idx
