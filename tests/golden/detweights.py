"""Platform-independent deterministic weights for golden fixtures.

``fill_(module, seed)`` overwrites every parameter and buffer of ``module`` (iterating ``state_dict()`` in key
order) with values derived from a 64-bit integer hash of (seed, key, element index) -- pure integer arithmetic
in numpy, converted to float32 exactly -- so the reference model in the build container and this repo's model
on any machine get bit-identical weights without storing them.
"""
import zlib

import numpy as np
import torch

_M = np.uint64(6364136223846793005)
_A = np.uint64(1442695040888963407)


def _uniform(n, key, seed):
    """n float32 values in [-0.5, 0.5), exact multiples of 2^-24."""
    k = np.uint64(zlib.crc32(key.encode()) + 1000003 * seed)
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + k) * _M + _A
        x ^= x >> np.uint64(29)
        x = x * _M + _A
        x ^= x >> np.uint64(32)
    bits = (x >> np.uint64(40)).astype(np.int64)  # 24 bits
    return (bits.astype(np.float32) / np.float32(1 << 24)) - np.float32(0.5)


@torch.no_grad()
def fill_(module, seed=0):
    sd = module.state_dict()
    for key in sorted(sd.keys()):
        t = sd[key]
        if key.endswith("num_batches_tracked"):
            t.zero_()
            continue
        if key.endswith(".pe"):  # sinusoid table: keep
            continue
        u = torch.from_numpy(_uniform(t.numel(), key, seed)).view(t.shape)
        if key.endswith("running_var"):
            v = 1.0 + u                       # [0.5, 1.5)
        elif key.endswith("running_mean"):
            v = 0.2 * u
        elif key.endswith("bn.weight") or ".bn" in key and key.endswith("weight") or key.endswith("a_2") \
                or (t.dim() == 1 and key.endswith("weight")):
            v = 1.0 + 0.2 * u                 # norm scales
        elif t.dim() == 1:
            v = 0.1 * u                       # biases
        else:
            fan_in = t[0].numel() if t.dim() > 1 else t.numel()
            if key.endswith("lut.weight"):
                v = 0.5 * u
            else:
                v = u * float(2.0 * np.sqrt(3.0 / fan_in))  # U(-sqrt(3/fan_in), +sqrt(3/fan_in))
        t.copy_(v.to(t.dtype))
    return module
