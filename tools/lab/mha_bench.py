"""Packed self-attention forward / backward at the encoder shape (B 8, L 256, h 8, d_k 16), 20 iterations: run under
rocprofv3 --kernel-trace --stats to read the per-kernel averages."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import spacap3d_amd  # noqa
from spacap3d_amd.attention import self_attention_packed
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L, h, dk = 8, 256, 8, 16
qkv = torch.randn(B, L, 3 * h * dk, device=dev, requires_grad=True)
mask = (torch.rand(B, 1, L, device=dev) > 0.2)
w = torch.randn(B, L, h * dk, device=dev)
for need_p in (False, True):
    for it in range(20):
        out, p = self_attention_packed(qkv, h, mask=mask, dropout_p=0.1, training=True, need_p=need_p)
        (out * w).sum().backward()
torch.cuda.synchronize()
print("done")
