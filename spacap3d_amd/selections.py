"""Test-only inspection point of the discrete selections of a training forward.

Every ReLU gate and every max-pool arg-max of the fused operators is re-derived in the backward from tensors the forward
saved (the pre-activations ``z`` with their BatchNorm rows, the arg-max map).  ``HOOK``, when set, is called by those
operators inside their forward, right BEFORE the tensors are saved for the backward, and may edit them in place:
tests/test_golden.py uses it to force the selections the reference's own run made (tests/golden/
train_step_cfg1_selections.npz) so that gradients can be held against the reference's at 1e-3 instead of at the level of a
flipped near-tie.  Never set outside tests; ``None`` costs one attribute read per operator call.

    HOOK(kind, gammas, zs, stats, **extra)
      kind "sa"       fused SA module (sa_mlp._SAMLP): gammas / zs / stats are 3-lists (zs[0] is None when the first layer's
                      pre-activation is not stored); stats rows are (mean, 1/std, gamma/std, beta): bn(z) = (z - row0) * row2 + row3;
                      z is point-major (B*N*S, C).  extra: arg (B, N, C3) uint8, out (B, N, C3), zmax (B, N, C3) or None, dims (B, N, S)
      kind "bn_relu"  fused_bn.BNReLU: 1-lists; z is (B, C, L) channel-major, stats rows are (mean, 1/std): bn(z) = (z - row0) * row1
                      * gamma + beta.  extra: beta
"""
HOOK = None


def visit(kind, **kw):
    if HOOK is not None:
        HOOK(kind, **kw)
