"""Operator backend selection.

The product has exactly one backend: the HIP library (``spacap3d_amd.ext`` + ``spacap3d_amd.attention``).
``set_backend`` exists so that *tests and the CPU-baseline leg of bench.py* can drive the same host code
with the CPU oracle (``oracle/``) as the checker; nothing inside this package ever selects it, and there is
no automatic fallback: if the HIP library is missing, ``ops()`` raises ImportError.
"""
import contextlib

_current = None


class HipBackend:
    """The nine ``_ext`` entry points + fused attention, all on libspacap_hip.so."""

    name = "hip"

    def __init__(self):
        from . import ext  # raises ImportError when the extension is not built
        from . import attention as _att
        self._ext = ext
        for n in ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
                  "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
                  "group_points_grad", "group_max", "group_max_grad", "three_interpolate_grad_pm"):
            setattr(self, n, getattr(ext, n))
        self.attention = _att.attention
        self.self_attention_packed = _att.self_attention_packed
        self.layer_norm = _att.layer_norm
        self.layer_norm_residual = _att.layer_norm_residual
        self.relation_feature = _att.relation_feature
        self.relation_layer1 = _att.relation_layer1
        from . import fused_bn as _fbn
        self.bn_relu_train = _fbn.bn_relu_train
        self.bn_relu_eval = _fbn.bn_relu_eval
        from . import fused_losses as _fl
        self.detection_losses = _fl.detection_losses
        self.relation_losses = _fl.relation_losses
        self.caption_head_loss = _fl.caption_head_loss
        self.loss_tail = _fl.loss_tail
        self.proposal_decode = _fl.proposal_decode
        self.l2norm_rows = _fl.l2norm_rows
        self.vote_assemble = _fl.vote_assemble
        from . import fused_dropout as _fd
        self.relu_dropout = _fd.relu_dropout
        self.dropout_add = _fd.dropout_add
        from . import linear as _lin
        self.linear = _lin.linear
        self.ffn_tail = _lin.ffn_tail
        self.conv1x1 = _lin.conv1x1
        self.relation_tail = _lin.relation_tail
        self.relation_head = _lin.relation_head
        self.vocab_projection = _lin.vocab_projection
        from . import caption_prep as _cp
        self.caption_prep = _cp.caption_prep
        from . import tf_layer as _tf
        self.tf_stack = _tf
        from . import sa_mlp as _sa
        self.sa_mlp_train = _sa.sa_mlp_train
        self.sa_mlp_eval = _sa.sa_mlp_eval


def ops():
    global _current
    if _current is None:
        _current = HipBackend()
    return _current


def set_backend(b):
    global _current
    prev = _current
    _current = b
    return prev


@contextlib.contextmanager
def use_backend(b):
    prev = set_backend(b)
    try:
        yield b
    finally:
        set_backend(prev)
