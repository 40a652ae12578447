"""Point-major results behind the reference's channel-major interfaces.

The reference's modules exchange features as (B, C, N) tensors (lib/pointnet2/pointnet2_modules.py:212-276: SA modules return
``new_features (B, C, npoint)``); the fused kernels of this package produce and consume (B, N, C) -- rows = points, channels
fastest.  ``ChannelMajorOf`` is the TYPE of "a (B, C, N) tensor that is the transposed view of a dense point-major tensor":
a consumer that knows the layout asks ``point_major_of(x)`` and gets the (B, N, C) tensor back without a copy; everyone else
uses ``x`` as the ordinary channel-major tensor it also is.  Any torch operation on it returns a PLAIN tensor (the claim
"I am a transposed view of ..." holds for this object only), so the layout can never be carried along by accident: it is
either present as a type or absent.
"""
import torch


class ChannelMajorOf(torch.Tensor):
    @staticmethod
    def wrap(pm: torch.Tensor) -> "ChannelMajorOf":
        """``pm`` (B, N, C), dense or a column window of wider rows (channels contiguous) -> the (B, C, N) transposed view of it,
        typed."""
        res = pm.transpose(1, 2).as_subclass(ChannelMajorOf)
        res._pm = pm
        return res

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **(kwargs or {}))
        return out


def point_major_of(x):
    """The dense (B, N, C) tensor ``x`` is a transposed view of, or None when ``x`` is not a ``ChannelMajorOf``."""
    return x._pm if isinstance(x, ChannelMajorOf) and getattr(x, "_pm", None) is not None else None
