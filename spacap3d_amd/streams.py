"""Independent branches of one training step on side HIP streams.

After the proposal module the step has three branches that only meet again in the scalar loss:
  (a) decoder + caption loss: ~420 launches of microseconds each on 256 tokens (pure latency),
  (b) relation head + relation loss: ~15 heavy kernels on 524 288 proposal pairs (HBM / MFMA bound)
      [available as a branch, off by default: see _only below],
  (c) vote / objectness / box / class losses: ~300 tiny launches on (B, 256) tensors.
On one stream they run back to back; here (b) and (c) are issued on side streams between a fork (side waits for
the main stream) and a join (main waits for side), so latency-bound chains overlap with heavy kernels.  The autograd
engine replays each branch's backward on the stream of its forward and orders the hand-offs itself; ``join_all`` after
``backward()`` re-joins every side stream (required inside a hipGraph capture, and before the optimizer reads the
gradients).  Disabled (the default): ``branch`` is a no-op and everything runs on the current stream.

Memory: tensors that cross streams are kept alive in the step's ``data_dict`` until the step ends, and the main stream
waits for the side streams at every join, so a block is never handed back to an allocator pool while another stream
can still touch it.
"""
import contextlib

import torch

import os

_enabled = False
# Measured on MI355X (cfg2, hipGraph replay): the detection-loss branch gained 0.16 ms / step while those losses were
# ~300 tiny launches and LOSES 0.3 ms now that they are four (fused_losses.py), so the engine leaves branches off by
# default (Trainer(multi_stream=True) / bench.py --streams turn them on); the relation-head
# branch LOSES 0.7 ms (its HBM-bound kernels fill every CU and the decoder's tiny kernels queue behind them), so
# it stays on the main stream unless asked for (SPACAP_BRANCHES=relation,detection_loss).
_only = set(os.environ.get("SPACAP_BRANCHES", "detection_loss").split(","))
_side = {}
_open = []


def enable(flag: bool = True):
    global _enabled
    _enabled = bool(flag)


def enabled() -> bool:
    return _enabled


@contextlib.contextmanager
def branch(name, like):
    """Run the body on the side stream ``name`` (forked from the current stream of ``like``'s device)."""
    if not (_enabled and torch.is_tensor(like) and like.is_cuda) or (_only is not None and name not in _only):
        yield False
        return
    dev = like.device
    side = _side.get((dev, name))
    if side is None:
        side = _side[(dev, name)] = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)
    side.wait_stream(cur)
    if (dev, name) not in _open:
        _open.append((dev, name))
    with torch.cuda.stream(side):
        yield True


def join(name, like):
    """The current stream waits for side stream ``name``."""
    if torch.is_tensor(like) and like.is_cuda and (like.device, name) in _side and (like.device, name) in _open:
        torch.cuda.current_stream(like.device).wait_stream(_side[(like.device, name)])


def join_all(device, close=True):
    """The current stream waits for every side stream used since the last close (call after ``backward()``)."""
    for key in list(_open):
        if key[0] == device:
            torch.cuda.current_stream(device).wait_stream(_side[key])
            if close:
                _open.remove(key)
