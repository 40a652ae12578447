"""Adam over one flat parameter buffer (csrc/elementwise.hip: adam_flat_kernel).

Same update rule as ``torch.optim.Adam(params, lr, weight_decay=wd)`` that the reference builds
(scripts/train.py:262: lr 1e-3, weight_decay 1e-5; L2 decay added to the gradient, bias-corrected moments).  The
parameters that receive gradients are re-pointed at views of ONE flat fp32 buffer (their values are preserved), the
moments are flat buffers of the same size and the step count lives on the device, so an optimizer step is one
gradient pack (``FlatGradBucket.pack``: a multi-tensor copy) + one kernel launch, capturable in the step's hipGraph --
instead of the 8 launches PyTorch's fused multi-tensor Adam needs for the model's ~300 tensors.  The flat gradient
buffer is the one the all-reduce works on, so multi-rank runs pay nothing extra.
"""
import torch

from ._native import check, lib


class FlatAdam:
    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.bucket = bucket
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        flat_g = bucket.flat
        if not flat_g.is_cuda:
            raise RuntimeError("CPU not supported")
        self.flat_p = torch.zeros_like(flat_g)
        with torch.no_grad():
            for p, off in zip(bucket.params, bucket.offsets):   # same (16-byte aligned) layout as the gradient bucket
                view = self.flat_p[off:off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view          # the module's parameter now lives in the flat buffer
        self.m = torch.zeros_like(flat_g)
        self.v = torch.zeros_like(flat_g)
        self.step_t = torch.zeros((), dtype=torch.float32, device=flat_g.device)

    @torch.no_grad()
    def step(self, grad_scale=1.0, skip_word=None):
        """One update from the gradients currently in ``bucket.flat`` (pack them first).  ``skip_word``: a device int64
        tensor; the kernel leaves parameters and moments untouched when it is non-zero (engine.Trainer: the sticky error
        word of a timed-out stream wait).  A non-zero word is TERMINAL: it is sticky and the Trainer raises at its next
        step(), so the step counter below (which a skipped update still advances) never feeds another update."""
        dev = self.flat_p.device
        self.step_t += 1
        with torch.cuda.device(dev):
            check(lib.spacap_adam_flat_f32(self.flat_p.data_ptr(), self.bucket.flat.data_ptr(), self.m.data_ptr(),
                                           self.v.data_ptr(), self.flat_p.numel(), self.lr, self.betas[0], self.betas[1],
                                           self.eps, self.weight_decay, self.step_t.data_ptr(), float(grad_scale),
                                           skip_word.data_ptr() if skip_word is not None else None,
                                           torch.cuda.current_stream(dev).cuda_stream), "spacap_adam_flat_f32")
