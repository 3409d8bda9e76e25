"""model/loss.py:52-57 `info_nce_loss` -- the loss of the PGAT+LBM training path (config "loss": "info_nce_loss"; trainer.py:52-56
regroups the scores to [queries, 1 + negatives] and passes all-zero targets) -- as one HIP launch that also leaves the gradient.
The reference's other losses are plain torch one-liners and stay with torch."""
import torch

from . import _lib


class _InfoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target):
        x = output if (output.dtype == torch.float32 and output.stride(-1) == 1) else output.float().contiguous()
        B, C = x.shape
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        d_x = torch.empty((B, C), dtype=torch.float32, device=x.device)
        tgt = None
        if target is not None:
            tgt = target if (target.dtype == torch.int64 and target.is_contiguous()) else target.to(torch.int64).contiguous()
        _lib.call("txe_info_nce", x.data_ptr(), x.stride(0) if B > 1 else C, B, C, None if tgt is None else tgt.data_ptr(),
                  loss.data_ptr(), d_x.data_ptr(), C, _lib.stream_ptr())
        ctx.save_for_backward(d_x)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (d_x,) = ctx.saved_tensors
        unit = _UNIT.get(grad_loss.device)
        if unit is not None and grad_loss.data_ptr() == unit.data_ptr() and grad_loss.numel() == 1:
            return d_x, None                    # d_x * 1: the plain `loss.backward()` of trainer.py:60 (LossTensor.backward below)
        return d_x * grad_loss, None


_UNIT = {}      # device -> the constant 1.0 that `loss.backward()` starts from (never written after its creation)


class LossTensor(torch.Tensor):
    """The scalar loss.  `loss.backward()` without a gradient (trainer/trainer.py:60) makes autograd allocate a ones tensor and fill it
    (one ~5 us launch), and the loss function's backward then multiplies its saved gradient by that 1.0 (another): here the call starts
    from a cached device constant, which _InfoNCE.backward recognises by its address and answers with the saved gradient itself -- the
    same numbers, two launches fewer per step.  Any other use (an explicit gradient, arithmetic on the loss first,
    torch.autograd.backward / grad) takes the ordinary path."""

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        if gradient is None and self.numel() == 1 and self.is_cuda and not create_graph:
            gradient = _UNIT.get(self.device)
            if gradient is None:
                gradient = _UNIT[self.device] = torch.ones((), dtype=self.dtype, device=self.device)
            if gradient.dtype != self.dtype:
                gradient = None
        return super().backward(gradient, retain_graph, create_graph, inputs)


def info_nce_loss(output, target=None):
    """output: (batch_size, 1 + negative_size) scores; target: (batch_size,) long (all zeros in trainer.py:53; None means that).
    Returns sum-reduced cross entropy, like the reference."""
    if output.dim() != 2:
        raise ValueError("info_nce_loss expects a [batch, 1 + negatives] tensor")
    if not output.is_cuda:
        raise RuntimeError("taxoexpan_amd.loss.info_nce_loss runs on the MI355X only (no CPU path)")
    return _InfoNCE.apply(output, target).as_subclass(LossTensor)
