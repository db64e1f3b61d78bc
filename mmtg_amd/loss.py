"""Drop-in for the reference's ``MyLoss`` (src/loss.py:39-74): the rating-conditioned
sequence likelihood / unlikelihood objective, computed by the HIP loss kernels
(row log-sum-exp + per-sample reduction) with an analytic backward."""
from __future__ import annotations

import torch

from . import hip


def _as_rows(outputs):
    """logits [B,T,V] -> (tensor, ld) usable by the kernels without a copy when possible."""
    B, T, V = outputs.shape
    ok = (outputs.dtype == torch.float32 and outputs.stride(2) == 1 and outputs.stride(1) % 4 == 0
          and outputs.stride(0) == T * outputs.stride(1) and outputs.data_ptr() % 16 == 0)
    if ok:
        return outputs, outputs.stride(1)
    ld = (V + 3) // 4 * 4
    buf = torch.zeros(B, T, ld, device=outputs.device, dtype=torch.float32)
    buf[:, :, :V] = outputs
    return buf, ld


class _MyLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, targets, ratings, stage, P):
        B, T, V = outputs.shape
        L = targets.shape[1]
        if T != P + L:
            raise ValueError("outputs has %d positions, expected topic_prompt_length + len(targets) = %d" % (T, P + L))
        dev = outputs.device
        rows, ld = _as_rows(outputs.detach())
        targets = targets.to(dev, torch.long).contiguous()
        dummy_topic = torch.zeros(B, max(P, 1), device=dev, dtype=torch.long)
        M = B * T
        nll = torch.empty(M, device=dev)
        lse = torch.empty(M, device=dev)
        ce = torch.empty(B, device=dev)
        coef = torch.empty(B, device=dev)
        sc = torch.empty(2, device=dev)
        hip.loss_fwd(rows, ld, V, dummy_topic, targets, ratings.to(dev, torch.long).contiguous(), int(stage), False,
                     B, P, L, float(B), nll, lse, ce, coef, sc)
        ctx.save_for_backward(rows, targets, dummy_topic, lse, coef)
        ctx.dims = (B, P, L, V, ld)
        return sc[0].clone()

    @staticmethod
    def backward(ctx, g):
        rows, targets, dummy_topic, lse, coef = ctx.saved_tensors
        B, P, L, V, ld = ctx.dims
        d = torch.empty(B, P + L, V, device=rows.device, dtype=torch.float32)
        hip.loss_bwd(rows, ld, V, dummy_topic, targets, lse, coef, float(g), B, P, L, d, V, V)
        return d, None, None, None, None


class MyLoss(torch.nn.Module):
    def __init__(self, data_config, model_cfgs):
        super().__init__()
        self._max_topic_len = data_config.topic_prompt_length
        self._seq_len = model_cfgs["seq_len"]

    def forward(self, outputs, targets, ratings, stage):
        """
        outputs: (batch, topic_prompt_length + max_seq_length + 1, vocab)
        targets: (batch, max_seq_length + 1);  ratings: (batch)
        """
        if not outputs.is_cuda:
            raise RuntimeError("MyLoss runs on the MI355X only (HIP kernels, no CPU fallback)")
        return _MyLossFn.apply(outputs, targets, ratings, stage, self._max_topic_len)
