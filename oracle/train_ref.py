"""ORACLE (test infrastructure only - never imported by the product path).

Training-step restatement (train_control_var_hpu.py:157-250): teacher-forced forward (oracle.var_ref), token
cross-entropy, gradients by torch autograd over the functional oracle, gradient-norm clipping and the AdamW update
(torch.optim.AdamW semantics restated explicitly).  Pinned by tests/golden/train_step_d2.npz.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from controlvar_amd.spec import VarConfig, var_state_shapes
from . import var_ref

SD = Dict[str, torch.Tensor]


def trainable_keys(cfg: VarConfig) -> List[str]:
    return [k for k, (_, kind) in var_state_shapes(cfg).items() if kind == 'param']


def loss_and_grads(sd: SD, cfg: VarConfig, cls: torch.Tensor, x_wo_first: torch.Tensor, cond_type: Optional[torch.Tensor],
                   targets: torch.Tensor, ignore_mask: Optional[torch.Tensor] = None, prec=var_ref.FP32, mask_first: bool = True):
    """-> (loss, per-token loss (B*L,), {key: grad}) with the reduction of train_control_var_hpu.py:228-239."""
    leaf = {k: (v.detach().clone().requires_grad_(True) if k in set(trainable_keys(cfg)) else v) for k, v in sd.items()}
    logits = var_ref.forward_logits(leaf, cfg, cls, x_wo_first, cond_type, prec, mask_first)
    loss_tok = F.cross_entropy(logits.view(-1, logits.size(-1)), targets.view(-1), reduction='none')
    if ignore_mask is not None:
        m = ignore_mask.view(-1).float()
        loss = (loss_tok * m).mean() / (m.mean() + 1e-6)
    else:
        loss = loss_tok.mean()
    loss.backward()
    grads = {k: leaf[k].grad for k in trainable_keys(cfg)}
    return loss.detach(), loss_tok.detach(), grads


def clip_coef(grads: Dict[str, torch.Tensor], max_norm: float) -> Tuple[float, float]:
    """torch.nn.utils.clip_grad_norm_: total L2 norm and the factor min(1, max_norm / (norm + 1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
    return total, min(1.0, max_norm / (total + 1e-6))


def adamw_update(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float, wd: float,
                 b1: float = 0.9, b2: float = 0.95, eps: float = 1e-8):
    """one torch.optim.AdamW step (decoupled decay first, bias-corrected moments); returns (p, m, v)"""
    p = p * (1.0 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v
