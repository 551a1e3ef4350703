"""Checkpoint wire format: the on-disk files of the reference load and save unchanged.

* published tokenizer ``vae_ch160v4096z32.pth`` and transformers ``var_d*.pth`` / ``d*.pth`` are plain ``torch.save``d
  state dicts (README.md:23-27,126-133); DDP runs prefix every key with ``module.`` (train_control_var_hpu.py:478);
* a plain-VAR checkpoint becomes a ControlVAR initialisation by the surgery of ``load_var_weight``
  (train_control_var_hpu.py:472-534): drop ``lvl_1L / pos_start / attn_bias_for_masking`` (rebuilt by the constructor for
  the doubled sequence), double ``pos_1LC`` (680 -> 1360 rows), ``load_state_dict(strict=False)``;
* training snapshots are ``{model_state_dict, optimizer_state_dict, epoch, step}`` (:420-428) and ``resume`` (:430-447)
  bumps the epoch unless the file name carries 'latest'.

Everything here is host code over tensors' storage; no kernel is involved.  The optimizer section is laid out exactly as
``torch.optim.AdamW.state_dict()`` (see FusedAdamW.state_dict) so snapshots move between the reference and this package
in both directions.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Any, Dict, Mapping, Optional, Sequence, Tuple, Union

import torch

DROPPED_FOR_CONTROL = ('lvl_1L', 'pos_start', 'attn_bias_for_masking')      # train_control_var_hpu.py:484-485

StateLike = Union[str, os.PathLike, Mapping[str, Any]]


def read_state(src: StateLike) -> "OrderedDict[str, torch.Tensor]":
    """path or already-loaded object -> flat state dict: unwrap ``model_state_dict`` (:474-475) and strip the DDP
    ``module.`` prefix (:476-479; the reference replaces every occurrence, so do we)."""
    obj = torch.load(os.fspath(src), map_location='cpu') if isinstance(src, (str, os.PathLike)) else src
    if 'model_state_dict' in obj:
        obj = obj['model_state_dict']
    return OrderedDict((k.replace('module.', ''), v) for k, v in obj.items())


def var_to_control_var_state(var_state: Mapping[str, torch.Tensor], patch_nums: Sequence[int], interpos: bool = False,
                             separator: bool = False, mpos: bool = False, vocab_size: int = 4096) -> "OrderedDict[str, torch.Tensor]":
    """The 'interleave_append' surgery of load_var_weight (:482-534).

    separator=True (:504-517,523-533; only reached with interpos=False): per scale the pn^2 position rows are laid down for the control
    half, a freshly drawn row (trunc_normal, std sqrt(1/(3C)), torch's global generator as upstream) for each separator, and the
    image half is filled with ``pos * -1 if mpos else 1`` - operator precedence makes that the CONSTANT 1 unless mpos is set (:514);
    the head grows to V + 18 rows: pretrained rows first, the rest trunc_normal * 0.02 (weight) and 0 (bias).

    interpos=False (the default of the training script, :103): pos_1LC' = cat(pos_1LC, pos_1LC) along L (:521).
    interpos=True: per scale, the pn² rows are laid down twice back to back - the [mask | image] order of the
    interleaved sequence (:489-503; the trunc-normal draw there is overwritten completely, so the result is deterministic).
    """
    sd = OrderedDict(var_state)
    for key in DROPPED_FOR_CONTROL:
        del sd[key]                                  # KeyError on a file that is not a VAR checkpoint, as in the reference
    pos = sd['pos_1LC']
    if interpos:
        parts, at = [], 0
        for pn in patch_nums:
            rows = pos[:, at:at + pn * pn]
            parts += [rows, rows]
            at += pn * pn
        sd['pos_1LC'] = torch.cat(parts, dim=1)
    elif separator:
        import math
        C = pos.shape[-1]
        init_std = math.sqrt(1 / C / 3)
        parts, at = [], 0
        for i, pn in enumerate(patch_nums):
            sp = 1 if i != 0 else 0
            pe = torch.empty((pn * pn + sp) * 2, C)
            torch.nn.init.trunc_normal_(pe, mean=0, std=init_std)
            pe[:pn * pn] = pos[0, at:at + pn * pn]
            pe[pn * pn + sp:pn * pn * 2 + sp] = pos[0, at:at + pn * pn] * -1 if mpos else 1
            parts.append(pe)
            at += pn * pn
        sd['pos_1LC'] = torch.cat(parts, dim=0).unsqueeze(0)
        n_sp = (len(patch_nums) - 1) * 2
        weight, bias = torch.empty(vocab_size + n_sp, C), torch.empty(vocab_size + n_sp)
        torch.nn.init.trunc_normal_(weight, mean=0, std=init_std)
        torch.nn.init.trunc_normal_(bias, mean=0, std=init_std)
        weight.mul_(0.02); bias.mul_(0.0)
        weight[:vocab_size] = sd['head.weight']
        bias[:vocab_size] = sd['head.bias']
        sd['head.weight'], sd['head.bias'] = weight, bias
    else:
        sd['pos_1LC'] = torch.cat([pos, pos], dim=1)
    return sd


def load_var_weight(var, src: StateLike, interpos: bool = False, mpos: bool = False):
    """Initialise a ControlVAR from a pretrained VAR file (train_control_var_hpu.py:472-534).  Returns the
    (missing_keys, unexpected_keys) record of the non-strict load: for a published ``var_d*.pth`` the missing keys are
    exactly the three rebuilt tensors plus ``cond_embed.weight``."""
    sd = read_state(src)
    if getattr(var, 'mask_factor', 1) > 1:
        sd = var_to_control_var_state(sd, var.patch_nums, interpos=interpos, separator=bool(getattr(var, 'separator', False)), mpos=mpos,
                                      vocab_size=var.V)
    return var.load_state_dict(sd, strict=False)


def load_weights(model, src: StateLike, strict: bool = True):
    """``model.load_state_dict(torch.load(path), strict=True)`` for a tokenizer or transformer file, accepting the
    wrapped / ``module.``-prefixed forms as well (demo usage in README.md; vqvae.py:106-109 patch lives in VQVAE)."""
    return model.load_state_dict(read_state(src), strict=strict)


def save_checkpoint(model, optimizer, epoch: int, step: int, save_dir: str = '', latest: bool = False) -> str:
    """train_control_var_hpu.py:420-428; returns the path written."""
    ckpt = {'model_state_dict': model.state_dict(), 'optimizer_state_dict': optimizer.state_dict(), 'epoch': epoch, 'step': step}
    path = os.path.join(save_dir, f"checkpoint_step_{'latest' if latest else step}.pth")
    torch.save(ckpt, path)
    return path


def resume(var, optimizer, path: StateLike) -> Tuple[int, int]:
    """train_control_var_hpu.py:430-447 -> (completed_steps, starting_epoch).  The model section is loaded strictly and
    as stored (no prefix stripping - the reference does none here), the optimizer section if present."""
    state = torch.load(os.fspath(path), map_location='cpu') if isinstance(path, (str, os.PathLike)) else path
    if 'model_state_dict' in state:
        var.load_state_dict(state['model_state_dict'], strict=True)
    if 'optimizer_state_dict' in state and optimizer is not None:
        optimizer.load_state_dict(state['optimizer_state_dict'])
    completed_steps, starting_epoch = state['step'], state['epoch']
    if not (isinstance(path, (str, os.PathLike)) and 'latest' in os.fspath(path)):
        starting_epoch += 1
    return completed_steps, starting_epoch
