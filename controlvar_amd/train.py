"""Host-side training logic of the hot path.

* lr_wd_annealing / filter_params: restated from utils/lr_control.py:10-101 (per-step LR / weight-decay schedule and
  the decay / no-decay parameter split of train_control_var_hpu.py:609-615);
* Trainer: the step of train_control_var_hpu.py:157-250 over the HIP kernels (tokenise -> interleave -> teacher-forced
  forward -> fused cross-entropy -> hand-written backward -> gradient all-reduce -> clip -> fused AdamW).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

NOWD_KEYS = ('cls_token', 'start_token', 'task_token', 'cfg_uncond', 'pos_embed', 'pos_1LC', 'pos_start', 'start_pos', 'lvl_embed',
             'gamma', 'beta', 'ada_gss', 'moe_bias', 'scale_mul')          # train_control_var_hpu.py:609-615


def lr_wd_factors(sche_type: str, peak_lr: float, wd: float, wd_end: float, cur_it: int, wp_it: float, max_it: int,
                  wp0: float = 0.005, wpe: float = 0.001) -> Tuple[float, float]:
    """(lr, weight_decay) of iteration cur_it  (utils/lr_control.py:10-48)."""
    wp_it = round(wp_it)
    if cur_it < wp_it:
        cur = wp0 + (1 - wp0) * cur_it / wp_it
    else:
        pasd = (cur_it - wp_it) / (max_it - 1 - wp_it)
        rest = 1 - pasd
        if sche_type == 'cos':
            cur = wpe + (1 - wpe) * (0.5 + 0.5 * math.cos(math.pi * pasd))
        elif sche_type == 'lin':
            T = 0.15
            cur = 1 if pasd < T else wpe + (1 - wpe) * rest / (1 - T)
        elif sche_type == 'lin0':
            T = 0.05
            cur = 1 if pasd < T else wpe + (1 - wpe) * rest / (1 - T)
        elif sche_type == 'lin00':
            cur = wpe + (1 - wpe) * rest
        elif sche_type.startswith('lin'):
            T = float(sche_type[3:])
            max_rest = 1 - T
            wpe_mid = (1 + (wpe + (1 - wpe) * max_rest)) / 2
            cur = 1 + (wpe_mid - 1) * pasd / T if pasd < T else wpe + (wpe_mid - wpe) * rest / max_rest
        elif sche_type == 'exp':
            T = 0.15
            cur = 1 if pasd < T else math.exp((pasd - T) / (1 - T) * math.log(wpe))
        else:
            raise NotImplementedError(f'unknown sche_type {sche_type}')
    lr = cur * peak_lr
    pasd = cur_it / (max_it - 1)
    cur_wd = wd_end + (wd - wd_end) * (0.5 + 0.5 * math.cos(math.pi * pasd))
    return lr, cur_wd


def lr_wd_annealing(sche_type: str, optimizer, peak_lr, wd, wd_end, cur_it, wp_it, max_it, wp0=0.005, wpe=0.001):
    """Drop-in of utils/lr_control.py:10-64 for anything with torch-style ``param_groups`` (returns min/max lr, min/max wd)."""
    lr, cur_wd = lr_wd_factors(sche_type, peak_lr, wd, wd_end, cur_it, wp_it, max_it, wp0, wpe)
    inf = 1e6
    min_lr, max_lr, min_wd, max_wd = inf, -1, inf, -1
    for g in optimizer.param_groups:
        g['lr'] = lr * g.get('lr_sc', 1)
        max_lr, min_lr = max(max_lr, g['lr']), min(min_lr, g['lr'])
        g['weight_decay'] = cur_wd * g.get('wd_sc', 1)
        max_wd = max(max_wd, g['weight_decay'])
        if g['weight_decay'] > 0:
            min_wd = min(min_wd, g['weight_decay'])
    if min_lr == inf:
        min_lr = -1
    if min_wd == inf:
        min_wd = -1
    return min_lr, max_lr, min_wd, max_wd


def decays(name: str, ndim: int, nowd_keys: Iterable[str] = NOWD_KEYS) -> bool:
    """True if the parameter is weight-decayed (utils/lr_control.py:84-87)."""
    return not (ndim == 1 or name.endswith('bias') or any(k in name for k in nowd_keys))


def filter_params(model, nowd_keys: Iterable[str] = NOWD_KEYS):
    """names, params, [group 'D' (wd_sc 1), group 'ND' (wd_sc 0)] in first-seen order (utils/lr_control.py:67-101)."""
    groups: Dict[str, dict] = {}
    names, paras = [], []
    for name, p in model.named_parameters():
        name = name.replace('_fsdp_wrapped_module.', '')
        if not p.requires_grad:
            raise AssertionError(f'frozen parameter {name}')
        names.append(name)
        paras.append(p)
        gname = 'D' if decays(name, p.ndim, nowd_keys) else 'ND'
        groups.setdefault(gname, {'params': [], 'wd_sc': 1.0 if gname == 'D' else 0.0, 'lr_sc': 1.0})['params'].append(p)
    return names, paras, list(groups.values())
