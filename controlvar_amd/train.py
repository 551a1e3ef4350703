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


# =====================================================================================
# Training engine (GPU): forward with saved activations, hand-written backward, fused optimizer
# =====================================================================================
import torch  # noqa: E402

from . import ops  # noqa: E402
from ._lib import ACT_GELU_GRAD, ACT_GELU_TANH, ACT_NONE  # noqa: E402


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


class TrainEngine:
    """Teacher-forced forward + backward of ControlVAR / VAR over the HIP kernels (control_var.py:568-651 under autograd).

    Activations of every block are kept in HBM (288 GB: no recomputation).  Weight gradients are produced by the same
    MFMA GEMM as the forward on transposed operands (activations are transposed into zero-padded [N][Mpad] buffers),
    accumulated in fp32 into per-layer contiguous slabs so that each layer's slab can be all-reduced as soon as its
    backward is done (bucket = layer, overlapped with the rest of the backward on a side stream).
    """

    def __init__(self, var, drop_path: bool = True, reducer=None):
        self.var = var
        self.cfg = var.cfg
        self.drop_path = drop_path
        self.reducer = reducer
        self._B = None
        self._wt_gen = -1

    # ---------------------------------------------------------------- buffers
    def _setup(self, B: int):
        if self._B == B:
            return
        cfg, var = self.cfg, self.var
        dev, T = var.device, var.compute_dtype
        C, depth, V = cfg.C, cfg.depth, cfg.head_ld          # head columns incl. separator labels, padded to 8 (spec.VarConfig.head_ld)
        L = cfg.pyramid.L
        M = B * L
        hid = round(C * cfg.mlp_ratio)
        n_ada = depth * 6 * C + 2 * C
        self.M, self.Mp, self.Bp = M, _pad8(M), _pad8(B)
        f32 = dict(device=dev, dtype=torch.float32)
        tT = dict(device=dev, dtype=T)
        # saved activations
        self.Xs = torch.empty(depth + 1, M, C, **f32)
        self.X1s = torch.empty(depth, M, C, **f32)
        self.U = torch.empty(depth, M, C, **tT); self.O = torch.empty(depth, M, C, **tT)
        self.F1 = torch.empty(depth, M, C, **tT); self.U2 = torch.empty(depth, M, C, **tT); self.F2 = torch.empty(depth, M, C, **tT)
        self.A = torch.empty(depth, M, hid, **tT); self.Hh = torch.empty(depth, M, hid, **tT)
        self.arena = torch.empty(depth, B, L, 3 * C, **tT)
        self.LSE = torch.empty(depth, B, cfg.H, L, **f32)
        self.NORMS = torch.empty(depth, B, L, cfg.H, 2, **f32) if cfg.uses_cos_attn else None
        self.DSM = torch.empty(M, cfg.H, **f32) if cfg.uses_cos_attn else None
        self.UH = torch.empty(M, C, **tT)
        self.logits = torch.empty(M, V, **f32)
        self.loss_tok = torch.empty(M, **f32)
        # backward work buffers
        self.dlogits = torch.empty(M, V, **tT)
        self.dX = torch.empty(M, C, **f32)
        self.DF = torch.empty(M, C, **tT); self.DU = torch.empty(M, C, **tT)
        self.DH = torch.empty(M, hid, **tT)
        self.DQKV = torch.empty(B, L, 3 * C, **tT)
        nmax = max(hid, 3 * C, V)
        self.TA = torch.zeros(nmax * self.Mp, **tT)           # zero padding columns stay zero
        self.TB = torch.zeros(nmax * self.Mp, **tT)
        self.ncode = len(cfg.pyramid.code_positions()) - cfg.pyramid.first_l         # teacher-forcing tokens per sample (separators excluded)
        self.TA32 = torch.zeros(C * _pad8(B * self.ncode), **f32)
        self.TB32 = torch.zeros(cfg.cvae * _pad8(B * self.ncode), **f32)
        self.dada = torch.zeros(B, n_ada, **f32)
        self.ws = torch.empty(max(ops.train_ws_floats(M, B, C), 64 * max(hid, 3 * C, V, n_ada), L * C, B * cfg.H * L,
                                  6 * C * C if cfg.shared_aln else 0) + 16, **f32)
        # gradient slabs: [layer][w_qkv | w_proj | w_fc1 | w_fc2 | b_qkv | b_proj | b_fc1 | b_fc2]
        self.slab_off = {}
        o = 0
        for name, n in (('w_qkv', 3 * C * C), ('w_proj', C * C), ('w_fc1', hid * C), ('w_fc2', C * hid), ('b_qkv', 3 * C), ('b_proj', C),
                        ('b_fc1', hid), ('b_fc2', C), ('scale_mul', _pad8(cfg.H))):
            self.slab_off[name] = o
            o += n
        self.slab = o
        self.G_layers = torch.zeros(depth, self.slab, **f32)
        self.G_ada = torch.zeros(n_ada * C + n_ada, **f32)                       # [w_ada | b_ada]
        misc = [('w_head', V * C), ('b_head', V), ('w_we', C * cfg.cvae), ('b_we', C), ('pos', L * C), ('lvl', len(cfg.patch_nums) * C),
                ('pos_start', cfg.pyramid.first_l * C), ('class_emb', (cfg.num_classes + 1) * C), ('cond_embed', 5 * C)]
        if cfg.shared_aln:      # the shared generator's gradient = sum over blocks of the per-block (folded) ones
            misc += [('w_shared', 6 * C * C), ('b_shared', 6 * C)]
        if cfg.type_pos:
            misc += [('type', cfg.mask_factor * C)]
        if cfg.separator:
            misc += [('special', cfg.n_special * C)]
        self.misc_off = {}
        o = 0
        for name, n in misc:
            self.misc_off[name] = (o, n)
            o += n
        self.G_misc = torch.zeros(o, **f32)
        self.buckets = [self.G_layers[i] for i in range(depth)] + [self.G_ada, self.G_misc]
        self._B = B
        self._transposed_weights()

    def _transposed_weights(self):
        """W^T copies for the data-gradient GEMMs (refreshed after every optimizer step)."""
        P, cfg = self.var._pack(), self.cfg
        C, depth, V = cfg.C, cfg.depth, cfg.head_ld
        hid = P['w_fc1'].shape[1]
        T = self.var.compute_dtype
        dev = self.var.device
        mk = lambda *s: torch.empty(*s, device=dev, dtype=T)
        self.WT = dict(qkv=mk(depth, C, 3 * C), proj=mk(depth, C, C), fc1=mk(depth, C, hid), fc2=mk(depth, hid, C), head=mk(C, V), ada=mk(C, P['n_ada']))
        ops.transpose(P['w_qkv'], self.WT['qkv'], depth, 3 * C, C, C)
        ops.transpose(P['w_proj'], self.WT['proj'], depth, C, C, C)
        ops.transpose(P['w_fc1'], self.WT['fc1'], depth, hid, C, C)
        ops.transpose(P['w_fc2'], self.WT['fc2'], depth, C, hid, hid)
        ops.transpose(P['w_head'], self.WT['head'], 1, V, C, C)
        ops.transpose(P['w_ada'], self.WT['ada'], 1, P['n_ada'], C, C)
        self._wt_gen = self.var._pack_gen

    def grads(self) -> Dict[str, torch.Tensor]:
        """state_dict key -> gradient view (fp32)"""
        cfg = self.cfg
        C, depth, V, Vo = cfg.C, cfg.depth, cfg.head_ld, cfg.head_out
        hid = round(C * cfg.mlp_ratio)
        so = self.slab_off
        g: Dict[str, torch.Tensor] = {}
        for i in range(depth):
            s = self.G_layers[i]
            p = f'blocks.{i}.'
            g[p + 'attn.mat_qkv.weight'] = s[so['w_qkv']:so['w_qkv'] + 3 * C * C].view(3 * C, C)
            g[p + 'attn.proj.weight'] = s[so['w_proj']:so['w_proj'] + C * C].view(C, C)
            g[p + 'ffn.fc1.weight'] = s[so['w_fc1']:so['w_fc1'] + hid * C].view(hid, C)
            g[p + 'ffn.fc2.weight'] = s[so['w_fc2']:so['w_fc2'] + C * hid].view(C, hid)
            g[p + 'attn.q_bias'] = s[so['b_qkv']:so['b_qkv'] + C]
            g[p + 'attn.v_bias'] = s[so['b_qkv'] + 2 * C:so['b_qkv'] + 3 * C]
            g[p + 'attn.proj.bias'] = s[so['b_proj']:so['b_proj'] + C]
            g[p + 'ffn.fc1.bias'] = s[so['b_fc1']:so['b_fc1'] + hid]
            g[p + 'ffn.fc2.bias'] = s[so['b_fc2']:so['b_fc2'] + C]
            if cfg.uses_cos_attn:
                g[p + 'attn.scale_mul_1H11'] = s[so['scale_mul']:so['scale_mul'] + cfg.H].view(1, cfg.H, 1, 1)
            n_ada = depth * 6 * C + 2 * C
            if cfg.sa_block:        # constants in the folded bias (models._pack): [gamma1, gamma2, w1 - 1, w2 - 1, b1, b2]
                gb = self.G_ada[n_ada * C:][i * 6 * C:(i + 1) * 6 * C]
                if cfg.layer_scale >= 0:
                    g[p + 'gamma1'], g[p + 'gamma2'] = gb[0:C], gb[C:2 * C]
                g[p + 'norm1.weight'], g[p + 'norm2.weight'] = gb[2 * C:3 * C], gb[3 * C:4 * C]
                g[p + 'norm1.bias'], g[p + 'norm2.bias'] = gb[4 * C:5 * C], gb[5 * C:6 * C]
            elif cfg.shared_aln:    # folded bias_i = shared bias + ada_gss_i (models._pack): d ada_gss_i = d bias_i
                g[p + 'ada_gss'] = self.G_ada[n_ada * C:][i * 6 * C:(i + 1) * 6 * C].view(1, 1, 6, C)
            else:
                g[p + 'ada_lin.1.weight'] = self.G_ada[:n_ada * C].view(n_ada, C)[i * 6 * C:(i + 1) * 6 * C]
                g[p + 'ada_lin.1.bias'] = self.G_ada[n_ada * C:][i * 6 * C:(i + 1) * 6 * C]
        n_ada = depth * 6 * C + 2 * C
        mo = lambda k: self.G_misc[self.misc_off[k][0]:self.misc_off[k][0] + self.misc_off[k][1]]
        py = cfg.pyramid
        if cfg.sa_block:            # head = Sequential(LayerNorm, Linear) (control_var.py:205-207)
            hb = self.G_ada[n_ada * C:][depth * 6 * C:]
            g['head.0.weight'], g['head.0.bias'] = hb[:C], hb[C:]
            g['head.1.weight'] = mo('w_head').view(V, C)[:Vo]; g['head.1.bias'] = mo('b_head')[:Vo]
        else:
            g['head_nm.ada_lin.1.weight'] = self.G_ada[:n_ada * C].view(n_ada, C)[depth * 6 * C:]
            g['head_nm.ada_lin.1.bias'] = self.G_ada[n_ada * C:][depth * 6 * C:]
            g['head.weight'] = mo('w_head').view(V, C)[:Vo]; g['head.bias'] = mo('b_head')[:Vo]
        g['word_embed.weight'] = mo('w_we').view(C, cfg.cvae); g['word_embed.bias'] = mo('b_we')
        g['pos_1LC'] = mo('pos').view(1, py.L, C); g['lvl_embed.weight'] = mo('lvl').view(len(cfg.patch_nums), C)
        g['pos_start'] = mo('pos_start').view(1, py.first_l, C)
        g['class_emb.weight'] = mo('class_emb').view(cfg.num_classes + 1, C)
        if cfg.mask_factor == 2:
            g['cond_embed.weight'] = mo('cond_embed').view(5, C)
        if cfg.shared_aln:
            g['shared_ada_lin.1.weight'] = mo('w_shared').view(6 * C, C); g['shared_ada_lin.1.bias'] = mo('b_shared')
        if cfg.type_pos:
            g['type_embed.weight'] = mo('type').view(cfg.mask_factor, C)
        if cfg.separator:
            g['special_embed.weight'] = mo('special').view(cfg.n_special, C)
        return g

    # ---------------------------------------------------------------- forward + backward
    @torch.no_grad()
    def forward_backward(self, label_B: torch.Tensor, x_wo_first: torch.Tensor, cond_type: Optional[torch.Tensor], targets: torch.Tensor,
                         ignore_mask: Optional[torch.Tensor] = None, drop_seed: Optional[int] = None, mask_first: bool = True):
        """-> (loss scalar tensor, per-token loss (B*L,)); gradients land in self.grads() (overwritten, not accumulated)."""
        self.forward_train(label_B, x_wo_first, cond_type, drop_seed, mask_first)
        dev = self.var.device
        M, V = self.M, self.cfg.head_ld
        tg = targets.to(device=dev, dtype=torch.int32).contiguous().view(-1)
        if ignore_mask is not None:                                  # train_control_var_hpu.py:233-237
            w = ignore_mask.to(device=dev, dtype=torch.float32).contiguous().view(-1)
            gscale = 1.0 / (M * (float(w.mean()) + 1e-6))
        else:
            w, gscale = None, 1.0 / M
        ops.ce_fwd_bwd(self.logits, tg, w, gscale, self.loss_tok, self.dlogits, M, V)
        loss = (self.loss_tok * w).mean() / (w.mean() + 1e-6) if w is not None else self.loss_tok.mean()
        self.backward()
        return loss, self.loss_tok

    @torch.no_grad()
    def forward_train(self, label_B: torch.Tensor, x_wo_first: torch.Tensor, cond_type: Optional[torch.Tensor], drop_seed: Optional[int] = None,
                      mask_first: bool = True):
        """teacher-forced forward that keeps every block's activations; returns logits (B*L, V) fp32"""
        cfg, var = self.cfg, self.var
        P = var._pack(check=True)
        py, C, depth, V, H = cfg.pyramid, cfg.C, cfg.depth, cfg.head_ld, cfg.H
        L, fl = py.L, py.first_l
        dev, T = var.device, var.compute_dtype
        B = x_wo_first.shape[0]
        self._setup(B)
        if self._wt_gen != var._pack_gen:          # the W^T copies follow the packed weights (any optimizer, manual edits, resume)
            self._transposed_weights()
        M, Mp = self.M, self.Mp
        hid = P['w_fc1'].shape[1]
        n_ada = P['n_ada']
        eps = cfg.norm_eps
        from .spec import attention_levels
        lvl_end, holes = attention_levels(cfg)
        scale = float(cfg.attn_scale)
        from .models import _check_index_range
        _check_index_range(label_B, 0, cfg.num_classes, 'label_B')
        labels = label_B.to(dev)
        types = cond_type.to(dev) if (cond_type is not None and cfg.mask_factor == 2) else None
        if types is not None:
            _check_index_range(types, 0, 4, 'cond_type')
        if cfg.cond_drop_rate > 0:                                  # control_var.py:578,584: every forward(), train or eval
            labels = torch.where(torch.rand(B, device=dev) < cfg.cond_drop_rate, cfg.num_classes, labels)
            if types is not None:
                types = torch.where(torch.rand(B, device=dev) < cfg.cond_drop_rate, 4, types)
        labels = labels.to(torch.int32).contiguous()
        types = types.to(torch.int32).contiguous() if types is not None else None
        # DropPath row scales (helpers.py:39-46): per sample, per block, per branch
        dp1 = dp2 = None
        rate = cfg.drop_path_rate                                     # constructor argument; the factories pass 0.1 * depth / 24 (models/__init__.py:15,40)
        if self.drop_path and var.training and rate > 0:
            g = torch.Generator(device=dev)
            g.manual_seed(drop_seed if drop_seed is not None else int(torch.empty((), dtype=torch.int64).random_().item()))
            dpr = torch.linspace(0, rate, depth, device=dev).view(depth, 1)
            keep = 1 - dpr
            dp1 = (torch.rand(depth, B, device=dev, generator=g) < keep).float() / keep
            dp2 = (torch.rand(depth, B, device=dev, generator=g) < keep).float() / keep
        # ---- forward
        x0 = self.Xs[0]
        cond = torch.empty(B, C, device=dev, dtype=torch.float32)
        mask_first = self._mask_first = bool(mask_first) or cfg.mask_factor != 2
        table = P['lvl_pos_fwd'] if mask_first else P['lvl_pos_fwd_']
        var._first_tokens(P, labels, types, x0, cond, B, L, table, mask_first)
        tok = x_wo_first.to(device=dev, dtype=torch.float32).contiguous()
        if tok.shape[1] != self.ncode:
            raise AssertionError(f'teacher-forcing input has {tok.shape[1]} tokens, expected {self.ncode}')
        var._embed_teacher_forced(P, tok, x0, B, table, mask_first)
        cs = torch.empty(B, C, device=dev, dtype=T)
        ops.silu_cast(cond, cs)
        ada = torch.empty(B, n_ada, device=dev, dtype=torch.float32)
        ops.gemm(cs, P['w_ada'], ada, M=B, N=n_ada, K=C, bias=P['b_ada'])
        for i in range(depth):
            a0 = i * 6 * C
            x = self.Xs[i]
            ops.ln_modulate(x, ada, a0 + 2 * C, a0 + 4 * C, n_ada, L, self.U[i], M, C, eps)
            ops.gemm(self.U[i], P['w_qkv'], self.arena[i], M=M, N=3 * C, K=C, w_off=i * 3 * C * C, bias=P['b_qkv'][i])
            if cfg.uses_cos_attn:
                ops.cos_qk_norm(self.arena[i], B, H, L, 0, L, P['scale_mul'], sm_off=i * H, norms=self.NORMS[i])
            ops.attention(self.arena[i], self.O[i], B, H, L, 0, L, scale, lvl_end, lse=self.LSE[i], holes=holes)
            # x1 = x + (gamma1 * keep1) * proj(o): gate / DropPath scale / residual in the GEMM epilogue; the branch output the backward
            # needs (d gamma1 = sum dx * f) is stored next to it
            ops.gemm(self.O[i], P['w_proj'], self.X1s[i], M=M, N=C, K=C, w_off=i * C * C, bias=P['b_proj'][i], gate=ada, ldg=n_ada, gate_rows=L,
                     gate_off=a0, gate_scale=dp1[i].contiguous() if dp1 is not None else None, residual=x, pre_act=self.F1[i])
            ops.ln_modulate(self.X1s[i], ada, a0 + 3 * C, a0 + 5 * C, n_ada, L, self.U2[i], M, C, eps)
            # fc1 with the GELU in its epilogue; the pre-activation the backward needs is stored alongside (no separate gelu pass)
            ops.gemm(self.U2[i], P['w_fc1'], self.Hh[i], M=M, N=hid, K=C, w_off=i * hid * C, bias=P['b_fc1'][i], act=ACT_GELU_TANH,
                     pre_act=self.A[i])
            ops.gemm(self.Hh[i], P['w_fc2'], self.Xs[i + 1], M=M, N=C, K=hid, w_off=i * C * hid, bias=P['b_fc2'][i], gate=ada, ldg=n_ada, gate_rows=L,
                     gate_off=a0 + C, gate_scale=dp2[i].contiguous() if dp2 is not None else None, residual=self.X1s[i], pre_act=self.F2[i])
        ah = depth * 6 * C
        ops.ln_modulate(self.Xs[depth], ada, ah, ah + C, n_ada, L, self.UH, M, C, eps)
        ops.gemm(self.UH, P['w_head'], self.logits, M=M, N=V, K=C, bias=P['b_head'])
        self._saved = dict(ada=ada, cs=cs, cond=cond, labels=labels, types=types, tok=tok, dp1=dp1, dp2=dp2, B=B)
        return self.logits

    @torch.no_grad()
    def _wgrad(self, dY, n_out: int, X, k_in: int, G, w_off: int, b_off=None):
        """dW[n_out, k_in] = dY^T X into the fp32 gradient slab (+ the bias gradient = column sums of dY).  bf16 mode: `cvar_gemm_tn` reads both
        token-major operands in place (LDS transpose-read) - no transposed copies; shapes it does not take (a dimension that is not a
        multiple of its tile, the fp32 parity mode) go through two HBM transposes + `cvar_gemm` as in round 1."""
        M, Mp = self.M, self.Mp
        if (dY.dtype == torch.bfloat16 and X.dtype == torch.bfloat16 and n_out % 128 == 0 and k_in % 128 == 0
                and 2 * M * max(n_out, k_in) < 2 ** 31 - 1):
            # the bias gradient (column sums of dY) rides on the same pass: ones-operand MFMAs on the dY fragments (round 3; a separate colsum pass before)
            ops.gemm_tn(dY, X, G, T=M, Nn=n_out, Kk=k_in, c_off=w_off, colsum=G if b_off is not None else None, colsum_off=b_off or 0)
            return
        ops.transpose(dY, self.TA, 1, M, n_out, n_out, ld_out=Mp)
        ops.transpose(X, self.TB, 1, M, k_in, k_in, ld_out=Mp)
        ops.gemm(self.TA, self.TB, G, M=n_out, N=k_in, K=Mp, c_off=w_off)
        if b_off is not None:
            ops.rowsum(self.TA, Mp, G, n_out, M, out_off=b_off)

    def backward(self):
        """backward from self.dlogits (compute dtype, (B*L, V)) through head, blocks, adaLN generator and embeddings"""
        cfg, var = self.cfg, self.var
        P = var._pack()
        py, C, depth, V, H = cfg.pyramid, cfg.C, cfg.depth, cfg.head_ld, cfg.H
        L, fl = py.L, py.first_l
        dev, T = var.device, var.compute_dtype
        sv = self._saved
        ada, cs, cond, labels, types, tok, dp1, dp2, B = (sv[k] for k in ('ada', 'cs', 'cond', 'labels', 'types', 'tok', 'dp1', 'dp2', 'B'))
        M, Mp = self.M, self.Mp
        hid = P['w_fc1'].shape[1]
        n_ada = P['n_ada']
        eps = cfg.norm_eps
        from .spec import attention_levels
        lvl_end, holes = attention_levels(cfg)
        scale = float(cfg.attn_scale)
        ah = depth * 6 * C
        # ---- backward: head
        TA, TB, ws = self.TA, self.TB, self.ws
        so = self.slab_off
        mo = self.misc_off
        self.dada.zero_()
        ops.gemm(self.dlogits, self.WT['head'], self.DU, M=M, N=C, K=V)
        self._wgrad(self.dlogits, V, self.UH, C, self.G_misc, mo['w_head'][0], mo['b_head'][0])
        ops.ln_modulate_bwd(self.Xs[depth], self.DU, ada, ah, n_ada, L, None, self.dX, self.dada, ah, ah + C, n_ada, M, C, eps, ws)
        # ---- backward: blocks
        for i in reversed(range(depth)):
            a0 = i * 6 * C
            G = self.G_layers
            go = i * self.slab
            # FFN branch
            ops.gated_grad(self.dX, self.F2[i], ada, a0 + C, n_ada, dp2[i].contiguous() if dp2 is not None else None, self.DF, self.dada, a0 + C, n_ada, B, L, C, ws)
            ops.gemm(self.DF, self.WT['fc2'], self.DH, M=M, N=hid, K=C, w_off=i * hid * C, act=ACT_GELU_GRAD, aux=self.A[i])   # dH = (dF W2) * gelu'(A)
            self._wgrad(self.DF, C, self.Hh[i], hid, G, go + so['w_fc2'], go + so['b_fc2'])
            ops.gemm(self.DH, self.WT['fc1'], self.DU, M=M, N=C, K=hid, w_off=i * C * hid)
            self._wgrad(self.DH, hid, self.U2[i], C, G, go + so['w_fc1'], go + so['b_fc1'])
            ops.ln_modulate_bwd(self.X1s[i], self.DU, ada, a0 + 3 * C, n_ada, L, self.dX, self.dX, self.dada, a0 + 3 * C, a0 + 5 * C, n_ada, M, C, eps, ws)
            # attention branch
            ops.gated_grad(self.dX, self.F1[i], ada, a0, n_ada, dp1[i].contiguous() if dp1 is not None else None, self.DF, self.dada, a0, n_ada, B, L, C, ws)
            ops.gemm(self.DF, self.WT['proj'], self.DU, M=M, N=C, K=C, w_off=i * C * C)
            self._wgrad(self.DF, C, self.O[i], C, G, go + so['w_proj'], go + so['b_proj'])
            ops.attention_bwd(self.arena[i], self.O[i], self.DU, self.LSE[i], self.DQKV, ws, B, H, L, L, scale, lvl_end, holes=holes)
            if cfg.uses_cos_attn:           # normalisation + learned temperature of basic_var.py:99-104
                ops.cos_qk_norm_bwd(self.arena[i], self.DQKV, B, H, L, L, P['scale_mul'], self.NORMS[i], self.DSM, sm_off=i * H)
                ops.colsum(self.DSM, H, G, M, H, ws, out_off=go + so['scale_mul'])
            ops.gemm(self.DQKV, self.WT['qkv'], self.DU, M=M, N=C, K=3 * C, w_off=i * C * 3 * C)
            self._wgrad(self.DQKV, 3 * C, self.U[i], C, G, go + so['w_qkv'], go + so['b_qkv'])      # the k-bias third is unused (zero_k_bias is a buffer)
            ops.ln_modulate_bwd(self.Xs[i], self.DU, ada, a0 + 2 * C, n_ada, L, self.dX, self.dX, self.dada, a0 + 2 * C, a0 + 4 * C, n_ada, M, C, eps, ws)
            if self.reducer is not None:
                self.reducer.ready(i)
        # ---- backward: adaLN parameter generator (one GEMM for all blocks + head)
        Bp = self.Bp
        dada_T = self.dada.to(T)
        dsilu = torch.empty(B, C, device=dev, dtype=torch.float32)
        ops.gemm(dada_T, self.WT['ada'], dsilu, M=B, N=C, K=n_ada)
        tA = torch.zeros(n_ada, Bp, device=dev, dtype=T)
        tB = torch.zeros(C, Bp, device=dev, dtype=T)
        ops.transpose(dada_T, tA, 1, B, n_ada, n_ada, ld_out=Bp)
        ops.transpose(cs, tB, 1, B, C, C, ld_out=Bp)
        ops.gemm(tA, tB, self.G_ada, M=n_ada, N=C, K=Bp)
        ops.colsum(self.dada, n_ada, self.G_ada, B, n_ada, ws, out_off=n_ada * C)
        if cfg.shared_aln:          # rows = blocks: column sums over the depth axis of the folded per-block gradients
            ops.colsum(self.G_ada, 6 * C * C, self.G_misc, depth, 6 * C * C, ws, out_off=mo['w_shared'][0])
            ops.colsum(self.G_ada, 6 * C, self.G_misc, depth, 6 * C, ws, a_off=n_ada * C, out_off=mo['b_shared'][0])
        dcond = torch.empty(B, C, device=dev, dtype=torch.float32)
        ops.silu_bwd(cond, dsilu, dcond)
        if self.reducer is not None:
            self.reducer.ready(depth)
        # ---- backward: embeddings (dX now holds d loss / d x0)
        Gm = self.G_misc
        ops.colsum(self.dX, L * C, Gm, B, L * C, ws, out_off=mo['pos'][0])
        for k, (b0, e0) in enumerate(zip(py.begin, py.end)):
            ops.colsum(Gm, C, Gm, e0 - b0, C, ws, a_off=mo['pos'][0] + b0 * C, out_off=mo['lvl'][0] + k * C)
        if cfg.type_pos:            # type_1L: first (control) half of every scale has id 1, the image half id 0 (control_var.py:103-108)
            for k, (b0, e0) in enumerate(zip(py.begin, py.end)):
                half = (e0 - b0) // 2
                for tid, r0 in (((1, b0), (0, b0 + half)) if self._mask_first else ((0, b0), (1, b0 + half))):
                    ops.colsum(Gm, C, Gm, half, C, ws, accumulate=(k > 0), a_off=mo['pos'][0] + r0 * C, out_off=mo['type'][0] + tid * C)
        Gm[mo['pos_start'][0]:mo['pos_start'][0] + fl * C].copy_(Gm[mo['pos'][0]:mo['pos'][0] + fl * C])
        Mt = B * self.ncode
        Mtp = _pad8(Mt)
        if cfg.separator:           # code rows are interleaved with the separator rows: gather them; the separators' gradient is the batch sum of their rows
            code = torch.from_numpy(py.code_positions()[fl:]).to(dev)
            dXw = self.dX.view(B, L, C)[:, code].contiguous()               # (B, ncode, C)
            self._word_embed_grads(tok, B, self.ncode + fl, fl, C, Mt, Mtp, src=dXw, src_has_first=False)
            mapping = cfg.special_mapping(self._mask_first)
            for j, pos in enumerate(int(q) for q in py.special_positions()):
                ops.colsum(self.dX, L * C, Gm, B, C, ws, a_off=pos * C, out_off=mo['special'][0] + mapping[j] * C)
        else:
            self._word_embed_grads(tok, B, L, fl, C, Mt, Mtp)
        cls_o, cnd_o = mo['class_emb'][0], mo['cond_embed'][0]
        Gm[cls_o:cls_o + mo['class_emb'][1]].zero_()
        Gm[cnd_o:cnd_o + mo['cond_embed'][1]].zero_()
        g_class = Gm[cls_o:cls_o + mo['class_emb'][1]]
        sos_row = (fl - 1) if self._mask_first else 0                 # position of the class token; image first: [sos, cond_token]
        ops.scatter_add_rows(self.dX, L * C, labels, g_class, B, C, src_off=sos_row * C)
        ops.scatter_add_rows(dcond, C, labels, g_class, B, C)
        if cfg.mask_factor == 2:
            ops.scatter_add_rows(self.dX, L * C, types, Gm[cnd_o:cnd_o + mo['cond_embed'][1]], B, C, src_off=(1 - sos_row) * C)
        if self.reducer is not None:
            self.reducer.ready(depth + 1)

    def _word_embed_grads(self, tok, B, L, fl, C, Mt, Mtp, src=None, src_has_first=True):
        """src: gradient rows of the word-embedded tokens - self.dX with L rows per sample of which the first fl are skipped, or a compact
        (B, L - fl, C) gather of them (separator models)"""
        cfg = self.cfg
        mo = self.misc_off
        Gm = self.G_misc
        src = self.dX if src is None else src
        rows = L if src_has_first else L - fl
        skip = fl if src_has_first else 0
        if cfg.cvae == 32 and C % 64 == 0 and tok.dtype == torch.float32 and src.dtype == torch.float32:
            # one pass over the token-major tensors (round 3): dW and db together, no transposes, no per-sample launches
            ops.wordembed_grad(src, C, rows, skip, tok, L - fl, B, C, cfg.cvae, Gm, mo['w_we'][0], mo['b_we'][0])
            return
        for b in range(B):      # gradient rows of sample b -> columns [b*(L-fl), (b+1)*(L-fl)) of TA32 ([C][Mtp])
            ops.transpose(src, self.TA32[b * (L - fl):], 1, L - fl, C, C, in_off=(b * rows + skip) * C, ld_out=Mtp)
        ops.transpose(tok, self.TB32, 1, Mt, cfg.cvae, cfg.cvae, ld_out=Mtp)
        ops.gemm(self.TA32, self.TB32, Gm, M=C, N=cfg.cvae, K=Mtp, c_off=mo['w_we'][0])
        for b in range(B):
            ops.colsum(src, C, Gm, L - fl, C, self.ws, accumulate=(b > 0), a_off=(b * rows + skip) * C, out_off=mo['b_we'][0])


class BucketReducer:
    """Gradient all-reduce of the data-parallel step (replaces DDP's bucketed all-reduce, train_control_var_hpu.py:604, and
    dist.py:100-116): one SUM all-reduce per bucket (bucket = one transformer layer's gradient slab, then the adaLN
    generator, then head + embeddings), launched on a side stream the moment the backward has finished that bucket, so that
    RCCL traffic over xGMI overlaps the remaining backward.  The 1/world mean is folded into the optimizer's gradient scale."""

    def __init__(self, buckets: Sequence[torch.Tensor], group=None, force: bool = False):
        """force=True issues the collectives even in a one-rank group (SUM over one rank = identity): the real slabs then travel the
        whole path - side stream, RCCL call, event hand-back - on a single GPU, which is how the path is tested without a node."""
        import torch.distributed as dist
        self.dist = dist
        self.buckets = list(buckets)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.cuda = len(self.buckets) > 0 and self.buckets[0].is_cuda
        self.stream = torch.cuda.Stream() if (self.cuda and self.active) else None
        self.handles: List = []
        self.launched: List[int] = []
        self.bytes_sent = 0

    def ready(self, idx: int):
        self.launched.append(idx)
        if not self.active:
            return
        self.bytes_sent += self.buckets[idx].numel() * self.buckets[idx].element_size()
        b = self.buckets[idx]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self.handles.append(self.dist.all_reduce(b, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.handles.append(self.dist.all_reduce(b, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for h in self.handles:
            h.wait()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.handles.clear()
        done, self.launched = self.launched, []
        return done


class FusedAdamW:
    """torch.optim.AdamW(betas=(0.9, 0.95)) over the model's parameters (train_control_var_hpu.py:631-633) with the
    decay / no-decay groups of filter_params, gradient-norm clipping (:244-245) and the all-reduce mean folded into one
    fused kernel per tensor.  `param_groups` is torch-shaped so that lr_wd_annealing() drives it unchanged."""

    def __init__(self, var, lr: float, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.0, nowd_keys=NOWD_KEYS):
        self.var = var
        self.betas, self.eps = betas, eps
        self.named = [(n, p) for n, p in var.named_parameters()]
        self.state = {n: (torch.zeros_like(p.data), torch.zeros_like(p.data)) for n, p in self.named}
        groups: Dict[str, dict] = {}                             # first-seen order, as filter_params (utils/lr_control.py:67-101)
        for n, p in self.named:
            d = decays(n, p.ndim, nowd_keys)
            groups.setdefault('D' if d else 'ND', dict(names=[], lr=lr, weight_decay=weight_decay if d else 0.0,
                                                       wd_sc=1.0 if d else 0.0, lr_sc=1.0))['names'].append(n)
        self.param_groups = list(groups.values())
        self.steps = 0
        self._partial = None
        self._out2 = None
        self._table = None
        self._table_sig = None
        self.fuse_copies = True                                  # False: rebuild every packed copy from the fp32 parameters after a step

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor], max_norm: float = 0.0, world: int = 1) -> torch.Tensor:
        """returns a device tensor [grad_norm (of the world-averaged gradient), clip coefficient]"""
        dev = self.named[0][1].device
        n = len(self.named)
        if self._partial is None:
            self._partial = torch.zeros(n * 256, device=dev, dtype=torch.float64)
            self._out2 = torch.ones(2, device=dev, dtype=torch.float32)
        # one launch over a device table of (param, grad, m, v) instead of 2 x ~830 per-tensor launches; the table is rebuilt
        # only when a buffer moved (new batch geometry -> new gradient slabs, .to(), load_state_dict)
        group_of = {name: gi for gi, g in enumerate(self.param_groups) for name in g['names']}
        # bf16 compute: the update kernel also writes the rounded value of every weight MATRIX into its slot of the stacked GEMM-ready
        # copies (models.VAR._matrix_copies), so the step does not re-read the fp32 masters to rebuild them (stack + cast were 3.4 ms)
        copies, fresh = self.var._matrix_copies() if self.fuse_copies and hasattr(self.var, '_matrix_copies') else ({}, ())
        if not set(copies) <= {name for name, _ in self.named}:
            copies, fresh = {}, ()
        sig = (tuple(p.data_ptr() for _, p in self.named) + tuple(grads[name].data_ptr() for name, _ in self.named) +
               tuple(c.data_ptr() for c in copies.values()))
        if self._table is None or self._table_sig != sig:
            entries = [(p.data, grads[name], self.state[name][0], self.state[name][1], group_of[name], copies.get(name)) for name, p in self.named]
            self._table = ops.adam_table(entries, dev)
            self._table_sig = sig
        ops.sumsq_multi(self._table, n, self._partial)
        ops.clip_coef(self._partial, n * 256, 1.0 / world, float(max_norm), self._out2)
        self.steps += 1
        coef = self._out2[1:]
        ops.adamw_multi(self._table, n, [float(g['lr']) for g in self.param_groups], [float(g['weight_decay']) for g in self.param_groups],
                        self.betas[0], self.betas[1], self.eps, self.steps, coef, 1.0 / world)
        if fresh:
            self.var._pack(fresh=fresh)                          # the matrices are current; biases / tables are rebuilt from the parameters
        else:
            self.var._packed = None                              # GEMM-ready copies are refreshed lazily
        return self._out2

    # ---- wire format of torch.optim.AdamW.state_dict() (what train_control_var_hpu.py:420-447 saves and resumes)
    def state_dict(self) -> Dict[str, object]:
        state, groups, at = {}, [], 0
        for g in self.param_groups:
            idx = list(range(at, at + len(g['names'])))
            at += len(idx)
            if self.steps > 0:
                for i, name in zip(idx, g['names']):
                    m, v = self.state[name]
                    state[i] = {'step': torch.tensor(float(self.steps)), 'exp_avg': m, 'exp_avg_sq': v}
            groups.append({'lr': g['lr'], 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': g['weight_decay'], 'amsgrad': False,
                           'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                           'decoupled_weight_decay': True, 'wd_sc': g['wd_sc'], 'lr_sc': g['lr_sc'], 'params': idx})
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        groups = sd['param_groups']
        if len(groups) != len(self.param_groups) or any(len(a['params']) != len(b['names']) for a, b in zip(groups, self.param_groups)):
            raise ValueError('loaded state dict has a different number / size of parameter groups')     # torch's own check
        steps = set()
        for saved, g in zip(groups, self.param_groups):
            for k in ('lr', 'weight_decay', 'wd_sc', 'lr_sc'):
                if k in saved:
                    g[k] = saved[k]
            if 'betas' in saved:
                self.betas = tuple(saved['betas'])
            self.eps = saved.get('eps', self.eps)
            for i, name in zip(saved['params'], g['names']):
                st = sd['state'].get(i)
                m, v = self.state[name]
                if st is None:
                    m.zero_(); v.zero_()
                    continue
                if tuple(st['exp_avg'].shape) != tuple(m.shape):
                    raise ValueError(f'optimizer state for {name}: shape {tuple(st["exp_avg"].shape)} != {tuple(m.shape)}')
                m.copy_(st['exp_avg']); v.copy_(st['exp_avg_sq'])
                steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError(f'per-parameter step counts differ ({sorted(steps)}); the fused optimizer keeps one')
        self.steps = steps.pop() if steps else 0


class Trainer:
    """One process per GPU; the step of train_control_var_hpu.py:130-255 (minus its logging / gradient-accumulation quirks)."""

    def __init__(self, var, vae, peak_lr: float, weight_decay: float, weight_decay_end: Optional[float] = None, sche: str = 'lin0',
                 warmup_it: float = 0, max_it: int = 1000, clip: float = 2.0, wp0: float = 0.005, wpe: float = 0.01, drop_path: bool = True,
                 train_mode: Optional[bool] = None, force_reducer: bool = False):
        """train_mode: True puts the model in train() (DropPath active, as the reference's loop runs it, train_control_var_hpu.py:136),
        False in eval(); None leaves the mode as the caller set it."""
        import torch.distributed as dist
        self.var, self.vae = var, vae
        if train_mode is not None:
            var.train(train_mode)
        self.engine = TrainEngine(var, drop_path=drop_path)
        self.opt = FusedAdamW(var, lr=peak_lr, weight_decay=weight_decay)
        self.sched = dict(sche=sche, peak_lr=peak_lr, wd=weight_decay, wd_end=weight_decay if weight_decay_end is None else weight_decay_end,
                          wp_it=warmup_it, max_it=max_it, wp0=wp0, wpe=wpe)
        self.clip = clip
        self.it = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.force_reducer = force_reducer
        self.comm = True                        # False: skip the gradient exchange (bench.py measures the exposed communication time with it)
        self.comm_group = None                  # process group of the gradient exchange (None = default); see set_comm_group
        self._reducer_for = None

    def set_comm_group(self, group):
        """run the gradient all-reduce in `group` from the next step on (launcher.channel_groups: one group per RCCL channel cap)"""
        self.comm_group = group
        self._reducer_for = None

    @torch.no_grad()
    def tokenize(self, images: torch.Tensor, masks: torch.Tensor, mask_first: bool = True):
        """frozen tokenizer + 'interleave_append' (mask first unless bidirectional drew image first): train_control_var_hpu.py:157-204"""
        # one tokenizer pass over [masks ; images] (the reference makes two calls, :160-176): the ten-scale residual quantiser is one latency-bound
        # launch of ~2.7 ms whatever the batch, and the conv stack sees twice the tiles.  Every image is encoded and quantised on its own and no kernel of the
        # tokenizer sums in a batch-dependent order, so the ids equal the two-call form's exactly in both precision modes
        # (tests/test_gpu_train.py::test_tokenize_one_pass_equals_the_two_call_form)
        B = masks.shape[0]
        both = self.vae.img_to_idxBl(torch.cat((masks, images), dim=0))
        hboth = self.vae.idxBl_to_h(both)
        mi, ii = [t[:B] for t in both], [t[B:] for t in both]
        mh, ih = [t[:B] for t in hboth], [t[B:] for t in hboth]
        if not mask_first:
            mi, ii, mh, ih = ii, mi, ih, mh
        x = torch.cat([torch.cat((a, b), 1) for a, b in zip(mh, ih)], dim=1)
        cfg = self.var.cfg
        if cfg.separator:           # train_control_var_hpu.py:214-224: the label of a separator is V + its special_embed row
            mapping = cfg.special_mapping(mask_first)
            parts = [mi[0], ii[0]]
            for k in range(1, len(mi)):
                for h, ids in enumerate((mi[k], ii[k])):
                    parts += [ids, ids.new_full((ids.shape[0], 1), cfg.vocab + mapping[2 * (k - 1) + h])]
            labels = torch.cat(parts, dim=1)
        else:
            labels = torch.cat([torch.cat((a, b), 1) for a, b in zip(mi, ii)], dim=1)
        return x, labels

    @torch.no_grad()
    def step(self, images, masks, cls, types, ignore_mask=None, drop_seed=None, mask_first: Optional[bool] = None) -> Dict[str, object]:
        """mask_first=None draws the order as the reference does (image first with probability 1/2 when the model is bidirectional,
        python `random`, train_control_var_hpu.py:192-199); pass the matching ``ignore_mask`` / ``ignore_mask_`` (:234)."""
        if mask_first is None:
            import random
            mask_first = not (getattr(self.var, 'bidirectional', False) and random.random() < 0.5)
        s = self.sched
        _, max_lr, _, max_wd = lr_wd_annealing(s['sche'], self.opt, s['peak_lr'], s['wd'], s['wd_end'], self.it, s['wp_it'], s['max_it'], wp0=s['wp0'], wpe=s['wpe'])
        x, labels = self.tokenize(images, masks, mask_first)
        self.engine._setup(x.shape[0])
        if (self.world > 1 or self.force_reducer) and self.comm:
            if self._reducer_for is not self.engine.buckets:
                self._reducer = BucketReducer(self.engine.buckets, group=self.comm_group, force=self.force_reducer)
                self._reducer_for = self.engine.buckets
            self.engine.reducer = self._reducer
        else:
            self.engine.reducer = None
        loss, _ = self.engine.forward_backward(cls, x, types, labels, ignore_mask, drop_seed, mask_first)
        if self.engine.reducer is not None:
            self.engine.reducer.wait()
        norm_coef = self.opt.step(self.engine.grads(), self.clip, self.world)
        self.engine._transposed_weights()
        self.it += 1
        return dict(loss=loss, grad_norm=norm_coef[0], clip_coef=norm_coef[1], lr=max_lr, wd=max_wd, mask_first=mask_first)


class _TeacherForcedFn(torch.autograd.Function):
    """Autograd bridge so that the reference's own training code (`logits = var(...)`; `loss.backward()`,
    train_control_var_hpu.py:207-241) works unchanged: forward keeps the activations inside the engine, backward runs the
    hand-written kernels and hands one gradient per parameter back to autograd."""

    @staticmethod
    def forward(ctx, engine, label_B, x, cond_type, mask_first, *params):
        logits = engine.forward_train(label_B, x, cond_type, None, mask_first)
        ctx.engine = engine
        B = x.shape[0]
        return logits.view(B, -1, logits.shape[-1])[:, :, :engine.cfg.head_out].clone()       # head_ld padding columns are not part of the API

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.engine
        if eng.cfg.head_ld != eng.cfg.head_out:
            eng.dlogits.zero_()
        eng.dlogits[:, :eng.cfg.head_out].copy_(dlogits.reshape(eng.M, -1))
        eng.backward()
        g = eng.grads()
        return (None, None, None, None, None) + tuple(g[n].clone() for n, _ in eng.var.named_parameters())


def teacher_forced_with_grad(var, label_B, x, cond_type, mask_first: bool = True):
    eng = getattr(var, '_train_engine', None)
    if eng is None:
        eng = var._train_engine = TrainEngine(var, drop_path=True)
    return _TeacherForcedFn.apply(eng, label_B, x, cond_type, mask_first, *[p for _, p in var.named_parameters()])
