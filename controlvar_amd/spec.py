"""Static description of the hot path: token pyramid geometry, phi sharing map and the
state_dict layout (key -> shape) the reference's checkpoints use.

Everything here is restated from the reference (read-only at /root/reference):
  * pyramid / begin_ends ........ models/control_var.py:55-67, models/var.py:41-50
  * phi sharing ticks ........... models/quant.py:282-290
  * ControlVAR / VAR parameters . models/control_var.py:70-213, models/basic_var.py:57-200
  * VQVAE parameters ............ models/vqvae.py:28-49, models/vae_modules.py:40-225, models/quant.py:33-37
The tables are verified against the reference's own ``state_dict()`` by
``tests/golden/make_golden.py`` (strict load) and by ``tests/test_spec.py`` against the
recorded key/shape digest.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

DEFAULT_PATCH_NUMS: Tuple[int, ...] = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
HEAD_DIM = 64
NUM_CLASSES = 1000
NUM_COND_TYPES = 4          # mask, canny, depth, normal; id 4 = unconditional (control_var.py:583)


def phi_index_map(num_scales: int, share_quant_resi: int = 4) -> List[int]:
    """scale index -> which shared phi conv it uses (quant.py:282-290).

    ticks = linspace(1/(3K), 1-1/(3K), K) when K == 4 else linspace(1/(2K), 1-1/(2K), K);
    phi(si) = argmin |ticks - si/(SN-1)|.
    """
    K = share_quant_resi
    if K == 1:
        return [0] * num_scales
    if K == 0:  # non-shared: one phi per scale, same nearest-tick rule over SN ticks
        K = num_scales
    ticks = np.linspace(1 / 3 / K, 1 - 1 / 3 / K, K) if K == 4 else np.linspace(1 / 2 / K, 1 - 1 / 2 / K, K)
    out = []
    for si in range(num_scales):
        at = si / (num_scales - 1) if num_scales > 1 else 0.0
        out.append(int(np.argmin(np.abs(ticks - at))))
    return out


@dataclass(frozen=True)
class Pyramid:
    """Token pyramid of one model: per-scale token counts and offsets.

    mask_factor 2 = ControlVAR 'interleave_append' ([control pn^2 ; image pn^2] per scale),
    mask_factor 1 = plain VAR.
    """
    patch_nums: Tuple[int, ...] = DEFAULT_PATCH_NUMS
    mask_factor: int = 2
    separator: bool = False                       # one special token behind every half of every scale but the first (control_var.py:58-66)
    l: Tuple[int, ...] = field(init=False)        # tokens per scale
    begin: Tuple[int, ...] = field(init=False)    # first token of scale k
    end: Tuple[int, ...] = field(init=False)      # one past the last token of scale k
    L: int = field(init=False)
    first_l: int = field(init=False)

    def __post_init__(self):
        l = tuple(self.mask_factor * (pn * pn + (1 if (self.separator and i != 0) else 0)) for i, pn in enumerate(self.patch_nums))
        end = tuple(int(x) for x in np.cumsum(l))
        begin = (0,) + end[:-1]
        object.__setattr__(self, 'l', l)
        object.__setattr__(self, 'begin', begin)
        object.__setattr__(self, 'end', end)
        object.__setattr__(self, 'L', end[-1])
        object.__setattr__(self, 'first_l', l[0])

    @property
    def num_scales(self) -> int:
        return len(self.patch_nums)

    def sp(self, k: int) -> int:
        """special tokens per half of scale k"""
        return 1 if (self.separator and k != 0) else 0

    def code_positions(self) -> np.ndarray:
        """(sum mf*pn^2,) positions of the CODE tokens (everything but the separators) inside the L-long sequence, in order"""
        out = []
        for k, (b, pn) in enumerate(zip(self.begin, self.patch_nums)):
            half = pn * pn + self.sp(k)
            for h in range(self.mask_factor):
                out += list(range(b + h * half, b + h * half + pn * pn))
        return np.asarray(out, dtype=np.int64)

    def special_positions(self) -> np.ndarray:
        """(n_special,) positions of the separator tokens, in sequence order (= special_embed row order for mask-first sequences)"""
        out = []
        for k, (b, pn) in enumerate(zip(self.begin, self.patch_nums)):
            if self.sp(k):
                half = pn * pn + 1
                out += [b + h * half + pn * pn for h in range(self.mask_factor)]
        return np.asarray(out, dtype=np.int64)

    def level_of_token(self) -> np.ndarray:
        """(L,) int64: scale index of every position (lvl_1L, control_var.py:158-166)."""
        return np.concatenate([np.full((n,), k, dtype=np.int64) for k, n in enumerate(self.l)])


# --------------------------------------------------------------------------------------
# state_dict layouts
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class VarConfig:
    depth: int
    mask_factor: int = 2            # 2: ControlVAR interleave_append, 1: plain VAR / 'replace'
    multi_cond: bool = True         # cond_embed present (every shipped yaml sets multi_cond: True)
    control: bool = True            # ControlVAR class (control_var.py) vs VAR class (var.py)
    patch_nums: Tuple[int, ...] = DEFAULT_PATCH_NUMS
    vocab: int = 4096
    cvae: int = 32
    num_classes: int = NUM_CLASSES
    embed_dim: int = 0              # 0 -> 64*depth (models/__init__.py:14,39)
    num_heads: int = 0              # 0 -> depth
    norm_eps: float = 1e-6
    tau: float = 4.0
    cos_attn: bool = False          # ControlVAR forces True when depth == 30 (control_var.py:35)
    mlp_ratio: float = 4.0
    cond_drop_rate: float = 0.1
    drop_path_rate: float = 0.0     # stochastic depth: block i drops its branches with rate linspace(0, drop_path_rate, depth)[i] (control_var.py:123-124)
    shared_aln: bool = False        # N4: one SharedAdaLin for all blocks + per-block ada_gss (control_var.py:120, basic_var.py:194-205)
    type_pos: bool = False          # N4: type_embed added per control / image half (control_var.py:99-117,423,482,623)
    bidirectional: bool = False     # N4: image-first order allowed (mask_first=False: first two tokens and type ids swapped; control_var.py:403-407,587,624)
    sa_block: bool = False          # N4: aln < 0 -> SABlock (affine LayerNorms, no adaLN; basic_var.py:128-176) + head = Sequential(LN, Linear)
    layer_scale: float = -1.0       # SABlock only: >= 0 -> learned per-channel gamma1 / gamma2 (basic_var.py:145-149)
    separate_decoding: bool = False # N4: per scale the control half is decoded before the image half (control_var.py:170-180,428-485)
    indep: bool = False             # N4: with separate_decoding, the two halves of a scale are blind to each other (control_var.py:182-191);
                                    #     without it the flag only makes inference pass (all-visible) slices of the mask (:283,:497)
    separator: bool = False         # N4: +18 special tokens / head columns / special_embed rows (control_var.py:58-66,201-210).  Upstream indexes
                                    #     special_embed with V + k (:549,606) and raises; built with the evidently intended index k

    @property
    def C(self) -> int:
        return self.embed_dim or 64 * self.depth

    @property
    def H(self) -> int:
        return self.num_heads or self.depth

    @property
    def pyramid(self) -> Pyramid:
        return Pyramid(self.patch_nums, self.mask_factor, bool(self.separator) and self.mask_factor == 2)

    @property
    def n_special(self) -> int:
        return (len(self.patch_nums) - 1) * self.mask_factor if self.separator else 0

    @property
    def head_out(self) -> int:
        """columns of the head: the V codes + the separator labels (control_var.py:201-204)"""
        return self.vocab + self.n_special

    @property
    def head_ld(self) -> int:
        """head_out rounded up to 8 columns: the packed head / logits buffers carry zero-weight, -1e30-bias padding columns so that the
        GEMM epilogue stays on its 16-byte path (softmax / CE weight of a padding column is exactly 0)"""
        return (self.head_out + 7) // 8 * 8

    @staticmethod
    def special_mapping(mask_first: bool = True):
        """special_embed row / label offset of the i-th separator in sequence order (control_var.py:543,605; train_control_var_hpu.py:215)"""
        return list(range(18)) if mask_first else [i + 1 if i % 2 == 0 else i - 1 for i in range(18)]

    @property
    def uses_cos_attn(self) -> bool:
        return (self.depth == 30) if self.control else self.cos_attn

    @property
    def attn_scale(self) -> float:
        # basic_var.py:66-71: cos-attn uses scale 1, else 1/sqrt(head_dim)/tau
        return 1.0 if self.uses_cos_attn else 1.0 / np.sqrt(self.C // self.H) / self.tau


def attention_levels(cfg: VarConfig):
    """(lvl_end, holes) describing ``attn_bias_for_masking`` for the attention kernels (include/cvar.h cvar_attention): a query at
    position p sees keys [0, lvl_end[level(p)]) minus holes[level(p)] = [lo, hi).
      default .................. levels = scales (control_var.py:158-168);
      separate_decoding ........ levels = half scales: control queries stop at the end of their own half, image queries see the
                                 whole scale (:170-180);
      separate_decoding+indep .. the image half additionally does not see the control half of its own scale (:182-191).
    holes is None when no level has one."""
    py = cfg.pyramid
    if not (cfg.separate_decoding and cfg.mask_factor == 2):
        return list(py.end), None
    ends, holes = [], []
    for b, e in zip(py.begin, py.end):
        half = (e - b) // 2                           # pn^2 (+ 1 separator)
        ends += [b + half, e]
        holes += [(0, 0), (b, b + half) if cfg.indep else (0, 0)]
    return ends, (holes if cfg.indep else None)


def attention_bias_matrix(cfg: VarConfig):
    """the reference's own construction of the (L, L) additive {0, -inf} buffer (control_var.py:158-191), as a numpy bool 'visible' map"""
    py = cfg.pyramid
    L = py.L
    lvl = py.level_of_token()
    vis = lvl[:, None] >= lvl[None, :]
    if cfg.separate_decoding and cfg.mask_factor == 2:
        d, dT = np.zeros(L, np.int64), np.zeros(L, np.int64)
        for i, (b, e) in enumerate(zip(py.begin, py.end)):
            h = (e - b) // 2
            d[b:b + h], d[b + h:e] = 1 + 4 * i, 3 + 4 * i
            dT[b:b + h], dT[b + h:e] = 1 + 4 * i, 2 + 4 * i
        vis = d[:, None] >= dT[None, :]
        if cfg.indep:
            for i, (b, e) in enumerate(zip(py.begin, py.end)):
                h = (e - b) // 2
                d[b:b + h], d[b + h:e] = 3 + 4 * i, 1 + 4 * i
                dT[b:b + h], dT[b + h:e] = 2 + 4 * i, 0 + 4 * i
            vis = vis & (d[:, None] >= dT[None, :])
    return vis


def var_state_shapes(cfg: VarConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """key -> (shape, kind) with kind in {'param', 'buffer'}; dtype fp32 except lvl_1L (int64)."""
    C, V, L = cfg.C, cfg.vocab, cfg.pyramid.L
    hid = round(C * cfg.mlp_ratio)
    sd: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    sd['pos_start'] = ((1, cfg.pyramid.first_l, C), 'param')
    sd['pos_1LC'] = ((1, L, C), 'param')
    if cfg.type_pos:                                 # registered before lvl_1L upstream (control_var.py:99-117 vs :160-168)
        sd['type_1L'] = ((1, L), 'buffer')
        sd['type_1L_'] = ((1, L), 'buffer')
    sd['lvl_1L'] = ((1, L), 'buffer')
    sd['attn_bias_for_masking'] = ((1, 1, L, L), 'buffer')
    sd['word_embed.weight'] = ((C, cfg.cvae), 'param')
    sd['word_embed.bias'] = ((C,), 'param')
    sd['class_emb.weight'] = ((cfg.num_classes + 1, C), 'param')
    sd['lvl_embed.weight'] = ((len(cfg.patch_nums), C), 'param')
    if cfg.type_pos:
        sd['type_embed.weight'] = ((cfg.mask_factor, C), 'param')
    if cfg.shared_aln and not cfg.sa_block:
        sd['shared_ada_lin.1.weight'] = ((6 * C, C), 'param')
        sd['shared_ada_lin.1.bias'] = ((6 * C,), 'param')
    for i in range(cfg.depth):
        p = f'blocks.{i}.'
        if cfg.sa_block:
            if cfg.layer_scale >= 0:
                sd[p + 'gamma1'] = ((C,), 'param')
                sd[p + 'gamma2'] = ((C,), 'param')
            sd[p + 'norm1.weight'] = ((C,), 'param')
            sd[p + 'norm1.bias'] = ((C,), 'param')
        elif cfg.shared_aln:
            sd[p + 'ada_gss'] = ((1, 1, 6, C), 'param')
        sd[p + 'attn.q_bias'] = ((C,), 'param')
        sd[p + 'attn.v_bias'] = ((C,), 'param')
        sd[p + 'attn.zero_k_bias'] = ((C,), 'buffer')
        if cfg.uses_cos_attn:
            sd[p + 'attn.scale_mul_1H11'] = ((1, cfg.H, 1, 1), 'param')
        sd[p + 'attn.mat_qkv.weight'] = ((3 * C, C), 'param')
        sd[p + 'attn.proj.weight'] = ((C, C), 'param')
        sd[p + 'attn.proj.bias'] = ((C,), 'param')
        if cfg.sa_block:
            sd[p + 'norm2.weight'] = ((C,), 'param')
            sd[p + 'norm2.bias'] = ((C,), 'param')
        sd[p + 'ffn.fc1.weight'] = ((hid, C), 'param')
        sd[p + 'ffn.fc1.bias'] = ((hid,), 'param')
        sd[p + 'ffn.fc2.weight'] = ((C, hid), 'param')
        sd[p + 'ffn.fc2.bias'] = ((C,), 'param')
        if not (cfg.shared_aln or cfg.sa_block):
            sd[p + 'ada_lin.1.weight'] = ((6 * C, C), 'param')
            sd[p + 'ada_lin.1.bias'] = ((6 * C,), 'param')
    if cfg.sa_block:                                 # control_var.py:205-207: MultiInpIdentity + Sequential(norm, Linear)
        sd['head.0.weight'] = ((C,), 'param')
        sd['head.0.bias'] = ((C,), 'param')
        sd['head.1.weight'] = ((cfg.head_out, C), 'param')
        sd['head.1.bias'] = ((cfg.head_out,), 'param')
    else:
        sd['head_nm.ada_lin.1.weight'] = ((2 * C, C), 'param')
        sd['head_nm.ada_lin.1.bias'] = ((2 * C,), 'param')
        sd['head.weight'] = ((cfg.head_out, C), 'param')
        sd['head.bias'] = ((cfg.head_out,), 'param')
    if cfg.separator:
        sd['special_embed.weight'] = ((cfg.n_special, C), 'param')
    if cfg.control and cfg.multi_cond:
        sd['cond_embed.weight'] = ((NUM_COND_TYPES + 1, C), 'param')
    return sd


@dataclass(frozen=True)
class VaeConfig:
    vocab: int = 4096
    z_channels: int = 32
    ch: int = 160
    ch_mult: Tuple[int, ...] = (1, 1, 2, 2, 4)
    num_res_blocks: int = 2
    share_quant_resi: int = 4
    quant_resi: float = 0.5
    patch_nums: Tuple[int, ...] = DEFAULT_PATCH_NUMS
    gn_groups: int = 32
    gn_eps: float = 1e-6

    @property
    def phi_map(self) -> List[int]:
        return phi_index_map(len(self.patch_nums), self.share_quant_resi)


def _conv(sd, name, cout, cin, k):
    sd[name + '.weight'] = ((cout, cin, k, k), 'param')
    sd[name + '.bias'] = ((cout,), 'param')


def _norm(sd, name, c):
    sd[name + '.weight'] = ((c,), 'param')
    sd[name + '.bias'] = ((c,), 'param')


def _resblock(sd, name, cin, cout):
    _norm(sd, name + '.norm1', cin)
    _conv(sd, name + '.conv1', cout, cin, 3)
    _norm(sd, name + '.norm2', cout)
    _conv(sd, name + '.conv2', cout, cout, 3)
    if cin != cout:
        _conv(sd, name + '.nin_shortcut', cout, cin, 1)


def _attnblock(sd, name, c):
    _norm(sd, name + '.norm', c)
    _conv(sd, name + '.qkv', 3 * c, c, 1)
    _conv(sd, name + '.proj_out', c, c, 1)


def vae_state_shapes(cfg: VaeConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """VQVAE state_dict layout (vqvae.py:28-49; vae_modules.py:99-225; quant.py:31-37)."""
    sd: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    ch, mult, nres = cfg.ch, cfg.ch_mult, cfg.num_res_blocks
    nlev = len(mult)
    in_mult = (1,) + tuple(mult)
    # ---- encoder (vae_modules.py:112-142)
    _conv(sd, 'encoder.conv_in', ch, 3, 3)
    block_in = ch
    for lv in range(nlev):
        block_in = ch * in_mult[lv]
        block_out = ch * mult[lv]
        for b in range(nres):
            _resblock(sd, f'encoder.down.{lv}.block.{b}', block_in, block_out)
            block_in = block_out
            if lv == nlev - 1:
                _attnblock(sd, f'encoder.down.{lv}.attn.{b}', block_in)
        if lv != nlev - 1:
            _conv(sd, f'encoder.down.{lv}.downsample.conv', block_in, block_in, 3)
    _resblock(sd, 'encoder.mid.block_1', block_in, block_in)
    _attnblock(sd, 'encoder.mid.attn_1', block_in)
    _resblock(sd, 'encoder.mid.block_2', block_in, block_in)
    _norm(sd, 'encoder.norm_out', block_in)
    _conv(sd, 'encoder.conv_out', cfg.z_channels, block_in, 3)
    # ---- decoder (vae_modules.py:176-208)
    block_in = ch * mult[nlev - 1]
    _conv(sd, 'decoder.conv_in', block_in, cfg.z_channels, 3)
    _resblock(sd, 'decoder.mid.block_1', block_in, block_in)
    _attnblock(sd, 'decoder.mid.attn_1', block_in)
    _resblock(sd, 'decoder.mid.block_2', block_in, block_in)
    for lv in reversed(range(nlev)):
        block_out = ch * mult[lv]
        for b in range(nres + 1):
            _resblock(sd, f'decoder.up.{lv}.block.{b}', block_in, block_out)
            block_in = block_out
            if lv == nlev - 1:
                _attnblock(sd, f'decoder.up.{lv}.attn.{b}', block_in)
        if lv != 0:
            _conv(sd, f'decoder.up.{lv}.upsample.conv', block_in, block_in, 3)
    _norm(sd, 'decoder.norm_out', block_in)
    _conv(sd, 'decoder.conv_out', 3, block_in, 3)
    # ---- quantizer (quant.py:27-37)
    nphi = cfg.share_quant_resi if cfg.share_quant_resi > 1 else None
    if cfg.share_quant_resi == 1:
        _conv(sd, 'quantize.quant_resi.qresi', cfg.z_channels, cfg.z_channels, 3)
    elif cfg.share_quant_resi == 0:
        for k in range(len(cfg.patch_nums)):
            _conv(sd, f'quantize.quant_resi.{k}', cfg.z_channels, cfg.z_channels, 3)
    else:
        for k in range(nphi):
            _conv(sd, f'quantize.quant_resi.qresi_ls.{k}', cfg.z_channels, cfg.z_channels, 3)
    sd['quantize.ema_vocab_hit_SV'] = ((len(cfg.patch_nums), cfg.vocab), 'buffer')
    sd['quantize.embedding.weight'] = ((cfg.vocab, cfg.z_channels), 'param')
    _conv(sd, 'quant_conv', cfg.z_channels, cfg.z_channels, 3)
    _conv(sd, 'post_quant_conv', cfg.z_channels, cfg.z_channels, 3)
    return sd


def algorithmic_gflop_per_row(cfg: VarConfig, n_ada: int = 1) -> Dict[str, float]:
    """2*MAC GFLOP of the transformer for ONE sequence row over a full generation
    (SURVEY.md section 8(d) / BASELINE.md section 3).  n_ada=1: ada_lin hoisted."""
    C, V, depth = cfg.C, cfg.vocab, cfg.depth
    py = cfg.pyramid
    sumL = py.L
    sum_lL = sum(l * e for l, e in zip(py.l, py.end))
    linear = depth * 24 * C * C * sumL
    attn = depth * 4 * C * sum_lL
    ada = depth * 12 * C * C * n_ada + 4 * C * C * 10
    head = 2 * C * V * sumL
    wemb = 2 * cfg.cvae * C * (sumL - py.first_l)
    tot = linear + attn + ada + head + wemb
    return {k: v / 1e9 for k, v in dict(linear=linear, attn=attn, ada=ada, head=head, word_embed=wemb, total=tot).items()}


VAE_DECODE_GFLOP = 393.7    # per 256^2 image, ch=160 (FlopCounter-measured)
VAE_ENCODE_GFLOP = 215.1
