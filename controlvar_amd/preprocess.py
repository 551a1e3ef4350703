"""Tokenizer input pipeline on the device.

What the reference does per sample on the CPU (datasets/imagenetC.py:128-188 + datasets/transforms_image.py:103-121):

    image = PIL RGB;  cond = condition image .resize(image.size)                      (PIL default filter: BICUBIC)
    Resize(288, LANCZOS) on both -> RandomCrop(256) or CenterCrop(256) -> RandomHorizontalFlip -> ToTensor -> Normalize(.5, .5)
    ignore_mask / ignore_mask_ from the background of a segmentation-mask condition

Here JPEG decoding stays on the host; everything after it runs on the GPU from the decoded uint8 HWC array:

* ``resample_tables`` restates Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` (src/libImaging/Resample.c) in
  double precision on the host - a few hundred numbers per axis;
* ``cvar_resample_u8`` applies them (two integer passes, horizontal first, uint8 between the passes as in
  ``ImagingResample``) - bit-identical to ``Image.resize``;
* ``cvar_crop_flip_normalize`` / ``cvar_ignore_mask`` finish the sample.

Random choices (crop offset, flip) are inputs: the caller owns the RNG, as with the reference's ``random`` module.
Segmentation conditions (imagenetC.py:15-37): COCO RLE annotations (compressed strings as they lie in the reference's JSON files,
uncompressed run lengths, or decoded masks) -> colour map on the device (cvar_rle_paint); see _runs_of for what pins each form.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2                       # Resample.c


def _sinc(x: float) -> float:                     # sinc_filter
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x                        # math.sin is the C library's sin, the one Pillow calls


def _lanczos(x: float) -> float:                  # lanczos_filter: truncated sinc, support 3
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def _bicubic(x: float) -> float:                  # bicubic_filter, a = -0.5, support 2
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {'lanczos': (_lanczos, 3.0), 'bicubic': (_bicubic, 2.0)}


def resample_tables(in_size: int, out_size: int, filt: str) -> Tuple[np.ndarray, np.ndarray, int]:
    """bounds (out, 2) int32, fixed-point coeffs (out, ksize) int32, ksize: precompute_coeffs over the full axis (in0 = 0,
    in1 = in_size) followed by normalize_coeffs_8bpc, in the C code's operation order (plain Python floats are C doubles)."""
    f, fsupport = FILTERS[filt]
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    fixed = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)                 # (int) truncates toward zero, then the clamp
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v                                                # same left-to-right accumulation as the C loop
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            fixed[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[xx] = (xmin, xmax)
    return bounds, fixed, ksize


def resized_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision F.resize with an int size (shorter side -> size, the longer one int(size * long / short))"""
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def center_crop_offsets(h: int, w: int, out_h: int, out_w: int) -> Tuple[int, int]:
    """torchvision F.center_crop: int(round((h - out_h) / 2.0)) with Python's round (half to even)"""
    return int(round((h - out_h) / 2.0)), int(round((w - out_w) / 2.0))


# ------------------------------------------------------------------------------------------------------------- device side
def _tables_to_device(tables, device):
    import torch
    b, k, ksize = tables
    return torch.from_numpy(b).to(device), torch.from_numpy(np.ascontiguousarray(k)).to(device), ksize


def vertical_pass_first(h: int, w: int, out_h: int, out_w: int) -> bool:
    """Pass order of ``Image.resize``: horizontal then vertical, except that Pillow 12.2 runs the vertical pass first on very
    tall sources that shrink vertically (h > 100 * w and out_h < h).  The order matters because the intermediate image is
    rounded to uint8.  Established against the installed Pillow (tools/fuzz_resample.py; the threshold is exact: h = 100 w
    still goes horizontal first); no ImageNet-shaped input reaches it."""
    return out_w != w and out_h != h and h > 100 * w and out_h < h


def resize_u8(img, out_h: int, out_w: int, filt: str):
    """``Image.resize((out_w, out_h), filt)`` of a uint8 HWC device tensor: two separable passes (ImagingResample) in Pillow's order"""
    import torch
    from . import ops
    h, w, c = img.shape
    cur = img.contiguous()
    for axis in ((1, 0) if vertical_pass_first(h, w, out_h, out_w) else (0, 1)):
        if axis == 0 and out_w != w:
            b, k, ks = _tables_to_device(resample_tables(w, out_w, filt), img.device)
            nxt = torch.empty(h, out_w, c, dtype=torch.uint8, device=img.device)
            ops.resample_u8(cur, h, w, c, 0, out_w, b, k, ks, nxt)
            cur, w = nxt, out_w
        elif axis == 1 and out_h != h:
            b, k, ks = _tables_to_device(resample_tables(h, out_h, filt), img.device)
            nxt = torch.empty(out_h, w, c, dtype=torch.uint8, device=img.device)
            ops.resample_u8(cur, h, w, c, 1, out_h, b, k, ks, nxt)
            cur, h = nxt, out_h
    return cur


def preprocess_pair(image_u8, cond_u8, image_size: int = 256, mid_res: float = 1.125, crop: Optional[Tuple[int, int]] = None, flip: bool = False):
    """One (image, condition) sample: uint8 HWC device tensors -> two fp32 (3, image_size, image_size) tensors in [-1, 1].
    crop = (top, left) for the RandomCrop branch, None for the CenterCrop branch (create_image_mask_transforms, :103-121)."""
    import torch
    from . import ops
    h, w, _ = image_u8.shape
    if tuple(cond_u8.shape[:2]) != (h, w):
        cond_u8 = resize_u8(cond_u8, h, w, 'bicubic')                           # cond.resize(image.size), imagenetC.py:147
    mid = round(mid_res * image_size)
    nh, nw = resized_size(h, w, mid)
    outs = []
    top, left = crop if crop is not None else center_crop_offsets(nh, nw, image_size, image_size)
    for src in (image_u8, cond_u8):
        r = resize_u8(src, nh, nw, 'lanczos') if (nh, nw) != (h, w) else src.contiguous()
        dst = torch.empty(3, image_size, image_size, dtype=torch.float32, device=src.device)
        ops.crop_flip_normalize(r, nh, nw, 3, top, left, image_size, image_size, bool(flip), dst)
        outs.append(dst)
    return outs[0], outs[1]


def ignore_masks(cond, patch_nums: Sequence[int], first_masked_scale: int = 5, separator: bool = False) -> Dict[str, object]:
    """imagenetC.py:152-178 for a batch of segmentation-mask conditions (B, 3, H, W): {'ignore_mask', 'ignore_mask_'} (B, L).
    ``separator`` (:158,:169-170): every half-scale but the first pair starts with one extra always-kept token (L = 1378)."""
    import torch
    from . import ops
    B, _, H, W = cond.shape
    L = sum(2 * p * p for p in patch_nums)
    out = {}
    if separator:
        pos, at = [], 0
        for si, p in enumerate(patch_nums):
            for _half in range(2):
                at += 1 if si else 0
                pos += range(at, at + p * p)
                at += p * p
        pos = torch.tensor(pos, dtype=torch.int64, device=cond.device)
        L_sep = at
    for key, image_first in (('ignore_mask', 0), ('ignore_mask_', 1)):
        t = torch.empty(B, L, dtype=torch.float32, device=cond.device)
        ops.ignore_mask(cond.contiguous(), B, H, W, patch_nums, first_masked_scale, image_first, t, L)
        if separator:                                                      # placement only: the separator tokens are never ignored
            t = torch.ones(B, L_sep, dtype=torch.float32, device=cond.device).index_copy_(1, pos, t)
        out[key] = t
    return out


# ------------------------------------------------------------------------------------------------------------- segmentation condition
def create_color_map() -> np.ndarray:
    """imagenetC.py:31-37: the 5x5x5 colour cube without black -> (124, 3)"""
    lv = [0, 64, 128, 192, 255]
    return np.array([[r, g, b] for r in lv for g in lv for b in lv])[1:]


def runs_from_mask(mask) -> list:
    """binary (h, w) mask -> uncompressed COCO run lengths: column-major, first run counts zeros (possibly 0 long).  What a caller
    holding a decoded mask (pycocotools ``mask.decode``, the reference's own step at imagenetC.py:21) passes on."""
    m = np.asarray(mask)
    if m.ndim != 2:
        raise ValueError(f'mask must be (h, w), got {m.shape}')
    flat = (m != 0).T.reshape(-1)
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], change, [flat.size]])
    runs = np.diff(edges).tolist()
    return ([0] + runs) if flat.size and flat[0] else runs


_WARNED_RLE_STRING = False


def rle_from_string(s) -> list:
    """COCO compressed RLE string -> run lengths (pycocotools common/maskApi.c rleFrString, restated from the published algorithm:
    6-bit characters offset by 48, 5 payload bits + continuation bit 0x20, sign-extended by bit 0x10 of the last group, and every
    value after the third is a delta against the value two places back).  UNPINNED THIRD-PARTY FORMAT: pycocotools is neither vendored
    by the reference nor installed in the build image, so the known answers in tests/test_preprocess.py are hand-derived."""
    if isinstance(s, str):
        s = s.encode('ascii')
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, 1
        while more:
            if p >= len(s):
                raise ValueError('malformed COCO RLE string (ends inside a run: the last character has its continuation bit set)')
            c = s[p] - 48
            if not 0 <= c < 64:                       # the alphabet is the 64 characters '0' (48) .. 'o' (111)
                raise ValueError(f'malformed COCO RLE string (character {s[p]!r} at position {p} is outside the 6-bit alphabet)')
            x |= (c & 0x1f) << (5 * k)
            more = c & 0x20
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def rle_to_string(cnts: Sequence[int]) -> str:
    """inverse of rle_from_string (maskApi.c rleToString); used by the tests and tools/coco_rle_string.py"""
    out = bytearray()
    for i, x in enumerate(cnts):
        x = int(x)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(c + 48)
    return out.decode('ascii')


def _runs_of(segmentation) -> Tuple[list, int, int]:
    """a segmentation is a decoded (h, w) mask, a COCO RLE dict with UNCOMPRESSED counts (a list of run lengths, column-major, zeros
    first), or - the on-disk form the reference reads (imagenetC.py:20-21: json -> ``mask_utils.decode``) - an RLE dict whose counts
    are pycocotools' compressed STRING.  Strings go through pycocotools itself when it is importable (the pinned route, and the
    reference needs it installed anyway); otherwise through ``rle_from_string`` with a one-time warning that the codec is an unpinned
    restatement of a third-party wire format."""
    global _WARNED_RLE_STRING
    if isinstance(segmentation, np.ndarray) or (hasattr(segmentation, 'shape') and not isinstance(segmentation, dict)):
        m = np.asarray(segmentation)
        return runs_from_mask(m), int(m.shape[0]), int(m.shape[1])
    h, w = segmentation['size']
    counts = segmentation['counts']
    if isinstance(counts, (str, bytes)):
        try:
            from pycocotools import mask as mask_utils                         # absent in the build image
        except ImportError:
            mask_utils = None
        if mask_utils is not None:
            m = np.asarray(mask_utils.decode(segmentation))
            if m.ndim != 2 or (int(m.shape[0]), int(m.shape[1])) != (int(h), int(w)):
                raise ValueError(f'decoded mask is {tuple(m.shape)}, size says {h}x{w}')
            return runs_from_mask(m), int(m.shape[0]), int(m.shape[1])
        if not _WARNED_RLE_STRING:
            import warnings
            warnings.warn('controlvar_amd.preprocess: decoding a compressed COCO RLE string with the built-in restatement of pycocotools\' '
                          'maskApi.c (unpinned third-party format: pycocotools is not installed, so it could not be checked against it)',
                          RuntimeWarning, stacklevel=3)
            _WARNED_RLE_STRING = True
        runs = rle_from_string(counts)
        if any(r < 0 for r in runs):
            raise ValueError('malformed COCO RLE string (negative run length)')
    else:
        runs = [int(c) for c in counts]
    if sum(runs) != h * w:
        raise ValueError(f'RLE covers {sum(runs)} pixels, size says {h}x{w}')
    return runs, h, w


def annotation_colours(anns, image_size: int, colormap: Optional[np.ndarray] = None, min_area: int = 5000):
    """host half of process_anns (imagenetC.py:15-29): area filter, centroid of each mask from its runs, colour index
    ``(x * y) % len(colormap)`` with ``x = int(mean(cols) // (W / 11))``, ``y = int(mean(rows) // (H / 11))``.
    Returns (run_ends int32, ann_offsets int32, colours uint8 (n, 3)) ready for cvar_rle_paint."""
    colormap = create_color_map() if colormap is None else colormap
    ends, offsets, colours = [], [0], []
    for ann in anns:
        if ann['area'] < min_area:
            continue
        runs, h, w = _runs_of(ann['segmentation'])
        if (h, w) != (image_size, image_size):
            raise ValueError(f'mask of size {h}x{w}, expected {image_size} (the reference paints into a fixed 512 canvas)')
        e = np.cumsum(np.asarray(runs, np.int64))
        starts, stops = e[0::2][: len(e[1::2])], e[1::2]                      # the "ones" runs [start, stop) in column-major order
        n = int((stops - starts).sum())
        # sum of column / row indices over all set pixels, exactly (integers): position i -> column i // h, row i % h
        sx = sy = 0
        for a, b in zip(starts.tolist(), stops.tolist()):
            idx = np.arange(a, b, dtype=np.int64)
            sx += int((idx // h).sum()); sy += int((idx % h).sum())
        mean_x, mean_y = np.float64(sx) / np.float64(n), np.float64(sy) / np.float64(n)
        x = int(np.floor_divide(mean_x, w / 11))
        y = int(np.floor_divide(mean_y, h / 11))
        assert x * y < 124
        colours.append(colormap[(x * y) % len(colormap)])
        ends.append(e)
        offsets.append(offsets[-1] + len(e))
    run_ends = np.concatenate(ends).astype(np.int32) if ends else np.zeros(0, np.int32)
    return run_ends, np.asarray(offsets, np.int32), np.asarray(colours, np.uint8).reshape(-1, 3)


def paint_annotations(anns, image_size: int = 512, device='cuda'):
    """process_anns(anns, 512, colormap).astype(uint8) as a (512, 512, 3) uint8 device tensor"""
    import torch
    from . import ops
    run_ends, offsets, colours = annotation_colours(anns, image_size)
    out = torch.empty(image_size, image_size, 3, dtype=torch.uint8, device=device)
    n = len(offsets) - 1
    if n == 0:
        return out.zero_()
    ops.rle_paint(torch.from_numpy(run_ends).to(device), torch.from_numpy(offsets).to(device), torch.from_numpy(colours).to(device),
                  n, image_size, image_size, out)
    return out
