"""TEST INFRASTRUCTURE (oracle): numpy restatement of Pillow's 8-bit separable resampler and of the torchvision transform
chain of the reference's input pipeline (datasets/transforms_image.py:103-121, datasets/imagenetC.py:147-185).

The algorithm lives in third-party dependencies of the reference (Pillow's src/libImaging/Resample.c - ImagingResample,
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc; torchvision's
functional.resize / center_crop / to_tensor / normalize).  Pinned against Pillow itself: tests/golden/make_golden.py::case_preprocess
records Image.resize outputs (Pillow 12.2.0) for seeded inputs, and tests compare with the installed Pillow directly.
Only tests import this module.
"""
import numpy as np

from controlvar_amd.preprocess import PRECISION_BITS, center_crop_offsets, resample_tables, resized_size


def resample_pass(img: np.ndarray, axis: int, out: int, filt: str) -> np.ndarray:
    """img (H, W, C) uint8; axis 0 = horizontal (width -> out), 1 = vertical (height -> out)"""
    h, w, c = img.shape
    n_in = w if axis == 0 else h
    bounds, coeffs, _ = resample_tables(n_in, out, filt)
    src = img.astype(np.int64)
    res = np.empty((h, out, c) if axis == 0 else (out, w, c), np.uint8)
    for o in range(out):
        lo, n = bounds[o]
        k = coeffs[o, :n].astype(np.int64)
        if axis == 0:
            ss = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, lo:lo + n, :], k, axes=([1], [0]))
            res[:, o, :] = np.clip(ss >> PRECISION_BITS, 0, 255)
        else:
            ss = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[lo:lo + n, :, :], k, axes=([0], [0]))
            res[o, :, :] = np.clip(ss >> PRECISION_BITS, 0, 255)
    return res


def resize(img: np.ndarray, out_h: int, out_w: int, filt: str) -> np.ndarray:
    """Image.resize((out_w, out_h), filt): two passes with a uint8 image in between (ImagingResample); horizontal first
    except for very tall shrinking sources (Pillow 12.2: h > 100 w and out_h < h - observed, see tools/fuzz_resample.py)"""
    h, w = img.shape[:2]
    cur = img
    vertical_first = out_w != w and out_h != h and h > 100 * w and out_h < h
    for axis in ((1, 0) if vertical_first else (0, 1)):
        if axis == 0 and out_w != w:
            cur = resample_pass(cur, 0, out_w, filt)
        elif axis == 1 and out_h != h:
            cur = resample_pass(cur, 1, out_h, filt)
    return cur


def preprocess_pair(image: np.ndarray, cond: np.ndarray, image_size=256, mid_res=1.125, crop=None, flip=False):
    h, w, _ = image.shape
    if cond.shape[:2] != (h, w):
        cond = resize(cond, h, w, 'bicubic')
    nh, nw = resized_size(h, w, round(mid_res * image_size))
    top, left = crop if crop is not None else center_crop_offsets(nh, nw, image_size, image_size)
    outs = []
    for src in (image, cond):
        r = resize(src, nh, nw, 'lanczos') if (nh, nw) != (h, w) else src
        win = r[top:top + image_size, left:left + image_size]
        if flip:
            win = win[:, ::-1]
        t = (win.astype(np.float32) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
        outs.append(np.ascontiguousarray(t.transpose(2, 0, 1)))
    return outs[0], outs[1]


def ignore_masks(cond: np.ndarray, patch_nums, first_masked_scale=5, separator=False):
    """cond (3, H, W) float32 -> (ignore_mask, ignore_mask_) of length sum 2 pn^2 (imagenetC.py:152-178; torch 'nearest' index rule);
    separator: one kept token in front of every half-scale but the first pair (:158,:169-170).  Pinned by tests/golden/preprocess.npz,
    which make_golden.py records from the reference's own statements (lifted out of the file by ast)."""
    _, H, W = cond.shape
    bg = (cond[0] + cond[1] + cond[2]) == np.float32(-3.0)
    keep = np.where(bg, 0.0, 1.0).astype(np.float32)
    a, b = [], []
    for si, pn in enumerate(patch_nums):
        ones = np.ones(pn * pn + (1 if (separator and si) else 0), np.float32)
        if si < first_masked_scale:
            a += [ones, ones]; b += [ones, ones]
        else:
            sy = np.minimum(np.floor(np.arange(pn, dtype=np.float32) * (np.float32(H) / np.float32(pn))).astype(np.int64), H - 1)
            sx = np.minimum(np.floor(np.arange(pn, dtype=np.float32) * (np.float32(W) / np.float32(pn))).astype(np.int64), W - 1)
            m = keep[sy][:, sx].reshape(-1)
            if separator:
                m = np.concatenate([np.ones(1, np.float32), m])
            a += [m, ones]; b += [ones, m]
    return np.concatenate(a), np.concatenate(b)


def process_anns(anns, image_size, colormap):
    """datasets/imagenetC.py:15-29 with UNCOMPRESSED run lengths (or decoded masks) expanded explicitly, column-major, numpy only.
    Pinned by tests/golden/preprocess.npz (the reference's own process_anns text run on the decoded masks).  pycocotools' compressed
    string codec is outside this oracle (absent dependency, parity unpinned)."""
    mask = np.zeros((image_size, image_size, 3))
    for ann in anns:
        if ann['area'] < 5000:
            continue
        seg = ann['segmentation']
        if isinstance(seg, np.ndarray):
            m = (seg != 0).astype(np.uint8)
        else:
            h, w = seg['size']
            flat = np.zeros(h * w, np.uint8)
            pos, val = 0, 0
            for r in list(seg['counts']):
                flat[pos:pos + r] = val
                pos += r
                val ^= 1
            m = flat.reshape(w, h).T                       # column-major
        X, Y = m.shape[1], m.shape[0]
        index = np.where(m == 1)
        x = int(np.mean(index[1]) // (X / 11))
        y = int(np.mean(index[0]) // (Y / 11))
        assert x * y < 124
        mask[m.astype(bool)] = colormap[(x * y) % len(colormap)]
    return mask
