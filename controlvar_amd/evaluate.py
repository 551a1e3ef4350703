"""Evaluation loop: the consumer right after the hot path.

Restates ``validate`` / ``pix_cond_inference`` / ``cls_cond_inference`` of train_control_var_hpu.py:297-408 over this
package's models: 50 samples for each of the 1000 classes, classes sharded by rank (the last rank takes the remainder),
optional Gibbs alternation between control-teacher-forced and image-teacher-forced decoding, PNG dump.  One process per
GPU; there is no collective on this path (ranks write disjoint directories), so it scales as the generation itself.

Host code only: every tensor operation is a call into the models (which call the HIP library).  PNGs are written by a
small zlib encoder (no PIL / torchvision dependency on the GPU box).

Deviations from the reference, on purpose:
* the class-conditional branch of the reference saves ``images[b, 256]`` - ONE pixel row - instead of ``images[b, 256:]``
  (the image half, as its dataloader branch does, :361); this module saves the image half in both branches;
* ``validate`` logs to wandb when ``save_val`` is false; here ``save_val=False`` returns the tensors to the caller instead.
"""
from __future__ import annotations

import os
import struct
import zlib
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

COND_TYPES = {'mask': 0, 'canny': 1, 'depth': 2, 'normal': 3, 'none': 4}      # train_control_var_hpu.py:298,328


# ----------------------------------------------------------------------------------------------- schedules (pure host)
def class_slice(rank: int, gpus: int, num_classes: int = 1000) -> List[int]:
    """classes of one rank: ``slices = 1000 // gpus``; every rank takes ``slices`` classes, the last one up to 1000 (:368-370)"""
    slices = num_classes // gpus
    return list(range(slices * rank, slices * (rank + 1))) if rank != gpus - 1 else list(range(slices * rank, num_classes))


def sample_batches(batch_size: int, per_class: int = 50) -> List[Tuple[int, int]]:
    """[(i, B)] of the reference's ``for i in range(50 // bs + 1)`` loop with its tail batch and the B == 0 skip (:374-377)"""
    if not per_class > batch_size:
        raise AssertionError('the reference asserts 50 > batch_size')            # :374
    out = []
    for i in range(per_class // batch_size + 1):
        B = batch_size if i != per_class // batch_size else per_class - i * batch_size
        if B:
            out.append((i, B))
    return out


def seed_schedule(seed: int, classes: Sequence[int], batch_size: int, per_class: int = 50) -> Iterator[Tuple[int, int, int, int]]:
    """(cls, i, B, seed) in generation order.  The reference updates ``seed = seed + i * (cls + 1)`` in place (:379), i.e.
    the seed accumulates over batches AND classes of a rank; batches with B == 0 are skipped before the update."""
    for cls in classes:
        for i, B in sample_batches(batch_size, per_class):
            seed = seed + i * (cls + 1)
            yield cls, i, B, seed


# ----------------------------------------------------------------------------------------------- PNG (RGB8, no dependency)
def encode_png(rgb) -> bytes:
    """rgb: uint8 array-like (H, W, 3) -> PNG bytes (filter 0 on every row, zlib level 6)."""
    import numpy as np
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f'expected (H, W, 3) uint8, got {a.shape}')
    h, w = a.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * 3)], axis=1).tobytes()

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)

    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def to_uint8_hwc(images):
    """(B, 3, H, W) in [0, 1] -> (B, H, W, 3) uint8 exactly as ``permute(0,2,3,1).mul_(255)...astype(np.uint8)`` (:358): truncation"""
    return images.permute(0, 2, 3, 1).mul(255).cpu().numpy().astype('uint8')


def save_image_half(images, paths: Sequence[str], image_rows: int = 256) -> None:
    """write rows ``image_rows:`` (the generated image below the generated control) of each sample to ``paths``"""
    arr = to_uint8_hwc(images)
    for b, path in enumerate(paths):
        with open(path, 'wb') as f:
            f.write(encode_png(arr[b, image_rows:]))


# ----------------------------------------------------------------------------------------------- inference wrappers
def _as_type_tensor(cond_type, B: int, device):
    import torch
    if isinstance(cond_type, str):
        return torch.full((B,), COND_TYPES[cond_type], device=device, dtype=torch.long)
    return cond_type.to(device)


def cls_cond_inference(var, cls: int, B: int, cond_type='depth', guidance_scale=(6, 6, 6), top_k=900, top_p=0.96, seed=42):
    """train_control_var_hpu.py:326-336: B samples of one class, joint (control, image) generation with cfg = guidance_scale[0]"""
    import torch
    labels = torch.full((B,), int(cls), device=var.device, dtype=torch.long)
    return var.autoregressive_infer_cfg(B=B, label_B=labels, cond_type=_as_type_tensor(cond_type, B, var.device),
                                        cfg=guidance_scale[0], top_k=top_k, top_p=top_p, g_seed=seed)


def pix_cond_inference(var, vqvae, images, masks, conditions, cond_type, c_mask: bool, c_img: bool,
                       guidance_scale=(6, 6, 6), top_k=900, top_p=0.96, seed=42):
    """train_control_var_hpu.py:297-324: tokenise the control (c_mask) or else the image (c_img) and decode with those tokens
    teacher-forced.  ``guidance_scale`` is handed to ``conditional_infer_cfg`` whole (its 3-way CFG), as the reference does."""
    import torch
    dev = var.device
    B = masks.shape[0]
    images, masks = images.to(dev), masks.to(dev)
    labels = torch.full((B,), int(conditions), device=dev, dtype=torch.long) if isinstance(conditions, int) else conditions.to(dev)
    types = _as_type_tensor(cond_type, B, dev)
    pn = var.patch_nums
    cm = ci = None
    if c_mask:
        cm = vqvae.img_to_idxBl(masks, v_patch_nums=pn)
    elif c_img:
        ci = vqvae.img_to_idxBl(images, v_patch_nums=pn)
    return var.conditional_infer_cfg(B=B, label_B=labels, cfg=guidance_scale, top_k=top_k, top_p=top_p, g_seed=seed,
                                     c_mask=cm, c_img=ci, cond_type=types)


def _split_and_renormalise(images, rows: int = 256):
    """generated pair in [0, 1] -> (control, image) in [-1, 1] (:384-385)"""
    masks, imgs = images[:, :, :rows, :], images[:, :, rows:, :]
    return (masks - 0.5) / 0.5, (imgs - 0.5) / 0.5


def gibbs_refine(var, vqvae, images, cls: int, cond_type, steps: int, guidance_scale, top_k, top_p, seed):
    """:381-394: alternately regenerate the image given the control tokens, then (the reference's flags being sticky: c_mask
    stays truthy once set) again given the control tokens of the new sample."""
    c_mask: bool = False
    c_img: bool = False
    for _ in range(steps):
        masks, imgs = _split_and_renormalise(images)
        c_mask = True
        images = pix_cond_inference(var, vqvae, imgs, masks, cls, cond_type, c_mask, c_img, guidance_scale, top_k, top_p, seed)
        masks, imgs = _split_and_renormalise(images)
        c_img = True          # has no effect while c_mask is truthy (pix_cond_inference tests c_mask first), exactly as upstream
        images = pix_cond_inference(var, vqvae, imgs, masks, cls, cond_type, c_mask, c_img, guidance_scale, top_k, top_p, seed)
    return images


# ----------------------------------------------------------------------------------------------- the two validate() branches
def validate_classes(var, vqvae, project_dir: str, batch_size: int, rank: int = 0, gpus: int = 1, guidance_scale=(6, 6, 6),
                     top_k=900, top_p=0.96, seed=42, gibbs: int = 0, save_val: bool = True, classes: Optional[Sequence[int]] = None,
                     per_class: int = 50, cond_type: str = 'depth', progress: Optional[Callable[[int, int], None]] = None) -> Dict[str, object]:
    """class-conditional branch of validate() (:367-406).  Returns {'images': n written / generated, 'files': [...]} (and the
    last batch tensor under 'last' when save_val is false)."""
    was_training = var.training
    var.eval()
    classes = class_slice(rank, gpus) if classes is None else list(classes)
    files: List[str] = []
    last = None
    n = 0
    try:
        for cls, i, B, s in seed_schedule(seed, classes, batch_size, per_class):
            images = cls_cond_inference(var, cls, B, cond_type, guidance_scale, top_k, top_p, s)
            if gibbs:
                images = gibbs_refine(var, vqvae, images, cls, cond_type, gibbs, guidance_scale, top_k, top_p, s)
            n += B
            if save_val:
                d = os.path.join(project_dir, f'cfg_{guidance_scale[0]}', f'{cls}')
                os.makedirs(d, exist_ok=True)
                paths = [os.path.join(d, f'{i * batch_size + b}.png') for b in range(B)]
                save_image_half(images, paths)
                files += paths
            else:
                last = images
            if progress:
                progress(cls, i)
    finally:
        var.train(was_training)
    return {'images': n, 'files': files, 'last': last}


def validate_dataloader(var, vqvae, dataloader: Iterable[dict], project_dir: str, val_cond: str, c_mask: bool, c_img: bool, rank: int = 0,
                        guidance_scale=(6, 6, 6), top_k=900, top_p=0.96, seed=42, save_val: bool = True) -> Dict[str, object]:
    """pixel-conditional branch of validate() (:345-366): batches are dicts with 'image', 'mask', 'cls', 'type'"""
    if not (c_mask or c_img):
        raise ValueError('the dataloader branch needs c_mask or c_img (otherwise validate() takes the class branch)')
    was_training = var.training
    var.eval()
    save_path = os.path.join(project_dir, f'cfg_{guidance_scale[0]}_{guidance_scale[1]}_{guidance_scale[2]}_{val_cond}', f'{rank}')
    if save_val:
        os.makedirs(save_path, exist_ok=True)
    files: List[str] = []
    last = None
    n = 0
    try:
        for batch_idx, batch in enumerate(dataloader):
            images, masks, conditions, cond_type = batch['image'], batch['mask'], batch['cls'], batch['type']
            B = masks.shape[0]
            out = pix_cond_inference(var, vqvae, images, masks, conditions, cond_type, c_mask, c_img, guidance_scale, top_k, top_p, seed)
            n += B
            if save_val:
                paths = [os.path.join(save_path, f'{batch_idx * B + b}.png') for b in range(B)]
                save_image_half(out, paths)
                files += paths
            else:
                last = out
    finally:
        var.train(was_training)
    return {'images': n, 'files': files, 'last': last}
