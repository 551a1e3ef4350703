"""N3 evaluation loop: host schedules against the reference's formulas (train_control_var_hpu.py:338-408), the PNG writer
against PIL when it is installed, and (GPU) a tiny sharded run end to end."""
import io
import os

import numpy as np
import pytest
import torch

from controlvar_amd import evaluate as E


def test_class_slices_cover_all_classes_like_the_reference():
    for gpus in (1, 2, 3, 7, 8, 16):
        slices = 1000 // gpus
        got = [E.class_slice(r, gpus) for r in range(gpus)]
        for r in range(gpus):                      # literal restatement of :368-370
            want = [i for i in range(slices * r, slices * (r + 1))] if r != gpus - 1 else [i for i in range(slices * r, 1000)]
            assert got[r] == want
        flat = [c for g in got for c in g]
        assert flat == list(range(1000))


def test_sample_batches_and_seed_schedule():
    assert E.sample_batches(16) == [(0, 16), (1, 16), (2, 16), (3, 2)]
    assert E.sample_batches(25) == [(0, 25), (1, 25)]                      # tail batch of size 0 is skipped (:377)
    assert E.sample_batches(49) == [(0, 49), (1, 1)]
    with pytest.raises(AssertionError):
        E.sample_batches(50)
    # the reference mutates `seed` in place: replay its loop literally
    seed, want = 42, []
    for cls in (3, 4):
        for i in range(50 // 16 + 1):
            B = 16 if i != 50 // 16 else 50 - i * 16
            if B == 0:
                continue
            seed = seed + i * (cls + 1)
            want.append((cls, i, B, seed))
    assert list(E.seed_schedule(42, (3, 4), 16)) == want


def test_png_encoder_round_trips():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    data = E.encode_png(img)
    assert data[:8] == b'\x89PNG\r\n\x1a\n'
    Image = pytest.importorskip('PIL.Image')
    back = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
    np.testing.assert_array_equal(back, img)
    with pytest.raises(ValueError):
        E.encode_png(np.zeros((4, 4), np.uint8))


def test_uint8_conversion_truncates_like_the_reference():
    x = torch.tensor([0.0, 0.999, 1.0, 0.5, 0.00392]).view(1, 1, 1, 5).expand(1, 3, 1, 5).contiguous()
    want = x.permute(0, 2, 3, 1).clone().mul_(255).cpu().numpy().astype(np.uint8)
    np.testing.assert_array_equal(E.to_uint8_hwc(x), want)
    assert want[0, 0, 1, 0] == 254 and want[0, 0, 2, 0] == 255


@pytest.mark.gpu
def test_validate_classes_end_to_end_and_gibbs(gpu_device, tmp_path):
    from controlvar_amd import models
    vae = models.build_vae(ch=32, compute_dtype=torch.bfloat16).to(gpu_device)
    var = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True, compute_dtype=torch.bfloat16).to(gpu_device)
    var.train()
    kw = dict(batch_size=2, per_class=3, guidance_scale=(4, 4, 4), top_k=900, top_p=0.96, seed=7)
    r0 = E.validate_classes(var, vae, str(tmp_path / 'a'), rank=0, gpus=1, classes=[5, 6], **kw)
    assert var.training                                     # the reference restores train mode (:408)
    assert r0['images'] == 6 and len(r0['files']) == 6
    assert sorted(os.listdir(tmp_path / 'a' / 'cfg_4' / '5')) == ['0.png', '1.png', '2.png']
    Image = pytest.importorskip('PIL.Image')
    im = Image.open(r0['files'][0])
    assert im.size == (256, 256) and im.mode == 'RGB'
    # deterministic: a second run writes identical bytes
    r1 = E.validate_classes(var, vae, str(tmp_path / 'b'), rank=0, gpus=1, classes=[5, 6], **kw)
    for fa, fb in zip(r0['files'], r1['files']):
        assert open(fa, 'rb').read() == open(fb, 'rb').read()
    # Gibbs alternation: runs both teacher-forced passes and keeps the (B, 3, 512, 256) pair shape
    r2 = E.validate_classes(var, vae, str(tmp_path / 'c'), classes=[5], gibbs=1, save_val=False, **kw)
    assert r2['last'].shape == (1, 3, 512, 256) and r2['files'] == []
    assert float(r2['last'].min()) >= 0.0 and float(r2['last'].max()) <= 1.0
    # pixel-conditional branch on a synthetic "dataloader"
    from controlvar_amd.synth import synth_images
    batch = {'image': synth_images(2, 256, seed=1), 'mask': synth_images(2, 256, seed=2), 'cls': torch.tensor([1, 2]), 'type': torch.tensor([0, 2])}
    r3 = E.validate_dataloader(var, vae, [batch], str(tmp_path / 'd'), 'depth', c_mask=True, c_img=False, rank=3, guidance_scale=(4, 4, 4),
                               top_k=900, top_p=0.96, seed=7)
    assert r3['images'] == 2 and os.path.dirname(r3['files'][0]).endswith(os.path.join('cfg_4_4_4_depth', '3'))
