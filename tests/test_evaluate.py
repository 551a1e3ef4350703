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


def _decode_png_rgb8(data: bytes) -> np.ndarray:
    """inverse of E.encode_png for the files it writes (8-bit RGB, filter type 0 on every row) - no PIL needed on the GPU box"""
    import struct
    import zlib
    assert data[:8] == b'\x89PNG\r\n\x1a\n'
    o, idat, w, h = 8, b'', None, None
    while o < len(data):
        n, tag = struct.unpack('>I', data[o:o + 4])[0], data[o + 4:o + 8]
        body = data[o + 8:o + 8 + n]
        if tag == b'IHDR':
            w, h, depth, ctype = struct.unpack('>IIBB', body[:10])
            assert (depth, ctype) == (8, 2)
        elif tag == b'IDAT':
            idat += body
        o += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


@pytest.mark.gpu
def test_validate_outputs_match_the_reference_pixels(gpu_device, tmp_path):
    """N3 parity (fp32 mode, greedy so that the sampler's generator does not matter):
    * pixel-conditional branch (validate_dataloader -> pix_cond_inference -> conditional_infer_cfg) against the REFERENCE's recorded
      generations gen_d2_cmask.npz (c_mask, cfg (4,4,4)) and gen_d2_cimg.npz (c_img, cfg (3,2,1)) on the same control images;
    * class-conditional branch (validate_classes -> cls_cond_inference) and one Gibbs round against the pinned oracle run with the
      same class / condition type / seed schedule, compared on the PNG pixels the loop writes (uint8 TRUNCATION of x*255,
      train_control_var_hpu.py:358,399: a float difference below 1e-3 can move a value across an integer, so up to 1 LSB on
      a small fraction of pixels is allowed; any flipped token would change whole 16x16 patches)."""
    from conftest import golden
    from controlvar_amd import models
    from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN, VaeConfig, VarConfig, phi_index_map
    from controlvar_amd.synth import synth_images, synth_vae_state, synth_var_state
    from oracle import var_ref, vqvae_ref
    from oracle.vqvae_ref import MSQuant
    vae = models.build_vae(ch=32, compute_dtype=torch.float32).to(gpu_device)
    var = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True, compute_dtype=torch.float32, cond_drop_rate=0.0).to(gpu_device).eval()
    ctrl = synth_images(2, 256, seed=4)

    def px(files):
        return np.stack([_decode_png_rgb8(open(f, 'rb').read()) for f in files])

    def close_u8(a, b, what):
        d = np.abs(a.astype(np.int16) - b.astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 0.02, f'{what}: max diff {d.max()}, differing fraction {(d > 0).mean():.4f}'

    # ---- pixel-conditional branch vs the reference fixtures
    for name, cfg3, flags in (('gen_d2_cmask', (4.0, 4.0, 4.0), dict(c_mask=True, c_img=False)), ('gen_d2_cimg', (3.0, 2.0, 1.0), dict(c_mask=False, c_img=True))):
        g = golden(name)
        batch = {'image': ctrl, 'mask': ctrl, 'cls': torch.tensor([5, 6]), 'type': torch.tensor([2, 3])}
        r = E.validate_dataloader(var, vae, [batch], str(tmp_path / name), 'depth', rank=0, guidance_scale=cfg3, top_k=1, top_p=0.0, seed=0, **flags)
        got = px(r['files'])                                              # (2, 256, 256, 3): the image half
        assert got.shape == (2, 256, 256, 3)
        ref_crop2 = torch.from_numpy(g['img_crop2'])                      # img[:, :, -20:-4, 200:216] of the (512, 256) pair -> rows 236:252 of the half
        want = ref_crop2.permute(0, 2, 3, 1).mul(255).numpy().astype(np.uint8)
        close_u8(got[:, 236:252, 200:216], want, name)
        ref_mean = torch.from_numpy(g['img_mean'])                        # over the whole pair; check the loop's tensor path too
        r2 = E.validate_dataloader(var, vae, [batch], str(tmp_path / (name + 'x')), 'depth', guidance_scale=cfg3, top_k=1, top_p=0.0, seed=0, save_val=False, **flags)
        assert (r2['last'].mean(dim=(2, 3)).cpu() - ref_mean).abs().max() < 3e-4
    # ---- class branch + one Gibbs round vs the oracle
    cfg = VarConfig(depth=2)
    sdv, sd = synth_vae_state(VaeConfig(ch=32)), synth_var_state(cfg)
    msq = MSQuant(sdv, PN, phi_index_map(10))
    kw = dict(batch_size=2, per_class=3, guidance_scale=(4.0, 4.0, 4.0), top_k=1, top_p=0.0, seed=7)
    r = E.validate_classes(var, vae, str(tmp_path / 'cls'), classes=[5], **kw)
    got = px(r['files'])
    want = []
    with torch.no_grad():
        for cls, i, B, s in E.seed_schedule(7, [5], 2, 3):
            f = var_ref.generate(sd, cfg, msq, B, torch.full((B,), cls), 4.0, top_k=1, cond_type=torch.full((B,), 2))
            want.append(var_ref.decode_fhat(sdv, f)[:, :, 256:])
    want = torch.cat(want).permute(0, 2, 3, 1).mul(255).numpy().astype(np.uint8)
    close_u8(got, want, 'class branch')
    rg = E.validate_classes(var, vae, str(tmp_path / 'gibbs'), classes=[5], gibbs=1, save_val=False, batch_size=2, per_class=3, guidance_scale=(4.0, 4.0, 4.0),
                            top_k=1, top_p=0.0, seed=7)                  # 'last' = the tail batch (i = 1, B = 1)
    with torch.no_grad():
        labels, types = torch.full((1,), 5), torch.full((1,), 2)
        img = var_ref.decode_fhat(sdv, var_ref.generate(sd, cfg, msq, 1, labels, 4.0, top_k=1, cond_type=types))
        for _ in range(2):                                               # c_mask stays set: both passes teacher-force the control tokens
            masks = (img[:, :, :256] - 0.5) / 0.5
            c_ids = vqvae_ref.img_to_idxBl(sdv, msq, masks)
            img = var_ref.decode_fhat(sdv, var_ref.generate(sd, cfg, msq, 1, labels, (4.0, 4.0, 4.0), top_k=1, cond_type=types, four_way=True, c_mask=c_ids))
    close_u8(E.to_uint8_hwc(rg['last']), img.permute(0, 2, 3, 1).mul(255).numpy().astype(np.uint8), 'gibbs round')
