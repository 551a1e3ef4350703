"""N2 input pipeline: Pillow's resampler / torchvision's transform chain / the ignore-mask rule.
CPU: host tables + numpy oracle against the Pillow-recorded fixture (and the installed Pillow, when there is one).
GPU: the HIP kernels against the oracle, bit for bit."""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import golden
from controlvar_amd import preprocess as P
from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN
from controlvar_amd.synth import PREPROC_CASES, synth_photo_pair
from oracle import resample_ref as R


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_torchvision_size_rules():
    assert P.resized_size(375, 500, 288) == (288, 384)          # shorter side -> 288, longer int(288 * 500 / 375)
    assert P.resized_size(500, 333, 288) == (432, 288)          # int(288 * 500 / 333) = 432
    assert P.resized_size(288, 300, 288) == (288, 300)          # already at size: returned unchanged
    assert P.resized_size(1024, 683, 288) == (431, 288)
    assert P.center_crop_offsets(288, 384, 256, 256) == (16, 64)
    assert P.center_crop_offsets(289, 385, 256, 256) == (16, 64)    # round(16.5) = 16, round(64.5) = 64: half to even
    assert P.center_crop_offsets(291, 387, 256, 256) == (18, 66)    # round(17.5) = 18


def test_tables_are_pillow_shaped():
    b, k, ks = P.resample_tables(500, 288, 'lanczos')
    assert ks == int(np.ceil(3.0 * 500 / 288)) * 2 + 1 and k.shape == (288, ks) and b.shape == (288, 2)
    assert (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= 500).all()
    assert np.abs(k.sum(1) - (1 << 22)).max() <= ks                 # rows sum to 1.0 up to per-tap rounding
    b2, k2, ks2 = P.resample_tables(100, 288, 'bicubic')            # upscaling: filterscale clamps at 1
    assert ks2 == 5


@pytest.mark.parametrize('h,w,seed', PREPROC_CASES)
def test_oracle_resize_equals_pillow_fixture(h, w, seed):
    g = golden('preprocess')
    img, cond = synth_photo_pair(h, w, seed)
    nh, nw = P.resized_size(h, w, 288)
    big = R.resize(img, nh, nw, 'lanczos') if (nh, nw) != (h, w) else img
    c1 = R.resize(cond, h, w, 'bicubic')
    c2 = R.resize(c1, nh, nw, 'lanczos') if (nh, nw) != (h, w) else c1
    tag = f'{h}x{w}'
    for name, arr in (('img288', big), ('cond_fit', c1), ('cond288', c2)):
        assert tuple(g[f'{tag}_{name}_shape']) == arr.shape
        np.testing.assert_array_equal(arr[::37, ::29], g[f'{tag}_{name}_sample'])
        assert sha(arr) == str(g[f'{tag}_{name}_sha']), (tag, name)


def test_oracle_resize_equals_installed_pillow():
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(5)
    # the last four are very tall sources: Pillow 12.2 runs the vertical pass first when h > 100 w and the height shrinks
    for (h, w, oh, ow, f) in [(97, 61, 288, 181, 'lanczos'), (301, 777, 288, 743, 'lanczos'), (40, 40, 333, 129, 'bicubic'), (640, 480, 123, 77, 'bicubic'),
                              (701, 7, 255, 64, 'bicubic'), (700, 7, 255, 64, 'bicubic'), (500, 2, 255, 288, 'lanczos'), (512, 4, 700, 100, 'lanczos')]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.LANCZOS if f == 'lanczos' else Image.BICUBIC))
        np.testing.assert_array_equal(R.resize(img, oh, ow, f), want)


def test_pass_order_rule():
    assert not P.vertical_pass_first(700, 7, 255, 64) and P.vertical_pass_first(701, 7, 255, 64)
    assert not P.vertical_pass_first(701, 7, 702, 64) and not P.vertical_pass_first(701, 7, 255, 7) and not P.vertical_pass_first(480, 640, 288, 384)


@pytest.mark.gpu
def test_device_resampler_randomised_sweep_against_pillow(gpu_device):
    """120 random (size, channels, filter, content) cases incl. 1-pixel axes and very tall sources: bit-identical to Image.resize"""
    pytest.importorskip('PIL.Image')
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_resample.py'), '120', '9'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '120/120 cases ok' in r.stdout


def test_oracle_ignore_masks_equal_reference_tensor_code():
    g = golden('preprocess')
    a, b = R.ignore_masks(np.asarray(g['ign_cond']), PN)
    np.testing.assert_array_equal(a, g['ignore_mask'])
    np.testing.assert_array_equal(b, g['ignore_mask_'])
    assert a.shape == (1360,) and 0 < a.mean() < 1
    a, b = R.ignore_masks(np.asarray(g['ign_cond']), PN, separator=True)
    np.testing.assert_array_equal(a, g['ignore_mask_sep'])
    np.testing.assert_array_equal(b, g['ignore_mask__sep'])


@pytest.mark.gpu
@pytest.mark.parametrize('h,w,seed', PREPROC_CASES[:4])
def test_device_pipeline_is_bit_identical(gpu_device, h, w, seed):
    img, cond = synth_photo_pair(h, w, seed)
    di, dc = torch.from_numpy(img).to(gpu_device), torch.from_numpy(cond).to(gpu_device)
    nh, nw = P.resized_size(h, w, 288)
    if (nh, nw) != (h, w):
        got = P.resize_u8(di, nh, nw, 'lanczos').cpu().numpy()
        np.testing.assert_array_equal(got, R.resize(img, nh, nw, 'lanczos'))
        assert sha(got) == str(golden('preprocess')[f'{h}x{w}_img288_sha'])          # == Pillow's own output
    if min(nh, nw) < 256:
        return                                                                            # (the 120x90 case only exercises the resize)
    for crop, flip in ((None, False), ((7, 3), True)):
        gi, gc = P.preprocess_pair(di, dc, crop=crop, flip=flip)
        wi, wc = R.preprocess_pair(img, cond, crop=crop, flip=flip)
        assert gi.shape == (3, 256, 256) and gi.dtype == torch.float32
        np.testing.assert_array_equal(gi.cpu().numpy(), wi)
        np.testing.assert_array_equal(gc.cpu().numpy(), wc)


@pytest.mark.gpu
def test_device_ignore_masks(gpu_device):
    g = golden('preprocess')
    cond = torch.from_numpy(np.asarray(g['ign_cond'])).to(gpu_device)
    batch = torch.stack([cond, -torch.ones_like(cond), torch.zeros_like(cond)])          # real mask, all background, no background
    out = P.ignore_masks(batch, PN)
    np.testing.assert_array_equal(out['ignore_mask'][0].cpu().numpy(), g['ignore_mask'])
    np.testing.assert_array_equal(out['ignore_mask_'][0].cpu().numpy(), g['ignore_mask_'])
    a1, _ = R.ignore_masks(-np.ones((3, 256, 256), np.float32), PN)
    np.testing.assert_array_equal(out['ignore_mask'][1].cpu().numpy(), a1)
    sep = P.ignore_masks(batch, PN, separator=True)                              # imagenetC.py:158,169-170
    np.testing.assert_array_equal(sep['ignore_mask'][0].cpu().numpy(), g['ignore_mask_sep'])
    np.testing.assert_array_equal(sep['ignore_mask_'][0].cpu().numpy(), g['ignore_mask__sep'])
    assert float(out['ignore_mask'][2].min()) == 1.0
    from controlvar_amd import ops
    with pytest.raises(Exception):
        ops.ignore_mask(batch, 3, 256, 256, PN, 5, 0, torch.empty(3, 100, device=gpu_device), 100)      # L mismatch -> CVAR_EINVAL


# ---------------------------------------------------------------- segmentation condition: RLE -> colour map (imagenetC.py:15-37)
from controlvar_amd.synth import synth_annotations as _random_anns          # noqa: E402  (the generator make_golden.py feeds to the reference's text)


def _as_masks(anns):
    return [{'area': a['area'], 'segmentation': a['_mask']} for a in anns]


@pytest.mark.parametrize('seed', [2, 3, 4])
def test_oracle_process_anns_equals_the_references_own_text(seed):
    """fixture = datasets/imagenetC.py:15-37 lifted out of the reference file by ast and run on the decoded masks (make_golden.case_preprocess)"""
    g = golden('preprocess')
    assert np.array_equal(P.create_color_map(), g['colormap'])
    anns = _random_anns(seed, n=8)
    assert sum(a['area'] >= 5000 for a in anns) == int(g[f'anns{seed}_kept']) >= 2
    for form in (_as_masks(anns), anns):                                     # decoded masks (what the reference saw) and run lists
        got = R.process_anns(form, 512, P.create_color_map())
        assert got.dtype == np.float64
        np.testing.assert_array_equal(got.astype(np.uint8), g[f'anns{seed}_canvas'])


def test_uncompressed_rle_semantics_against_an_independent_implementation():
    """runs_from_mask / the run expansion of the oracle against transformers' SAM post-processing (_mask_to_rle / _rle_to_mask:
    "the format expected by pycoco tools", written independently of this repo): same counts for the same mask, same mask back."""
    # the module itself imports torchvision (absent in this image): take the two pure-torch functions out of the installed file by AST
    import ast
    import importlib.util
    import types
    from typing import Any
    spec = importlib.util.find_spec('transformers')
    if spec is None:
        pytest.skip('transformers not installed')
    path = os.path.join(os.path.dirname(spec.origin), 'models', 'sam', 'image_processing_sam.py')
    tree = ast.parse(open(path).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('_mask_to_rle', '_rle_to_mask')]
    assert len(fns) == 2, 'transformers SAM post-processing helpers not found'
    ns = {'torch': torch, 'Any': Any}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, 'exec'), ns)
    sam = types.SimpleNamespace(_mask_to_rle=ns['_mask_to_rle'], _rle_to_mask=ns['_rle_to_mask'])
    anns = _random_anns(1)
    masks = torch.from_numpy(np.stack([a['_mask'] for a in anns]).astype(bool))
    theirs = sam._mask_to_rle(masks)
    for a, t_ in zip(anns, theirs):
        assert t_['size'] == [512, 512] and [int(c) for c in t_['counts']] == a['_runs'] == P.runs_from_mask(a['_mask'])
        back = sam._rle_to_mask({'size': [512, 512], 'counts': a['_runs']}).numpy()
        assert np.array_equal(back, a['_mask'].astype(bool))
    full, empty = np.ones((4, 3), np.uint8), np.zeros((4, 3), np.uint8)
    assert P.runs_from_mask(full) == [0, 12] and P.runs_from_mask(empty) == [12]
    assert P.create_color_map().shape == (124, 3) and tuple(P.create_color_map()[0]) == (0, 0, 64)


def test_compressed_rle_strings_decode_with_a_warning_and_match_hand_derived_answers():
    """The reference's annotation files hold pycocotools' compressed strings (imagenetC.py:20-21).  The product decodes them - through
    pycocotools when installed, else through its own restatement with ONE RuntimeWarning (unpinned third-party format).  Known answers
    worked out BY HAND from the published format (6-bit groups + 48, low 5 bits payload, 0x20 = more, sign = bit 0x10 of the last group,
    deltas against the value two back from the fourth value on):
      [0, 4]          -> '0' '4'
      [5, 3, 40, 2]   -> '5' '3', then 40 as itself (third value, no delta): 40 = 0b01000 + 32 * 1 -> groups 8|0x20 = 40 -> 'X', then 1 -> '1';
                         then 2 - 3 = -1 -> one group 0b11111 (sign bit set, rest all ones) = 31 -> 'O'
      [0, 0, 0, 35]   -> '0' '0' '0', then 35 - 0 = 35 = 3 + 32 * 1 -> 3|0x20 = 35 -> 'S', then 1 -> '1'"""
    import importlib.util
    import warnings
    for runs, text in (([0, 4], '04'), ([5, 3, 40, 2], '53X1O'), ([0, 0, 0, 35], '000S1')):
        assert P.rle_to_string(runs) == text and P.rle_from_string(text) == runs
    rng = np.random.default_rng(0)
    for _ in range(20):
        runs = rng.integers(0, 70000, rng.integers(1, 60)).tolist()
        assert P.rle_from_string(P.rle_to_string(runs)) == runs       # deltas (negative values) and multi-character groups
    spec = importlib.util.spec_from_file_location('coco_rle_string', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'coco_rle_string.py'))
    C = importlib.util.module_from_spec(spec); spec.loader.exec_module(C)
    assert C.rle_from_string is P.rle_from_string                      # the converter is the product's codec, not a second copy
    anns = _random_anns(3, n=8)
    packed = [{'area': a['area'], 'segmentation': {'size': [512, 512], 'counts': P.rle_to_string(a['_runs'])}} for a in anns]
    want = P.annotation_colours(anns, 512)
    have_coco = importlib.util.find_spec('pycocotools') is not None
    P._WARNED_RLE_STRING = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        got = P.annotation_colours(packed, 512)
        P.annotation_colours(packed, 512)
    assert len([x for x in w if 'unpinned third-party format' in str(x.message)]) == (0 if have_coco else 1)
    for a_, b_ in zip(got, want):
        assert np.array_equal(a_, b_)
    with pytest.raises(ValueError):
        P.annotation_colours([{'area': 9999, 'segmentation': {'size': [512, 512], 'counts': P.rle_to_string([512 * 512 - 1])}}], 512)


def test_annotation_colours_follow_the_centroid_rule():
    anns = _random_anns(2)
    want = R.process_anns(anns, 512, P.create_color_map())
    run_ends, offsets, colours = P.annotation_colours(anns, 512)
    kept = [a for a in anns if a['area'] >= 5000]
    assert len(offsets) - 1 == len(kept) == len(colours) and len(kept) >= 2
    for a, col in zip(kept, colours):                                  # the last-painted colour survives somewhere unless fully covered
        assert tuple(col) in {tuple(c) for c in P.create_color_map()}
    assert want.shape == (512, 512, 3) and want.max() > 0
    with pytest.raises(ValueError):
        P.annotation_colours([{'area': 9999, 'segmentation': {'size': [256, 256], 'counts': [256 * 256]}}], 512)
    # decoded masks (what pycocotools.mask.decode hands the reference, imagenetC.py:21) give the same runs / colours as the run lists
    as_masks = _as_masks(anns)
    e2, o2, c2 = P.annotation_colours(as_masks, 512)
    assert np.array_equal(e2, run_ends) and np.array_equal(o2, offsets) and np.array_equal(c2, colours)
    assert np.array_equal(R.process_anns(as_masks, 512, P.create_color_map()), want)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [2, 3, 4])
def test_device_mask_painting_equals_process_anns(gpu_device, seed):
    anns = _random_anns(seed, n=8)
    want = golden('preprocess')[f'anns{seed}_canvas']                       # the reference's own process_anns text (make_golden.case_preprocess)
    assert np.array_equal(R.process_anns(anns, 512, P.create_color_map()).astype(np.uint8), want)
    got = P.paint_annotations(anns, 512, gpu_device).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(P.paint_annotations(_as_masks(anns), 512, gpu_device).cpu().numpy(), want)
    packed = [{'area': a['area'], 'segmentation': {'size': [512, 512], 'counts': P.rle_to_string(a['_runs'])}} for a in anns]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        np.testing.assert_array_equal(P.paint_annotations(packed, 512, gpu_device).cpu().numpy(), want)
    empty = P.paint_annotations([a for a in anns if a['area'] < 5000], 512, gpu_device)
    assert int(empty.max()) == 0


def test_rle_string_decoder_rejects_malformed_input():
    """the compressed-RLE decoder is on the product path for annotation files - a truncated string (continuation bit set on the last character) and characters
    outside the 6-bit alphabet ('0' .. 'o') must raise ValueError, not IndexError / silent garbage"""
    from controlvar_amd.preprocess import rle_from_string, rle_to_string
    good = rle_to_string([5, 3, 40, 2, 1000, 7])
    assert rle_from_string(good) == [5, 3, 40, 2, 1000, 7]
    trunc = rle_to_string([5, 3, 1000])[:-1]                      # 1000 needs two characters: cut inside the run
    with pytest.raises(ValueError, match='malformed COCO RLE'):
        rle_from_string(trunc)
    for bad in ('5#3', 'ab\x7f', '0 1', 'p'):                      # '#', DEL, ' ' below / above the alphabet; 'p' = 112 is one past 'o'
        with pytest.raises(ValueError, match='malformed COCO RLE'):
            rle_from_string(bad)
