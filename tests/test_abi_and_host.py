"""CPU-side checks: the C-ABI library loads and exports every symbol include/cvar.h declares,
the ctypes mirror of cvar_gemm_desc matches the C layout, host tables agree with the oracle,
and the product path refuses to run without a GPU (no silent fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'cvar.h')


@pytest.fixture(scope='module')
def lib():
    from controlvar_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from controlvar_amd.build import build_lib
        build_lib(verbose=False)
    return _lib.load()


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cvar_[a-z0-9_]+)\s*\(', src)))


def test_every_header_symbol_is_exported_and_bound(lib):
    from controlvar_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/cvar.h but not exported'
        assert s in _lib.SIGNATURES, f'{s} has no ctypes signature'
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.cvar_abi_version() == _lib.ABI_VERSION
    assert lib.cvar_status_str(-2).decode() == 'unsupported shape or dtype'


def test_gemm_desc_layout_matches_c(tmp_path, lib):
    """compile a tiny C program against include/cvar.h and compare sizeof/offsetof with the ctypes mirror"""
    from controlvar_amd._lib import GemmDesc
    fields = [f[0] for f in GemmDesc._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "cvar.h"\nint main(){printf("%zu\\n", sizeof(cvar_gemm_desc));\n'
    for f in fields:
        prog += f'printf("%zu\\n", offsetof(cvar_gemm_desc, {f}));\n'
    prog += 'return 0;}\n'
    src = tmp_path / 'layout.c'
    src.write_text(prog)
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert vals[0] == ctypes.sizeof(GemmDesc)
    for f, off in zip(fields, vals[1:]):
        assert getattr(GemmDesc, f).offset == off, f


def test_argument_validation_without_gpu(lib):
    from controlvar_amd._lib import GemmDesc
    d = GemmDesc()
    assert lib.cvar_gemm(ctypes.byref(d), None) == -1            # null pointers -> CVAR_EINVAL, no launch
    assert lib.cvar_ln_modulate(None, None, None, 0, 1, None, 0, 4, 64, 1e-6, None) == -1
    assert lib.cvar_groupnorm_ws_bytes(2, 65536, 160) == (2 * 128 * 160 * 2 + 2 * 160 * 2) * 4


def test_pyramid_tables_equal_oracle():
    from controlvar_amd.pyramid import area_matrix, bicubic_matrix, packed_tables
    from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN
    from oracle import interp
    for p in PN:
        assert np.array_equal(area_matrix(16, p), interp.area_matrix(16, p))
        assert np.allclose(bicubic_matrix(p, 16), interp.bicubic_matrix(p, 16), atol=1e-15)
    up, down, offs = packed_tables(PN)
    assert up.shape == down.shape == (16 * sum(PN),) and list(offs[:3]) == [0, 16, 48]
    assert np.allclose(up[offs[-1]:].reshape(16, 16), np.eye(16)) and np.allclose(down[offs[-1]:].reshape(16, 16), np.eye(16))


def test_spec_flops_and_pyramid():
    from controlvar_amd.spec import Pyramid, VarConfig, algorithmic_gflop_per_row
    py = Pyramid()
    assert py.l == (2, 8, 18, 32, 50, 72, 128, 200, 338, 512) and py.L == 1360
    assert Pyramid(mask_factor=1).L == 680
    f = algorithmic_gflop_per_row(VarConfig(depth=24), n_ada=10)
    assert abs(f['total'] - 2041.3) < 0.1 and abs(f['attn'] - 168.9) < 0.1


def test_lr_schedule_fixture_readable():
    from conftest import golden
    g = golden('lr_lin0')
    assert g['table'].shape[1] == 5


def test_product_path_requires_gpu_and_library():
    """no CPU fallback: building works on CPU, computing raises"""
    from controlvar_amd import models
    vae = models.build_vae(ch=32)
    m = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True)
    assert set(m.state_dict()) >= {'pos_1LC', 'blocks.1.ada_lin.1.weight', 'cond_embed.weight', 'head_nm.ada_lin.1.bias'}
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            m.autoregressive_infer_cfg(2, torch.tensor([1, 2]), cond_type=torch.tensor([0, 1]))
        with pytest.raises(RuntimeError):
            vae.fhat_to_img(torch.zeros(1, 32, 16, 16))


def test_product_does_not_import_oracle():
    """the shipped package must never route through oracle/ (voids parity claims)"""
    pkg = os.path.join(ROOT, 'controlvar_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), fn


def test_state_dict_roundtrip_with_module_prefix():
    """DDP checkpoints carry a 'module.' prefix (train_control_var_hpu.py:478); keys/shapes are the wire format"""
    from controlvar_amd import models
    from controlvar_amd.spec import VarConfig
    from controlvar_amd.synth import synth_var_state
    vae = models.build_vae(ch=32)
    m = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True)
    sd = synth_var_state(VarConfig(depth=2), seed=5)
    wrapped = {'module.' + k: v for k, v in sd.items()}
    m.load_state_dict({k[len('module.'):]: v for k, v in wrapped.items()}, strict=True)
    assert torch.equal(m.state_dict()['blocks.0.ffn.fc1.weight'], sd['blocks.0.ffn.fc1.weight'])


# ------------------------------------------------------------------------------ variant flags (host side)
@pytest.mark.parametrize('fixture,kw', [('forward_d2v', dict(shared_aln=True, type_pos=True)), ('forward_d2sa', dict(aln=-1, layer_scale=0.1)),
                                        ('forward_d2sa0', dict(aln=-1)), ('forward_d2b', dict(bidirectional=True, type_pos=True)),
                                        ('forward_d2s', dict(separate_decoding=True)), ('forward_d2si', dict(separate_decoding=True, indep=True)),
                                        ('forward_d2p', dict(separator=True)), ('forward_d2psi', dict(separator=True, separate_decoding=True, indep=True))])
def test_variant_models_have_the_reference_state_dict_layout(fixture, kw):
    """the module tree built from spec.var_state_shapes yields the reference's state_dict() key order for every built variant"""
    import numpy as np
    from controlvar_amd import models
    keys = [str(k) for k in np.load(os.path.join(os.path.dirname(__file__), 'golden', fixture + '.npz'))['keys']]
    vae = models.build_vae(ch=32)
    m = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True, **kw)
    assert list(m.state_dict().keys()) == keys


def test_unbuilt_variant_flags_fail_loudly_and_indep_defaults_follow_upstream():
    from controlvar_amd import models
    vae = models.build_vae(ch=32)
    with pytest.raises(NotImplementedError):
        models.build_control_var(vae, depth=2, mask_type='replace', separator=True)
    with pytest.raises(NotImplementedError):
        models.build_control_var(vae, depth=2, mask_type='replace', type_pos=True)
    with pytest.raises(NotImplementedError):
        models.build_control_var(vae, depth=2, mask_type='replace', separate_decoding=True)
    # class default indep=True (control_var.py:31), factory default False (models/__init__.py:28): both accepted, a no-op without separate_decoding
    m = models.ControlVAR(vae, depth=2, embed_dim=128, num_heads=2, mask_factor=2, multi_cond=True)
    assert m.indep is True and not m.separate_decoding
    f = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True)
    assert f.indep is False
    assert torch.equal(m.state_dict()['attn_bias_for_masking'], f.state_dict()['attn_bias_for_masking'])
    s = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True, separate_decoding=True, indep=True)
    assert not torch.equal(s.state_dict()['attn_bias_for_masking'], f.state_dict()['attn_bias_for_masking'])


# ------------------------------------------------------------------------------------------------ round 6: host-side additions
def test_board_sampler_degrades_without_a_device():
    """controlvar_amd.telemetry is a measurement aid: without an AMD SMI device (this container) it must report 'unavailable' with a reason and never raise"""
    from controlvar_amd.telemetry import BoardSampler
    with BoardSampler(0, period_s=0.01) as bs:
        pass
    s = bs.summary()
    assert isinstance(s, dict) and 'available' in s
    if not s['available']:
        assert s.get('reason')


def test_encoder_precision_and_deterministic_plan_keywords():
    """the two constructor keywords this library adds: validated at build time, no GPU needed"""
    import torch
    from controlvar_amd import models
    for ep in ('bf16', 'bf16x3', 'fp32'):
        assert models.build_vae(ch=32, compute_dtype=torch.bfloat16, encoder_precision=ep).encoder_precision == ep
    assert models.build_vae(ch=32, compute_dtype=torch.bfloat16).encoder_precision == 'bf16'
    assert models.build_vae(ch=32, compute_dtype=torch.float32).encoder_precision == 'fp32'
    with pytest.raises(ValueError):
        models.build_vae(ch=32, compute_dtype=torch.bfloat16, encoder_precision='fp8')
    with pytest.raises(ValueError):
        models.build_vae(ch=32, compute_dtype=torch.float32, encoder_precision='bf16x3')      # an fp32 model encodes in fp32
    vae = models.build_vae(ch=32)
    m = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True, deterministic_plan=True)
    assert m.deterministic_plan is True
    assert models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True).deterministic_plan is False
    assert models.build_var(vae, depth=2, deterministic_plan=True).deterministic_plan is True


def test_current_md_and_readme_table_regenerate_from_the_committed_profiles(tmp_path):
    """profiles/CURRENT.md and README's numbers table are generated files: the generators must run on the committed evidence (tools/current_md.py, tools/fill_readme.py)"""
    import subprocess, sys, glob
    lines = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_default_line.json')))
    if not lines:
        pytest.skip('no bench line under profiles/')
    tag = os.path.basename(lines[-1]).split('_')[0]
    out = tmp_path / 'CURRENT.md'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'current_md.py'), tag, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    txt = out.read_text()
    assert 'images/s' in txt and 'frac' in txt and len(txt.splitlines()) <= 80
