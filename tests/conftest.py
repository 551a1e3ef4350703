import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: CPU test that takes more than a few seconds')


def golden(name):
    """Load a fixture recorded from the reference by tests/golden/make_golden.py."""
    path = os.path.join(GOLDEN, name + '.npz')
    if not os.path.exists(path):
        pytest.skip(f'fixture {name}.npz missing')
    return dict(np.load(path))


@pytest.fixture(scope='session')
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


# ------------------------------------------------------------------------------------------------ integer-parity bookkeeping
PARITY_REPORT = os.environ.get('CVAR_PARITY_REPORT', os.path.join(ROOT, 'gpurun_out', 'parity_report.jsonl'))


def record(what, **vals):
    """append one measured parity record (flip counts, logit distances) to the JSON-lines report tools/parity_report.py summarises into
    profiles/rNN_parity_report.json; never fails a test"""
    import json
    try:
        os.makedirs(os.path.dirname(PARITY_REPORT), exist_ok=True)
        with open(PARITY_REPORT, 'a') as f:
            f.write(json.dumps(dict(what=what, **vals)) + '\n')
    except OSError:
        pass


def ids_parity(got, ref, margin, tol, what, strict=True):
    """Integer parity of ids (VQ ids, greedy tokens, argmax of logits) against a reference-recorded fixture.

    strict=True  (every fp32-mode fixture): the ids must be IDENTICAL - zero flips.
    strict=False (bf16 mode, random-feature sweeps): a flip is tolerated only where the reference's own top-1/top-2 margin is < tol.
    The flip count is printed and recorded either way.  Returns (flips, rows_ok) - rows_ok[b] is True for every batch row without a
    flip, so float / image checks always run on those rows (nothing is skipped behind `if flips == 0`)."""
    got, ref = np.asarray(got).astype(np.int64), np.asarray(ref).astype(np.int64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    mism = got != ref
    n = int(mism.sum())
    worst = float(np.asarray(margin)[mism].max()) if n else 0.0
    rows_ok = ~mism.reshape(mism.shape[0], -1).any(axis=1)
    print(f'[parity] {what}: {n} of {mism.size} ids differ from the reference' + (f' (largest reference margin at a flip {worst:.3e})' if n else ''))
    record(what, kind='ids', flips=n, total=int(mism.size), worst_margin_at_flip=worst, strict=bool(strict), tol=float(tol))
    if strict:
        assert n == 0, f'{what}: {n} of {mism.size} ids differ (strict); largest reference margin at a flip {worst:.3e}'
    elif n:
        assert worst < tol, f'{what}: {n} id mismatches, largest reference margin at a mismatch {worst:.3e} >= {tol:.3e}'
    return n, rows_ok


def maxabs_on(diff, rows_ok, min_rows=1):
    """max |diff| over the batch rows without an id flip.  At least ``min_rows`` rows must qualify (with none the check
    compared nothing and passed); a caller that tolerates an all-flipped batch says so with ``min_rows=0`` and gets 0.0."""
    import torch
    ok = torch.as_tensor(np.asarray(rows_ok), dtype=torch.bool)
    assert int(ok.sum()) >= min_rows, f'float parity check has {int(ok.sum())} flip-free rows of {ok.numel()}, needs {min_rows}'
    if not bool(ok.any()):
        return 0.0
    return float(diff[ok].abs().max())


def rows_ok_per_sample(rows_ok, n_samples):
    """ids rows of a CFG run are laid out branch-major ([branch][sample], control_var.py:270-283: the branches are concatenated along
    the batch dimension): a sample counts as clean only if all of its branches are.  Checks the layout assumption it relies on."""
    ok = np.asarray(rows_ok)
    assert ok.ndim == 1 and ok.size % n_samples == 0, (ok.shape, n_samples)
    return ok.reshape(-1, n_samples).all(axis=0)
