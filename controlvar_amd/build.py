"""Build libcvar_hip.so (gfx950) in-tree with hipcc.  ``python -m controlvar_amd.build``"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', 'build')
LIB = os.path.join(HERE, 'libcvar_hip.so')
SOURCES = ['gemm.hip', 'gemm_conv.hip', 'gemm_f32.hip', 'gemm_skinny.hip', 'conv_halo.hip', 'conv_c8.hip', 'gemm_tn.hip', 'ops.hip', 'attn.hip', 'sample.hip', 'msq.hip', 'train.hip', 'preproc.hip', 'probe.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result']
# per-source extras.  attn.hip: keep MFMA results in VGPRs - the softmax consumes every score with vector ALU ops, and the AGPR form
# costs one v_accvgpr_read per score and tile (plus writes for the rescale)
EXTRA = {'attn.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (need ROCm 7.x with gfx950 support)')


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    h.update(repr(sorted(EXTRA.items())).encode())
    return h.hexdigest()


def _compile(src: str) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + '.o')
    cmd = [_hipcc(), *FLAGS, *EXTRA.get(src, []), '-c', os.path.join(CSRC, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    return obj


def build_lib(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'cvar_common.h'), os.path.join(CSRC, 'gemm_params.h'),
                                                         os.path.join(os.path.dirname(HERE), 'include', 'cvar.h')]
    stamp = os.path.join(OBJ, 'digest.txt')
    dig = _digest(deps)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        if verbose:
            print(f'[controlvar_amd.build] {LIB} up to date')
        return LIB
    with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    with open(stamp, 'w') as f:
        f.write(dig)
    if verbose:
        print(f'[controlvar_amd.build] built {LIB}')
    return LIB


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
