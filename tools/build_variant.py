#!/usr/bin/env python3
"""Build an A/B variant of libcvar_hip.so: ONE source recompiled with extra -D flags, linked against the up-to-date objects of the
regular build.  `python tools/build_variant.py attn.hip pksum -DCVAR_ATTN_PKSUM=1` -> ab/libcvar_pksum.so (run with CVAR_LIB=ab/libcvar_pksum.so)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from controlvar_amd import build as B
src, tag, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_lib(verbose=False)
os.makedirs(os.path.join(ROOT, 'ab'), exist_ok=True)
obj = os.path.join(ROOT, 'ab', f'{os.path.splitext(src)[0]}_{tag}.o')
subprocess.check_call([B._hipcc(), *B.FLAGS, *B.EXTRA.get(src, []), *extra, '-c', os.path.join(B.CSRC, src), '-o', obj])
objs = [obj if s == src else os.path.join(B.OBJ, os.path.splitext(s)[0] + '.o') for s in B.SOURCES]
lib = os.path.join(ROOT, 'ab', f'libcvar_{tag}.so')
subprocess.check_call([B._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', lib])
print(lib)
