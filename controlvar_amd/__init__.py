"""controlvar_amd - MI355X-native hot path of lxa9867/ControlVAR.

Public surface mirrors the reference's ``models`` package (models/__init__.py:1-45):
``VQVAE``, ``VAR``, ``ControlVAR``, ``build_var``, ``build_control_var`` (+ ``build_vae``).
Heavy imports (torch, the HIP library) happen lazily so that ``import controlvar_amd``
and the pure-host modules (spec, synth) work everywhere.
"""
from . import spec  # noqa: F401

__all__ = ['VQVAE', 'VAR', 'ControlVAR', 'build_var', 'build_control_var', 'build_vae', 'spec', 'register_torch_ops']


def register_torch_ops():
    """Register the C-ABI kernels as ``torch.ops.cvar.*`` (controlvar_amd/torch_ops.py) and return that namespace."""
    import torch
    from . import torch_ops  # noqa: F401
    return torch.ops.cvar


def __getattr__(name):
    if name in ('VQVAE', 'VAR', 'ControlVAR', 'build_var', 'build_control_var', 'build_vae'):
        from . import models as _m
        return getattr(_m, name)
    raise AttributeError(name)
