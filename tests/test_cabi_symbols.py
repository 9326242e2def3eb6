"""CPU: the C-ABI library loads and exports every symbol include/nc_hip.h declares (no compute calls)."""
import ctypes
import os

import pytest

from neuroclear_amd import _lib


def test_header_lists_symbols():
    syms = _lib.header_symbols()
    assert len(syms) >= 35
    for must in ('nc_conv_fwd', 'nc_conv_dgrad', 'nc_conv_wgrad', 'nc_instnorm_act_fwd', 'nc_unet_deconv_fwd',
                 'nc_assemble_scatter_add', 'nc_dice_cut_cube', 'nc_adam_step'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in _lib.header_symbols() if not hasattr(L, s)]
    assert not missing, missing
    L.nc_version.restype = ctypes.c_int
    assert L.nc_version() >= 100


def test_no_cpu_fallback():
    import torch
    from neuroclear_amd import ops
    with pytest.raises(_lib.NcError):
        ops.conv(torch.zeros(1, 1, 4, 4, 4), torch.zeros(1, 1, 3, 3, 3), None, 1, 1)
    with pytest.raises(_lib.NcError):
        ops.instance_norm_act(torch.zeros(1, 2, 4, 4, 4))
