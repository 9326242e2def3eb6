"""ctypes binding of libnc_hip.so (include/nc_hip.h).  The product path has NO fallback: if the library is missing, or
a tensor is not a dense fp32 CUDA(HIP) tensor, the call raises."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# NC_HIP_LIB: load another build of the same library (kernel timing experiments, tools/ablate_*.py)
LIB_PATH = os.environ.get('NC_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libnc_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'nc_hip.h')

_lib = None


class NcError(RuntimeError):
    pass


def header_symbols():
    """Every function name declared in include/nc_hip.h (used by the CPU symbol-export test)."""
    src = open(HEADER_PATH).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nc_[a-zA-Z0-9_]+)\s*\(', src)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NcError('libnc_hip.so is not built (%s): run `python -c "import __graft_entry__ as g; g.build()"` '
                          'or `make -C neuroclear_amd/csrc`; there is no CPU fallback' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name in header_symbols():
            fn = getattr(L, name)
            if name.endswith('_bytes') or name.endswith('_floats'):
                fn.restype = ctypes.c_size_t
            elif name == 'nc_last_error':
                fn.restype = ctypes.c_char_p
            elif name in ('nc_set_force_direct', 'nc_prof_begin', 'nc_sconv_set_cfg', 'nc_sconv_set_tune', 'nc_set_conv_split', 'nc_set_s3_fusion', 'nc_set_c8x_mode', 'nc_set_split_terms', 'nc_set_h2_guard', 'nc_set_epi_stats', 'nc_set_dl_collapse', 'nc_set_p2d_terms', 'nc_set_s3x_w64'):
                fn.restype = None
            else:
                fn.restype = ctypes.c_int
        _lib = L
    return _lib


def check(code, what=''):
    if code != 0:
        raise NcError('%s failed (%d): %s' % (what, code, lib().nc_last_error().decode()))


P = ctypes.c_void_p
I = ctypes.c_int
L_ = ctypes.c_long
F = ctypes.c_float
Z = ctypes.c_size_t
