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


@pytest.mark.parametrize('name,kernel', [('conv_s3x', 'k_conv_s3x'), ('conv_s3x', 'k_conv_s3w'), ('conv_p2d', 'k_conv_p2d')])
def test_hand_counted_loads_are_not_touched_in_flight(tmp_path, name, kernel):
    """conv_s3x.hip / conv_p2d.hip issue their weight / bias loads as inline assembly and wait for them with hand-placed s_waitcnt (the
    compiler's own vmcnt bookkeeping would wait for freshly issued LDS-DMA in front of every k-step).  The compiler does not know those
    registers are in flight: this compiles the kernel as the Makefile does, keeps the ISA, and checks (tools/check_asm_loads.py: a text-order
    model of the in-order vector-memory queue) that no instruction reads or overwrites a destination register while its load can still be
    outstanding, and that nothing in the kernels is a function call (a lambda left out of line spills live -- possibly in-flight -- registers
    around the call), and that nothing spills.  Round 5: the epilogue-statistics variants of k_conv_s3x failed exactly this way (accumulators
    zeroed on top of the last step's dummy request) until the final wait named the request's registers."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, 'neuroclear_amd', 'csrc', name + '.hip')
    obj = str(tmp_path / (name + '.o'))
    subprocess.run([hipcc, '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Wno-unused-function', '-Wno-int-to-pointer-cast',
                    '-save-temps=obj', '-c', src, '-o', obj], check=True, capture_output=True, cwd=os.path.dirname(src))
    asm = str(tmp_path / (name + '-hip-amdgcn-amd-amdhsa-gfx950.s'))
    assert os.path.exists(asm)
    sys.path.insert(0, os.path.join(root, 'tools'))
    import check_asm_loads
    assert check_asm_loads.main(asm, kernel) == 0
    text = open(asm).read()
    assert 's_swappc_b64' not in text and '.vgpr_spill_count: 0' in text
    for line in text.splitlines():
        if '.vgpr_spill_count:' in line:
            assert line.strip().endswith(' 0'), line
    if kernel == 'k_conv_s3w':
        # its one counted wait: at k-step 1 behind an epilogue, vmcnt(63) stands for "the DMA requested at k-step 0 has landed" because the
        # tile's 64 (+ 1 record) stores were issued AFTER those requests.  In the ISA: walking back from the first vmcnt(63) of each
        # instantiation, at least 64 store instructions come before the youngest LDS-DMA request (a scheduler that moved the requests
        # behind the stores would silently break the count).
        body, kern = {}, None
        for line in text.splitlines():
            code = line.split(';')[0].strip()
            if code.endswith(':') and not code.startswith('.'):
                kern = code[:-1]
            if kern and 'k_conv_s3w' in kern and code:
                body.setdefault(kern, []).append(code)
        assert len(body) == 2, list(body)
        for kern, L in body.items():
            sites = [i for i, c in enumerate(L) if c.startswith('s_waitcnt vmcnt(63)')]
            assert sites, kern
            j, n = sites[0] - 1, 0
            while j >= 0 and not (L[j].startswith('buffer_load') and ' lds' in L[j]):
                n += L[j].startswith('buffer_store')
                j -= 1
            assert j >= 0 and n >= 64, (kern, n)


def test_lds_dma_is_not_issued_through_the_builtin():
    """__builtin_amdgcn_global_load_lds makes hipcc book a FLAT access that may touch LDS: every later LDS-read wait in the kernel
    becomes lgkmcnt(0) and LDS reads behind an outstanding request get vmcnt(0).  The kernels issue the same
    instruction through common.hpp's nc_dma_lds16 / nc_dma_lds4 (inline assembly) and wait by hand -- keep it that way."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for f in glob.glob(os.path.join(root, 'neuroclear_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(root, 'neuroclear_amd', 'csrc', '*.hpp')):
        for n, line in enumerate(open(f), 1):
            code = line.split('//')[0]
            if '__builtin_amdgcn_global_load_lds' in code:
                bad.append('%s:%d' % (os.path.basename(f), n))
    assert not bad, bad
