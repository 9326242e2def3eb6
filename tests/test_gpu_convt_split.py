"""ConvTranspose3d(kernel 2, stride 2) forward on the split-operand arithmetic (csrc/convt_s3.hip; the two nn.ConvTranspose3d of
unet_deconv, models/networks.py:471-478): against fp64, against the fp32 kernels, fp32 and S3 outputs of one call describe the same values."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _call(x, w, b, want_y=True, want_s3=False, xs=None, ctot=None, c0=0):
    from neuroclear_amd import ops
    from neuroclear_amd._lib import I, Z, check, lib
    N, C, D, H, W = x.shape
    K = w.shape[1]
    assert lib().nc_convT_k2s2_split_supported(I(N), I(C), I(D), I(H), I(W), I(K))
    ws = torch.empty(lib().nc_convT_k2s2_split_ws_bytes(I(N), I(C), I(D), I(H), I(W), I(K)), dtype=torch.uint8, device=x.device)
    y = torch.empty(N, K, 2 * D, 2 * H, 2 * W, device=x.device) if want_y else None
    ctot = ctot or K
    ys = torch.zeros(lib().nc_s3_bytes(I(N), I(ctot), ctypes.c_long(8 * D * H * W)), dtype=torch.uint8, device=x.device) if want_s3 else None
    check(lib().nc_convT_k2s2_fwd_split(ops._ptr(x), ops._ptr(xs), ops._ptr(w), ops._ptr(b), ops._ptr(y), ops._ptr(ys), I(ctot), I(c0), I(N), I(C),
                                        I(D), I(H), I(W), I(K), ops._ptr(ws), Z(ws.numel()), ops._stream()), 'nc_convT_k2s2_fwd_split')
    return y, ys


def _from_s3(raw, N, C, S):
    t = raw.view(torch.bfloat16).view(N, C // 8, 3, S, 8).float()
    return ((t[:, :, 0] + t[:, :, 1]) + t[:, :, 2]).permute(0, 1, 3, 2).reshape(N, C, S)


@pytest.mark.parametrize('N,C,K,n', [(1, 128, 64, (9, 10, 13)), (2, 256, 128, (5, 6, 7)), (1, 128, 64, (3, 35, 35)), (1, 64, 32, (4, 4, 20))])
def test_split_conv_transpose_against_fp64_and_fp32(N, C, K, n):
    from neuroclear_amd import ops
    torch.manual_seed(5)
    x = torch.randn(N, C, *n, device=DEV)
    w = torch.randn(C, K, 2, 2, 2, device=DEV) * 0.05
    b = torch.randn(K, device=DEV)
    ref = torch.nn.functional.conv_transpose3d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride=2)
    sc = ref.pow(2).mean().sqrt().item()
    y, ys = _call(x, w, b, want_y=True, want_s3=True)
    y32 = ops.conv_transpose_fwd_raw(x, w, b) if hasattr(ops, 'conv_transpose_fwd_raw') else torch.nn.functional.conv_transpose3d(x, w, b, stride=2)

    def err(t):
        e = t.double().cpu() - ref
        return e.abs().max().item() / sc, e.pow(2).mean().sqrt().item() / sc
    ms, rs = err(y)
    m32, r32 = err(y32)
    print((N, C, K, n), 'split max %.2e rms %.2e | fp32 max %.2e rms %.2e' % (ms, rs, m32, r32))
    assert rs < 3e-7 and ms < 3e-6, (ms, rs)
    assert rs <= 1.5 * r32 + 5e-8, (rs, r32)
    # the S3 output of the same call is the exact three-term form of the fp32 output
    S2 = 8 * n[0] * n[1] * n[2]
    assert torch.equal(_from_s3(ys, N, K, S2), y.reshape(N, K, S2))
    # the operand converted by the call or by the caller: the same kernel, the same bits; no bias: exactly the bias less
    from neuroclear_amd._lib import I, check, lib
    xs = torch.empty(lib().nc_s3_bytes(I(N), I(C), ctypes.c_long(n[0] * n[1] * n[2])), dtype=torch.uint8, device=DEV)
    check(lib().nc_to_s3(ops._ptr(x), ops._ptr(xs), I(N), I(C), ctypes.c_long(n[0] * n[1] * n[2]), ops._stream()), 'nc_to_s3')
    y2, _ = _call(x, w, b, xs=xs)
    assert torch.equal(y, y2)


def test_split_conv_transpose_writes_its_slice_of_a_wider_s3_tensor():
    torch.manual_seed(6)
    N, C, K, n = 1, 128, 64, (4, 6, 9)
    x = torch.randn(N, C, *n, device=DEV)
    w = torch.randn(C, K, 2, 2, 2, device=DEV) * 0.05
    y, ys = _call(x, w, None, want_y=True, want_s3=True, ctot=128, c0=64)
    S2 = 8 * n[0] * n[1] * n[2]
    full = _from_s3(ys, N, 128, S2)
    assert torch.equal(full[:, 64:], y.reshape(N, K, S2))
    assert float(full[:, :64].abs().max()) == 0.0  # the other channels are not touched
