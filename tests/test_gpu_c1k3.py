"""Conv3d 1 -> 64 channels, 3^3, padding 1 (the first layer of unet_deconv, models/networks.py:420-425) on its own fp32 MFMA kernel
(csrc/conv_c1k3.hip): against fp64 and against the direct VALU path, ragged shapes included."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('N,K,n', [(1, 64, (12, 12, 12)), (2, 64, (5, 9, 37)), (1, 128, (3, 7, 8)), (1, 64, (4, 3, 140)), (1, 64, (2, 70, 33))])
@pytest.mark.parametrize('bias', [True, False])
def test_one_channel_3x3x3_forward(N, K, n, bias):
    from neuroclear_amd import ops
    from neuroclear_amd._lib import I, lib
    torch.manual_seed(3)
    x = torch.randn(N, 1, *n, device=DEV)
    w = torch.randn(K, 1, 3, 3, 3, device=DEV) * 0.2
    b = torch.randn(K, device=DEV) if bias else None
    assert lib().nc_conv_fwd_path(I(1), I(K), I(3), I(3), I(3), I(1), I(1)) == 1
    y = ops.conv_fwd_raw(x, w, b, 1, 1)
    ref = F.conv3d(x.double().cpu(), w.double().cpu(), None if b is None else b.double().cpu(), padding=1)
    sc = ref.pow(2).mean().sqrt().item()
    e = (y.double().cpu() - ref).abs().max().item() / sc
    assert e < 2e-6, e
    lib().nc_set_force_direct(I(1))
    try:
        yd = ops.conv_fwd_raw(x, w, b, 1, 1)
    finally:
        lib().nc_set_force_direct(I(0))
    assert (y - yd).abs().max().item() / sc < 2e-6
