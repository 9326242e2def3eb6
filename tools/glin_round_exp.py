"""Experiment: where the 16-bit error of deep_linear_gen's output at initialisation comes from -- weights rounded to bf16
(round-to-nearest vs error-diffused over the taps of each (co, ci) pair) vs activations rounded between the layers."""
import contextlib, io, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from argparse import Namespace
from neuroclear_amd.models import create_model

o = Namespace(gpu_ids=[0], isTrain=True, image_dimension=3, checkpoints_dir='/tmp/nc_ckpt', name='c3',
              preprocess='none', gan_mode='lsgan', randomize_projection_depth=True, projection_depth=10,
              min_projection_depth=2, lambda_plane=[1, 1, 1], lambda_A=5.0, input_nc=1, output_nc=1, ngf=64,
              ndf=64, netG='unet_deconv', netG_B='deep_linear_gen', netD='basic', n_layers_D=3,
              norm='instance', no_dropout=True, init_type='kaiming', init_gain=0.02, lr=1e-4, beta1=0.1,
              direction='AtoB', model='axial_to_lateral_gan_apollo', precision='fp32')
torch.manual_seed(21); np.random.seed(21)
with contextlib.redirect_stdout(io.StringIO()):
    m = create_model(o)
ws = [p.detach().double() for p in m.netG_B.parameters()]
print([tuple(w.shape) for w in ws])
x = (0.4986 + 0.0116 * torch.randn(1, 1, 48, 48, 48, device='cuda')).double()


def rne(t):
    return t.float().bfloat16().double()


def diffuse(w):
    """bf16 values whose running sum over the taps of each (co, ci) pair follows the exact running sum"""
    K, C = w.shape[:2]
    f = w.reshape(K, C, -1).clone()
    out = torch.empty_like(f)
    r = torch.zeros(K, C, dtype=torch.double, device=w.device)
    for t in range(f.shape[2]):
        v = f[:, :, t] + r
        q = rne(v)
        out[:, :, t] = q
        r = v - q
    return out.reshape(w.shape)


def run(wl, round_act):
    h = rne(x) if round_act else x
    for i, w in enumerate(wl):
        h = F.conv3d(h, w, padding=w.shape[-1] // 2)
        if round_act and i < 3:
            h = rne(h)
    return h


ref = run(ws, False)
inner = (slice(None), slice(None), slice(8, -8), slice(8, -8), slice(8, -8))
def first3(f):
    return [f(w) if i < 3 else w for i, w in enumerate(ws)]


for name, wl, ra in (('3 conv RNE', first3(rne), False), ('3 conv diffused', first3(diffuse), False), ('3 conv diffused + act', first3(diffuse), True),
                     ('3 conv RNE + act', first3(rne), True),
                     ('weights RNE', [rne(w) for w in ws], False), ('weights diffused', [diffuse(w) for w in ws], False),
                     ('activations RNE', ws, True), ('both RNE', [rne(w) for w in ws], True),
                     ('diffused + act', [diffuse(w) for w in ws], True)):
    y = run(wl, ra)
    print('%-18s mean diff %+.5f (interior %+.5f)  rms %.5f   ref mean %.4f interior %.4f' % (
        name, float((y - ref).mean()), float((y - ref)[inner].mean()), float((y - ref).pow(2).mean().sqrt()),
        float(ref.mean()), float(ref[inner].mean())))
