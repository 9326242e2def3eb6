"""Experiment: fake / rec of the configs[3] Apollo model (seed 21, 148^3 x 4 or smaller) in bf16 against fp32."""
import contextlib, io, sys
import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_config3 as T
from argparse import Namespace
from neuroclear_amd.models import create_model
from neuroclear_amd.util import seed as S

crop, batch = int(sys.argv[1]), int(sys.argv[2])
out = {}
for prec in ('fp32', 'bf16'):
    o = Namespace(gpu_ids=[0], isTrain=True, image_dimension=3, checkpoints_dir='/tmp/nc_ckpt', name='c3',
                  preprocess='none', gan_mode='lsgan', randomize_projection_depth=True, projection_depth=10,
                  min_projection_depth=2, lambda_plane=[1, 1, 1], lambda_A=5.0, input_nc=1, output_nc=1, ngf=64,
                  ndf=64, netG='unet_deconv', netG_B='deep_linear_gen', netD='basic', n_layers_D=3,
                  norm='instance', no_dropout=True, init_type='kaiming', init_gain=0.02, lr=1e-4, beta1=0.1,
                  direction='AtoB', model='axial_to_lateral_gan_apollo', precision=prec)
    torch.manual_seed(21); np.random.seed(21)
    with contextlib.redirect_stdout(io.StringIO()):
        m = create_model(o)
    vols = [S.random_volume(100 + b, crop) for b in range(batch)]
    real = torch.stack([torch.from_numpy((v.astype(np.float64) / 65535.0).astype(np.float32))[None] for v in vols]).cuda()
    m.set_input({'A': real, 'A_paths': 'synthetic'})
    m.forward()
    out[prec] = (m.fake.detach().double(), m.rec.detach().double())
    print(prec, 'cycle', float((m.rec.detach() - real).abs().mean() * 5), 'mean fake %.5f rec %.5f real %.5f' % (
        float(m.fake.mean()), float(m.rec.mean()), float(real.mean())))
    del m
for i, n in enumerate(('fake', 'rec')):
    a, b = out['bf16'][i].flatten(), out['fp32'][i].flatten()
    print(n, 'rms rel %.3e  slope %.5f  mean diff %.3e  mean|b| %.4f std b %.4f' % (
        float((a - b).norm() / b.norm()), float((a * b).sum() / (b * b).sum()), float((a - b).mean()), float(b.abs().mean()), float(b.std())))
