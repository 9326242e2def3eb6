"""Experiment: deep_linear_gen forward in bf16 (whole-network 16-bit call) against fp32 on the same input / weights:
relative error and the regression slope of the 16-bit output on the fp32 output (a slope != 1 is a systematic bias)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd.models import networks

torch.manual_seed(3)
size, batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 2
kind = sys.argv[2] if len(sys.argv) > 2 else 'deep_linear_gen'
net = networks.define_G(1, 1, 64, kind, 'instance', False, 'kaiming', 0.02, [0])
x = torch.rand(batch, 1, size, size, size, device='cuda')
with torch.enable_grad():
    y32 = net(x).detach()
    for prec in ('bf16', 'fp16'):
        ops.set_conv_precision(prec)
        y16 = net(x).detach()
        ops.set_conv_precision("fp32")
        a, b = y16.double().flatten(), y32.double().flatten()
        slope = float((a * b).sum() / (b * b).sum())
        print(prec, 'max rel err %.3e  rms rel %.3e  slope %.6f  mean|y32| %.4f' % (
            float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm()), slope, float(b.abs().mean())))
        inner = (slice(None), slice(None), slice(8, -8), slice(8, -8), slice(8, -8))
        a, b = y16[inner].double().flatten(), y32[inner].double().flatten()
        print('   interior: rms rel %.3e slope %.6f' % (float((a - b).norm() / b.norm()), float((a * b).sum() / (b * b).sum())))
