"""Accuracy of an fp32 convolution computed as bf16 x bf16 MFMA products of a three-term operand split (a = a0 + a1 + a2, each
term a bf16; products a_i * b_j with i + j <= 2 -> six MFMAs), emulated here with six calls of the existing bf16 kernel, against
the fp32 MFMA kernel -- both measured against an fp64 reference on the CPU."""
import sys
import torch
sys.path.insert(0, '.')
from neuroclear_amd import ops


def split3(t):
    a0 = t.bfloat16().float()
    r = t - a0
    a1 = r.bfloat16().float()
    r = r - a1
    a2 = r.bfloat16().float()
    return a0, a1, a2


def main():
    torch.manual_seed(0)
    dev = 'cuda'
    for (C, K, n) in ((64, 64, 24), (128, 64, 20), (16, 64, 28)):
        x = torch.randn(1, C, n, n, n, device=dev)
        x = torch.relu(x) * 1.3 + 0.01 * torch.randn_like(x)
        w = torch.randn(K, C, 3, 3, 3, device=dev) * 0.02
        ref = torch.nn.functional.conv3d(x.double().cpu(), w.double().cpu(), padding=1)
        scale = ref.pow(2).mean().sqrt().item()
        ops.set_conv_precision('fp32')
        y32 = ops.conv_fwd_raw(x, w, None, 1, 1)
        ops.set_conv_precision('bf16')
        xs, ws = split3(x), split3(w)
        assert (xs[0] + xs[1] + xs[2] == x).all() and (ws[0] + ws[1] + ws[2] == w).all()
        terms = {}
        for i in range(3):
            for j in range(3):
                terms[(i, j)] = ops.conv_fwd_raw(xs[j], ws[i], None, 1, 1)
        def total(pairs):
            acc = torch.zeros_like(y32)
            for ij in sorted(pairs, key=lambda ij: -(ij[0] + ij[1])):
                acc += terms[ij]
            return acc
        y1 = total([(0, 0)])
        y3 = total([(0, 0), (0, 1), (1, 0)])
        y6 = total([(i, j) for i in range(3) for j in range(3) if i + j <= 2])
        y9 = total([(i, j) for i in range(3) for j in range(3)])
        ops.set_conv_precision('fp32')
        for name, y in (('fp32 mfma', y32), ('bf16 x1', y1), ('bf16 x3', y3), ('bf16 x6', y6), ('bf16 x9', y9)):
            e = (y.double().cpu() - ref)
            print('C=%3d K=%3d n=%2d  %-10s max %.3e  rms %.3e   (relative to output rms %.3f)' % (
                C, K, n, name, e.abs().max().item() / scale, e.pow(2).mean().sqrt().item() / scale, scale))


if __name__ == '__main__':
    main()
