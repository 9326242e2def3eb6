"""deep_linear_gen: nc_set_dl_collapse(2) (layers 1 .. 5 as one position-typed 7^3 kernel, csrc/dl_typed.hip) against mode 1 (round 5's collapsed tail +
rank forms), mode 0 (layer by layer) and an fp64 torch evaluation of the reference chain; timings of forward + backward.  usage: python tools/dl_typed_check.py"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from neuroclear_amd import ops
from neuroclear_amd._lib import lib, I
from neuroclear_amd.models import networks

L = lib()
torch.manual_seed(3)
net = networks.define_G(1, 1, 64, 'deep_linear_gen', 'instance', False, 'kaiming', 0.02, [0])
Ws = [p.detach().double() for p in net.parameters()]


def ref64(x, r):
    ws = [w.clone().requires_grad_(True) for w in Ws]
    xi = x.double().clone().requires_grad_(True)
    a = F.conv3d(xi, ws[0], padding=3)
    a = F.conv3d(a, ws[1], padding=2)
    a = F.conv3d(a, ws[2], padding=1)
    y = F.conv3d(F.conv3d(F.conv3d(a, ws[3]), ws[4]), ws[5])
    (y * r.double()).sum().backward()
    return y.detach(), xi.grad, [w.grad for w in ws]


def run(mode, x, r):
    L.nc_set_dl_collapse(I(mode))
    for q in net.parameters():
        q.grad = None
    xi = x.clone().requires_grad_(True)
    y = net(xi)
    (y * r).sum().backward()
    return y.detach().clone(), xi.grad.clone(), [q.grad.clone() for q in net.parameters()]


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


g = torch.Generator(device='cuda').manual_seed(1)
for shape in ((1, 1, 16, 16, 16), (1, 1, 12, 20, 28), (2, 1, 16, 24, 16), (1, 1, 40, 40, 40), (1, 1, 108, 108, 108)):
    x = torch.rand(shape, device='cuda', generator=g)
    r = torch.randn(shape, device='cuda', generator=g)
    big = shape[-1] >= 100
    if not big:
        y64, dx64, g64 = ref64(x, r)
    out = {m: run(m, x, r) for m in (2, 1, 0)}
    msg = []
    for m in (2, 1, 0):
        y, dx, gs = out[m]
        if big:
            yr, dxr, gr = [t.double() for t in out[0][:2]] + [[t.double() for t in out[0][2]]]
        else:
            yr, dxr, gr = y64, dx64, g64
        msg.append('mode %d: y %.2e dx %.2e dW ' % (m, rel(y, yr), rel(dx, dxr)) + ' '.join('%.1e' % rel(a, b) for a, b in zip(gs, gr)))
    print(shape, 'vs', 'mode 0' if big else 'fp64', '\n   ' + '\n   '.join(msg), flush=True)
    y2, dx2, g2 = run(2, x, r)
    print('   run-to-run identical:', bool(torch.equal(y2, out[2][0]) and torch.equal(dx2, out[2][1]) and all(torch.equal(a, b) for a, b in zip(g2, out[2][2]))), flush=True)

x = torch.rand(1, 1, 108, 108, 108, device='cuda', generator=g)
r = torch.randn(1, 1, 108, 108, 108, device='cuda', generator=g)
for m in (2, 1, 2, 1, 0):
    run(m, x, r); run(m, x, r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run(m, x, r)
    e1.record()
    torch.cuda.synchronize()
    print('mode %d: forward + backward at 108^3: %.3f ms' % (m, e0.elapsed_time(e1) / 10), flush=True)
L.nc_set_dl_collapse(I(2))
