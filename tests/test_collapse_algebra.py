"""The algebra behind the default evaluation of DeepLinearGenerator (reference models/networks.py:893-917: Conv3d 7^3 1 -> 64, 5^3, 3^3 64 -> 64,
then 1 x 1: 64 -> 32 -> 16 -> 1; no bias, nothing between the layers, every convolution zero-pads its own input), checked on the CPU in fp64
against torch autograd of the layer-by-layer chain -- independent of any kernel (csrc/gen_nets.hip):

  * layers 2 .. 5 are ONE 64 -> 1 convolution of act1 with E[c][t] = sum_k e[k] W2[k][c][t], e = W5 W4 W3 -- exact at the faces too;
  * dW2 .. dW5 and dL/dact1 follow from dy, act1 and the weights (q[c][t] = sum_v dy[v] act1[c][v + t - 1]);
  * the 5^3 layer's two gradients follow from 27 shifted copies of the one-channel dy, Dsh[a][v] = dy[v - (a - 1)] inside the volume:
    dW1 = E . P with P a weight gradient between Dsh and act0, dL/dact0 = a forward convolution of Dsh with composed weights Wf;
  * act1 itself is needed by nobody: y is a shifted sum of a 64 -> 27 convolution of act0, q a contraction of P.
Odd, unequal extents on purpose: every face, edge and corner of the volume is a different case of the padding argument."""
import numpy as np
import torch
import torch.nn.functional as F


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 1, 7, 7, 7), (64, 64, 5, 5, 5), (64, 64, 3, 3, 3), (32, 64, 1, 1, 1), (16, 32, 1, 1, 1), (1, 16, 1, 1, 1)]
    return [(torch.randn(s, generator=g, dtype=torch.float64) / np.sqrt(np.prod(s[1:]))).requires_grad_(True) for s in shapes]


def _chain(x, W):
    a0 = F.conv3d(x, W[0], padding=3)
    a1 = F.conv3d(a0, W[1], padding=2)
    a2 = F.conv3d(a1, W[2], padding=1)
    return F.conv3d(F.conv3d(F.conv3d(a2, W[3]), W[4]), W[5]), a0, a1


def test_collapsed_tail_and_rank_forms_equal_the_layered_chain():
    torch.manual_seed(0)
    D, H, Wd = 7, 9, 10
    W = _weights(3)
    x = torch.randn(1, 1, D, H, Wd, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(1, 1, D, H, Wd, dtype=torch.float64)
    y, a0, a1 = _chain(x, W)
    a0.retain_grad()
    a1.retain_grad()
    (y * dy).sum().backward()

    with torch.no_grad():
        w2, w3, w4, w5 = W[2], W[3][:, :, 0, 0, 0], W[4][:, :, 0, 0, 0], W[5][:, :, 0, 0, 0]
        a = w5 @ w4                     # 1 x 32
        e = a @ w3                      # 1 x 64
        E = torch.einsum('k,kcdhw->cdhw', e[0], w2)                      # [64][3][3][3]
        # forward: ONE 64 -> 1 convolution of act1
        yc = F.conv3d(a1, E[None], padding=1)
        assert torch.allclose(yc, y, rtol=0, atol=1e-12)
        # dL/dact1 = flip(E) (*) dy: a 1 -> 64 convolution of the one-channel dy
        g1 = F.conv3d(dy, E.flip(1, 2, 3)[:, None], padding=1)
        assert torch.allclose(g1, a1.grad, rtol=0, atol=1e-12)
        # q[c][t] = sum_v dy[v] act1[c][v + t - 1], then everything in weight space
        a1p = F.pad(a1[0], (1, 1, 1, 1, 1, 1))
        q = torch.stack([torch.stack([torch.stack([(a1p[:, tz:tz + D, ty:ty + H, tx:tx + Wd] * dy[0]).sum((1, 2, 3)) for tx in range(3)], -1)
                                      for ty in range(3)], -2) for tz in range(3)], -3)   # [64][3][3][3]
        assert torch.allclose(torch.einsum('k,cdhw->kcdhw', e[0], q), W[2].grad, rtol=0, atol=1e-11)
        r = torch.einsum('kcdhw,cdhw->k', w2, q)
        assert torch.allclose(torch.outer(a[0], r), W[3].grad[:, :, 0, 0, 0], rtol=0, atol=1e-11)
        s = w3 @ r
        assert torch.allclose(torch.outer(w5[0], s), W[4].grad[:, :, 0, 0, 0], rtol=0, atol=1e-11)
        assert torch.allclose((w4 @ s)[None], W[5].grad[:, :, 0, 0, 0], rtol=0, atol=1e-11)

        # the 5^3 layer from the 27 shifted copies of dy (the truncation of g1 to the volume is IN Dsh)
        dyp = F.pad(dy[0, 0], (1, 1, 1, 1, 1, 1))
        Dsh = torch.stack([dyp[2 - az:2 - az + D, 2 - ay:2 - ay + H, 2 - ax:2 - ax + Wd] for az in range(3) for ay in range(3) for ax in range(3)])
        Ef = E.reshape(64, 27)
        assert torch.allclose(torch.einsum('ka,adhw->kdhw', Ef, Dsh)[None], a1.grad, rtol=0, atol=1e-12)   # g1 = E . Dsh
        a0p = F.pad(a0[0], (2, 2, 2, 2, 2, 2))
        P = torch.stack([torch.stack([torch.stack([torch.einsum('adhw,cdhw->ac', Dsh, a0p[:, tz:tz + D, ty:ty + H, tx:tx + Wd]) for tx in range(5)], -1)
                                      for ty in range(5)], -2) for tz in range(5)], -3)    # [27][64][5][5][5]
        assert torch.allclose(torch.einsum('ka,acdhw->kcdhw', Ef, P), W[1].grad, rtol=0, atol=1e-11)
        Wf = torch.einsum('kcdhw,ka->cadhw', W[1].flip(2, 3, 4), Ef)                        # [64][27][5][5][5]
        g0 = F.conv3d(Dsh[None], Wf, padding=2)
        assert torch.allclose(g0, a0.grad, rtol=0, atol=1e-11)
        # the forward without act1: Z_t = F_t (*) act0 (a 64 -> 27 convolution), y[v] = sum_t [v + t - 1 inside] Z_t[v + t - 1]
        Ft = torch.einsum('ka,kcdhw->acdhw', Ef, W[1])                                      # [27][64][5][5][5]
        Z = F.conv3d(a0, Ft, padding=2)[0]
        Zp = F.pad(Z, (1, 1, 1, 1, 1, 1))
        ysum = sum(Zp[(tz * 3 + ty) * 3 + tx, tz:tz + D, ty:ty + H, tx:tx + Wd] for tz in range(3) for ty in range(3) for tx in range(3))
        assert torch.allclose(ysum[None, None], y, rtol=0, atol=1e-11)
        # ... and q itself is a contraction of the same P (no pass over act1 is needed for it)
        assert torch.allclose(torch.einsum('kcdhw,acdhw->ka', W[1], P).reshape(64, 3, 3, 3), q, rtol=0, atol=1e-11)


def test_composing_layers_0_and_1_is_not_exact_at_the_faces():
    """Why the collapse stops at layer 2: layers 0 / 1 pad DIFFERENT tensors.  A composed 11^3 kernel reproduces act1 only where its reach stays
    away from the faces; the mismatch sits within 2 voxels of a face (the 5^3 kernel's reach into the zero padding of act0)."""
    W = _weights(4)
    x = torch.randn(1, 1, 12, 13, 14, dtype=torch.float64)
    with torch.no_grad():
        _, _, a1 = _chain(x, W)
        C = torch.zeros(64, 11, 11, 11, dtype=torch.float64)   # C[k][r] = sum_c sum_{s + b = r} W1[k][c][s] W0[c][b]
        for sz in range(5):
            for sy in range(5):
                for sx in range(5):
                    C[:, sz:sz + 7, sy:sy + 7, sx:sx + 7] += torch.einsum('kc,cdhw->kdhw', W[1][:, :, sz, sy, sx], W[0][:, 0])
        comp = F.conv3d(x, C[:, None], padding=5)
        d = (comp - a1).abs()[0].amax(0)
        assert float(d[2:-2, 2:-2, 2:-2].max()) < 1e-11
        assert float(d.max()) > 1e-3


def _types(D, H, Wd):
    """Per voxel, per axis: 0 on the low face, 2 on the high face, 1 inside (extents >= 2)."""
    def ax(L):
        t = torch.ones(L, dtype=torch.long)
        t[0], t[L - 1] = 0, 2
        return t
    tz, ty, tx = ax(D), ax(H), ax(Wd)
    return (tz[:, None, None] * 9 + ty[None, :, None] * 3 + tx[None, None, :]).expand(D, H, Wd)   # type index 0 .. 26, 13 = interior


def _allowed(tau):
    """Taps t of the 3^3 kernel that stay inside the volume at a voxel of type tau: per axis, a low-face voxel cannot use t = 0, a high-face one t = 2."""
    m = torch.ones(3, 3, 3, dtype=torch.float64)
    for axis, ta in enumerate((tau // 9, (tau // 3) % 3, tau % 3)):
        if ta == 0:
            m.index_fill_(axis, torch.tensor([0]), 0.0)
        if ta == 2:
            m.index_fill_(axis, torch.tensor([2]), 0.0)
    return m


def test_layers_1_to_5_as_one_position_typed_7x7x7_kernel():
    """Round 6: layers 1 .. 5 -- the 5^3 64 -> 64 layer AND the collapsed tail -- are ONE 64 -> 1 convolution of act0 with a 7^3
    kernel that depends only on the voxel's POSITION TYPE (27 types: per axis low face / inside / high face): H_tau = sum over the taps t the
    type allows of E[., t] composed with W1.  The 3^3 kernel E reaches one voxel, so the zero padding of act1 only matters ON the faces, and
    there it just removes the taps that point outside: exact everywhere (faces, edges, corners), and the 64 x 64 x 125 products per voxel of the
    5^3 layer become 64 x 343.  Backward: dL/dact0 = sum_v dy[v] H_tau(v) shifted; every parameter gradient passes through
    dH_tau[c][d] = sum over the voxels of type tau of dy[v] act0[c][v + d - 3] and weight-space contractions."""
    torch.manual_seed(1)
    D, H, Wd = 5, 6, 7
    W = _weights(5)
    x = torch.randn(1, 1, D, H, Wd, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(1, 1, D, H, Wd, dtype=torch.float64)
    y, a0, a1 = _chain(x, W)
    a0.retain_grad()
    a1.retain_grad()
    (y * dy).sum().backward()
    with torch.no_grad():
        w1, w2, w3, w4, w5 = W[1], W[2], W[3][:, :, 0, 0, 0], W[4][:, :, 0, 0, 0], W[5][:, :, 0, 0, 0]
        e = (w5 @ w4) @ w3
        E = torch.einsum('k,kcdhw->cdhw', e[0], w2)                                           # [64 c'][3][3][3]
        typ = _types(D, H, Wd)

        def compose(Em):  # H[c][d] = sum_{c', t, s: t + s = d} Em[c'][t] W1[c'][c][s]: a full 3-D convolution of the two kernels per (c', c)
            Hk = torch.zeros(64, 7, 7, 7, dtype=torch.float64)
            for tz in range(3):
                for ty in range(3):
                    for tx in range(3):
                        Hk[:, tz:tz + 5, ty:ty + 5, tx:tx + 5] += torch.einsum('k,kcdhw->cdhw', Em[:, tz, ty, tx], w1)
            return Hk
        Hs = torch.stack([compose(E * _allowed(tau)[None]) for tau in range(27)])              # [27][64][7][7][7]
        a0p = F.pad(a0[0], (3, 3, 3, 3, 3, 3))
        # forward: the interior kernel everywhere (ONE convolution), the boundary voxels recomputed with their own type's kernel
        y13 = F.conv3d(a0, Hs[13][None], padding=3)[0, 0]
        yt = y13.clone()
        for z in range(D):
            for yy in range(H):
                for xx in range(Wd):
                    tau = int(typ[z, yy, xx])
                    if tau != 13:
                        yt[z, yy, xx] = (Hs[tau] * a0p[:, z:z + 7, yy:yy + 7, xx:xx + 7]).sum()
        assert torch.allclose(yt[None, None], y, rtol=0, atol=1e-11)
        interior = typ == 13
        assert torch.allclose(y13[interior], y[0, 0][interior], rtol=0, atol=1e-11) and not torch.allclose(y13, y[0, 0], rtol=0, atol=1e-6)
        # dL/dact0 = the 1 -> 64 convolution of dy with the flipped interior kernel + the boundary voxels' difference kernels scattered
        g0 = F.conv3d(dy, Hs[13].flip(1, 2, 3)[:, None], padding=3)[0]
        g0p = F.pad(torch.zeros_like(g0), (3, 3, 3, 3, 3, 3))
        for z in range(D):
            for yy in range(H):
                for xx in range(Wd):
                    tau = int(typ[z, yy, xx])
                    if tau != 13:
                        g0p[:, z:z + 7, yy:yy + 7, xx:xx + 7] += dy[0, 0, z, yy, xx] * (Hs[tau] - Hs[13])
        g0 = g0 + g0p[:, 3:3 + D, 3:3 + H, 3:3 + Wd]
        assert torch.allclose(g0[None], a0.grad, rtol=0, atol=1e-11)
        # parameter gradients: dH_tau, then weight space
        dH = torch.zeros(27, 64, 7, 7, 7, dtype=torch.float64)
        for z in range(D):
            for yy in range(H):
                for xx in range(Wd):
                    dH[int(typ[z, yy, xx])] += dy[0, 0, z, yy, xx] * a0p[:, z:z + 7, yy:yy + 7, xx:xx + 7]
        q = torch.zeros(64, 3, 3, 3, dtype=torch.float64)
        dw1 = torch.zeros_like(w1)
        for tau in range(27):
            al = _allowed(tau)
            for tz in range(3):
                for ty in range(3):
                    for tx in range(3):
                        if al[tz, ty, tx] == 0:
                            continue
                        sl = dH[tau][:, tz:tz + 5, ty:ty + 5, tx:tx + 5]                       # [c][s]
                        q[:, tz, ty, tx] += torch.einsum('cdhw,kcdhw->k', sl, w1)
                        dw1 += torch.einsum('k,cdhw->kcdhw', E[:, tz, ty, tx], sl)
        assert torch.allclose(dw1, W[1].grad, rtol=0, atol=1e-11)
        a1p = F.pad(a1[0], (1, 1, 1, 1, 1, 1))
        q_ref = torch.stack([torch.stack([torch.stack([(a1p[:, tz:tz + D, ty:ty + H, tx:tx + Wd] * dy[0]).sum((1, 2, 3)) for tx in range(3)], -1)
                                          for ty in range(3)], -2) for tz in range(3)], -3)
        assert torch.allclose(q, q_ref, rtol=0, atol=1e-11)   # (dW2 .. dW5 follow from q as in the test above)
