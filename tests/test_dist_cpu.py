"""CPU, world_size 2 and 3, gloo: the N > 1 paths that need no GPU code --
 * the cube sharding schedules of diced inference (neuroclear_amd.test_dice): 'gather' (lock-step rounds, tiles to rank 0,
   bit-identical to the single-process order), 'reduce' (each rank overlap-adds its own cubes, one reduce(sum); +-1 LSB) and
   'slab' (contiguous cube ranges, one point-to-point exchange of owned z-slabs, per-rank finalisation, integer slabs to rank 0;
   +-1 LSB), all checked against the oracle's single-process assemble; 125 cubes = an odd count, like 729;
 * the weight broadcast in front of the loop (broadcast_parameters);
 * the gradient exchange of the flat optimizer buffers (FlatAdam.all_reduce_mean, bucketed / asynchronous form too)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _dice_worker(rank, world, port, out_path):
    _init(rank, world, port)
    from neuroclear_amd.test_dice import sharded_cube_loop
    from neuroclear_amd.util import seed as S
    from oracle import dice as odice
    vol = S.random_volume(3, (50, 50, 50))  # 5 x 5 x 5 = 125 cubes: an odd count leaves rank 1 idle in the last round
    R, ov, b = 16, 4, 2
    padded = odice.pad_for_dicing(vol, R, ov)
    steps = odice.grid_steps(padded.shape, R, ov)
    n = steps[0] * steps[1] * steps[2]
    refl = odice.reflect_pad(padded, b)

    def net(c):  # stand-in network: position independent, non-trivial
        return c * 0.75 + 0.01

    got = {}
    E = R + 2 * b
    sharded_cube_loop(n, rank, world,
                      produce=lambda i: net(torch.from_numpy(odice.normalize(odice.cut_cube(refl, i, steps, R, ov, b)))),
                      consume=lambda j, t: got.__setitem__(j, t.numpy().copy()),
                      empty_like=lambda: torch.zeros((E, E, E)))
    if rank == 0:
        assert sorted(got) == list(range(n))
        assert list(got) == list(range(n)), 'rank 0 must consume the cubes in index order'
        out = odice.assemble([got[j] for j in range(n)], padded.shape, vol.shape, R, ov, b, 'uint16')
        ref = odice.assemble([net(torch.from_numpy(odice.normalize(odice.cut_cube(refl, i, steps, R, ov, b)))).numpy()
                              for i in range(n)], padded.shape, vol.shape, R, ov, b, 'uint16')
        np.save(out_path, np.array([int(np.array_equal(out, ref)), n]))
    else:
        assert not got
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_dice_loop_gather(tmp_path, world):
    port = _free_port()
    out = str(tmp_path / 'r.npy')
    mp.spawn(_dice_worker, args=(world, port, out), nprocs=world, join=True)
    eq, n = np.load(out)
    assert eq == 1 and n == 125


def _dice_reduce_worker(rank, world, port, out_path):
    _init(rank, world, port)
    from neuroclear_amd.test_dice import broadcast_parameters, sharded_cube_loop_reduce
    from neuroclear_amd.util import seed as S
    from oracle import dice as odice
    vol = S.random_volume(3, (50, 50, 50))
    R, ov, b = 16, 4, 2
    padded = odice.pad_for_dicing(vol, R, ov)
    steps = odice.grid_steps(padded.shape, R, ov)
    n = steps[0] * steps[1] * steps[2]
    refl = odice.reflect_pad(padded, b)
    step = R - ov
    # "weights" of the stand-in network differ per rank until rank 0's are broadcast
    lin = torch.nn.Linear(1, 1)
    with torch.no_grad():
        lin.weight.fill_(0.75 + rank)
        lin.bias.fill_(0.01 + rank)
    broadcast_parameters(lin, 0)
    w, c = float(lin.weight), float(lin.bias)

    def net(x):
        return x * w + c

    acc = torch.zeros(padded.shape, dtype=torch.float32)
    mine = []

    def add_local(i, tile):  # the assembler's per-cube work on this rank's own accumulator (assemble_dice.py:167-173)
        mine.append(i)
        zi, yi, xi = i // (steps[1] * steps[2]), (i % (steps[1] * steps[2])) // steps[2], i % steps[2]
        z, y, x = zi * step, yi * step, xi * step
        acc[z:z + R, y:y + R, x:x + R] += tile[b:-b, b:-b, b:-b] / 8

    sharded_cube_loop_reduce(n, rank, world,
                             produce=lambda i: net(torch.from_numpy(odice.normalize(odice.cut_cube(refl, i, steps, R, ov, b)))),
                             add_local=add_local, accumulator=lambda: acc)
    assert mine == list(range(rank, n, world))
    if rank == 0:
        ref = odice.assemble([(torch.from_numpy(odice.normalize(odice.cut_cube(refl, i, steps, R, ov, b))) * 0.75 + 0.01).numpy()
                              for i in range(n)], padded.shape, vol.shape, R, ov, b, 'uint16')
        got = odice.finalize(acc.numpy(), padded.shape, vol.shape, R, ov, 'uint16')
        d = int(np.abs(got.astype(np.int64) - ref.astype(np.int64)).max())
        np.save(out_path, np.array([d, n, int(w == 0.75 and abs(c - 0.01) < 1e-9)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_dice_loop_reduce(tmp_path, world):
    """One reduce(sum) of per-rank accumulators: summation order differs from the index order by <= 1 ulp per voxel, the
    truncating uint16 cast may flip by 1 LSB -- and rank 0's weights must have reached every rank."""
    port = _free_port()
    out = str(tmp_path / 'r.npy')
    mp.spawn(_dice_reduce_worker, args=(world, port, out), nprocs=world, join=True)
    d, n, bc = np.load(out)
    assert d <= 1 and n == 125 and bc == 1


def _dice_slab_worker(rank, world, port, out_path):
    _init(rank, world, port)
    from neuroclear_amd.test_dice import slab_exchange, slab_gather, slab_plan
    from neuroclear_amd.util import seed as S
    from oracle import dice as odice
    vol = S.random_volume(3, (50, 50, 50))
    R, ov, b = 16, 4, 2
    padded = odice.pad_for_dicing(vol, R, ov)
    steps = odice.grid_steps(padded.shape, R, ov)
    n = steps[0] * steps[1] * steps[2]
    refl = odice.reflect_pad(padded, b)
    step = R - ov
    plan = slab_plan(steps, step, R, padded.shape[0], world)
    ok = plan['cubes'][0][0] == 0 and plan['cubes'][-1][1] == n and all(plan['cubes'][r][1] == plan['cubes'][r + 1][0] for r in range(world - 1))
    ok = ok and max(c[1] - c[0] for c in plan['cubes']) - min(c[1] - c[0] for c in plan['cubes']) <= 1  # <= 1 cube of imbalance
    ok = ok and plan['own'][0][0] == 0 and plan['own'][-1][1] == padded.shape[0]

    def net(x):
        return x * 0.75 + 0.01
    za, zb = plan['local'][rank]
    local = torch.zeros((zb - za,) + padded.shape[1:], dtype=torch.float32)
    for i in range(*plan['cubes'][rank]):
        tile = net(torch.from_numpy(odice.normalize(odice.cut_cube(refl, i, steps, R, ov, b))))
        zi, yi, xi = i // (steps[1] * steps[2]), (i % (steps[1] * steps[2])) // steps[2], i % steps[2]
        z, y, x = zi * step, yi * step, xi * step
        assert za <= z and z + R <= zb  # every cube of the rank lies inside the planes its accumulator holds
        local[z - za:z - za + R, y:y + R, x:x + R] += tile[b:-b, b:-b, b:-b] / 8
    ok = ok and (zb - za) < padded.shape[0]  # the point of the exercise: no rank holds the whole volume
    o0, o1 = plan['own'][rank]
    own = torch.zeros((o1 - o0,) + padded.shape[1:], dtype=torch.float32)
    slab_exchange(rank, world, plan, local, own)
    # finalise the owned planes: count volume, (acc / count) * 8 * 65535, truncating cast, crop of the dicing pad (oracle arithmetic on a
    # volume that is zero outside the slab: only the slab's planes are kept)
    full = np.zeros(padded.shape, np.float32)
    full[o0:o1] = own.numpy()
    fin = odice.finalize(full, padded.shape, vol.shape, R, ov, 'uint16')
    z0, z1 = min(o0, vol.shape[0]), min(o1, vol.shape[0])
    slab = torch.from_numpy(np.ascontiguousarray(fin[z0:z1]).view(np.int16))  # the wire dtype of the product: int16 bits of uint16
    parts = slab_gather(rank, world, plan, slab, vol.shape[0])
    if rank == 0:
        got = torch.cat(parts, 0).numpy().view(np.uint16).astype(np.int64)
        ref = odice.assemble([net(torch.from_numpy(odice.normalize(odice.cut_cube(refl, i, steps, R, ov, b)))).numpy() for i in range(n)],
                             padded.shape, vol.shape, R, ov, b, 'uint16')
        d = int(np.abs(got - ref.astype(np.int64)).max()) if got.shape == ref.shape else 99
        np.save(out_path, np.array([d, n, int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 4])
def test_sharded_dice_loop_slab(tmp_path, world):
    """assemble='slab': contiguous cube ranges, local accumulators of a few z-layers, ONE point-to-point exchange of the planes each
    rank owns, per-rank finalisation, integer slabs to rank 0 -- against the oracle's single-process assemble (+-1 LSB: the order in
    which the ranks' partial sums meet differs from the index order)."""
    port = _free_port()
    out = str(tmp_path / 's.npy')
    mp.spawn(_dice_slab_worker, args=(world, port, out), nprocs=world, join=True)
    d, n, ok = np.load(out)
    assert d <= 1 and n == 125 and ok == 1


def _adam_worker(rank, world, port, out_path):
    _init(rank, world, port)
    from neuroclear_amd.models.axial_to_lateral_gan_apollo_model import FlatAdam
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    opt = FlatAdam(params, lr=1e-4, betas=(0.1, 0.999))
    assert params[0].data.data_ptr() == opt.flat.data_ptr()  # parameters are views into the flat buffer
    opt.zero_grad()
    loss = (params[0] * (rank + 1)).sum() + (params[1] ** 2).sum() * (rank + 1)
    loss.backward()
    opt._collect()  # gradients that autograd produced outside the flat buffer are copied into it
    g_local = opt.grad.clone()
    opt.all_reduce_mean()
    gathered = [torch.empty_like(g_local) for _ in range(world)]
    dist.all_gather(gathered, g_local)
    want = sum(gathered) / world
    ok = torch.allclose(opt.grad, want, rtol=0, atol=1e-7) and torch.equal(params[0].grad.reshape(-1), opt.grad[:15])
    if rank == 0:
        np.save(out_path, np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_adam_gradient_exchange_two_ranks(tmp_path):
    port = _free_port()
    out = str(tmp_path / 'a.npy')
    mp.spawn(_adam_worker, args=(2, port, out), nprocs=2, join=True)
    assert np.load(out)[0] == 1


def _adam_overlap_worker(rank, world, port, out_path):
    _init(rank, world, port)
    from neuroclear_amd.models.axial_to_lateral_gan_apollo_model import FlatAdam
    torch.manual_seed(0)
    shapes = [(40, 8), (40,), (64, 16), (64,), (8, 8), (8,)]

    def build(overlap):
        torch.manual_seed(1)
        ps = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
        return ps, FlatAdam(ps, lr=1e-3, betas=(0.1, 0.999), overlap_all_reduce=overlap)

    def loss(ps):
        return sum(((p * (rank + 1 + i)) ** 2).sum() for i, p in enumerate(ps))

    ok = True
    pa, oa = build(False)
    pb, ob = build(True)
    for it in range(3):
        for ps, o in ((pa, oa), (pb, ob)):
            o.zero_grad()
            loss(ps).backward()
            o.all_reduce_mean()
        ok = ok and torch.allclose(oa.grad, ob.grad, rtol=1e-6, atol=0)  # (ring order may differ per range: <= 1 ulp)
    bk = ob._buckets
    ok = ok and len(bk) >= 2 and bk[0]['lo'] == 0 and bk[-1]['hi'] == ob.flat.numel() and \
        all(bk[i]['hi'] == bk[i + 1]['lo'] for i in range(len(bk) - 1)) and not oa._buckets
    # a parameter that receives no gradient this step: its bucket falls back to the synchronous call
    pb[2].requires_grad_(False)
    pa[2].requires_grad_(False)
    for ps, o in ((pa, oa), (pb, ob)):
        o.zero_grad()
        loss(ps).backward()
        o.all_reduce_mean()
    ok = ok and torch.allclose(oa.grad, ob.grad, rtol=1e-6, atol=0) and float(ob.grad.abs().sum()) > 0

    # The HOOK-DRIVEN path proper: gradients written straight into slices of the flat gradient buffer (ops._grad_destination) and handed
    # to autograd as views of it -- what the whole-network backward calls do on the GPU.  With plain autograd gradients (above) every
    # bucket takes the synchronous fallback; here every bucket's collective must have been issued from inside backward.
    from neuroclear_amd import ops

    class Direct(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *params):
            ctx.packed = ops._pack_params(params)
            ctx.shapes = [tuple(q.shape) for q in params]
            return sum(((q * (rank + 1 + i)) ** 2).sum() for i, q in enumerate(params))

        @staticmethod
        def backward(ctx, dy):
            dpar = ops._grad_destination(ctx.packed)
            off, views = 0, []
            for i, shp in enumerate(ctx.shapes):
                n = int(np.prod(shp))
                dpar[off:off + n].copy_((2.0 * (rank + 1 + i) ** 2 * ctx.packed[off:off + n]) * dy)
                views.append(dpar[off:off + n].view(shp))
                off += n
            return tuple(views)

    pc, oc = build(True)
    pd, od = build(False)
    for it in range(2):
        oc.zero_grad()
        Direct.apply(*pc).backward()
        ok = ok and all(b['work'] is not None for b in oc._buckets) and len(oc._buckets) >= 2  # issued by the hooks, all of them
        ok = ok and all(q.grad.data_ptr() == oc.grad.data_ptr() + 4 * o for q, o in zip(pc, np.cumsum([0] + [q.numel() for q in pc])[:-1]))
        oc.all_reduce_mean()
        od.zero_grad()
        loss(pd).backward()
        od.all_reduce_mean()
        ok = ok and torch.allclose(oc.grad, od.grad, rtol=1e-6, atol=0)  # (the Adam launch itself needs the GPU: tests/test_gpu_ops.py)
    # a second backward into the same optimizer before the exchange was collected must not be averaged silently wrong: it raises
    oc.zero_grad()
    Direct.apply(*pc).backward()
    try:
        loss(pc).backward()
        ok = False
    except RuntimeError as e:
        ok = ok and 'all-reduce was already issued' in str(e)
    for b in oc._buckets:  # drain what was issued, so that every rank leaves the group in step
        if b['work'] is not None:
            b['work'].wait()
    if rank == 0:
        np.save(out_path, np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_flat_adam_bucketed_async_all_reduce(tmp_path, world):
    """The bucketed, hook-driven exchange (issued while backward is still producing earlier buckets) must leave exactly
    the gradients the single synchronous all-reduce leaves."""
    port = _free_port()
    out = str(tmp_path / 'b.npy')
    mp.spawn(_adam_overlap_worker, args=(world, port, out), nprocs=world, join=True)
    assert np.load(out)[0] == 1


def test_default_assemble_mode():
    """`torchrun ... test_dice --skip_real --normalize_intensity` (assemble=None) must run: the default of world > 1 is 'slab' only when
    nothing needs the whole volume on one rank (the sharded run itself: tests/test_gpu_dist.py)."""
    from neuroclear_amd.test_dice import default_assemble
    assert default_assemble(1) == 'gather' and default_assemble(1, normalize_intensity=True) == 'gather'
    assert default_assemble(8) == 'slab'
    assert default_assemble(8, normalize_intensity=True) == 'reduce' and default_assemble(2, with_real=True) == 'reduce'


def _p2p_check_worker(rank, world, port, out_path, inject):
    _init(rank, world, port)
    if inject == 'env':
        os.environ['NC_TEST_FAIL_P2P'] = '1'
    from neuroclear_amd import test_dice as td
    dev = torch.device('cpu')
    mode = td.resolve_assemble(None, rank, world, dev)
    again = td.resolve_assemble('slab', rank, world, dev)  # remembered per (world, device): no second ring
    explicit_reduce = td.resolve_assemble('reduce', rank, world, dev)
    got = [None] * world
    dist.all_gather_object(got, (mode, again, explicit_reduce))
    if rank == 0:
        np.save(out_path, np.array([[m == 'slab', a == 'slab', e == 'reduce'] for m, a, e in got], dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('inject,expect_slab', [('none', 1), ('env', 0)])
def test_p2p_selfcheck_falls_back_to_reduce_on_every_rank(tmp_path, inject, expect_slab):
    """the 1 KiB batch_isend_irecv ring in front of the first sharded run.  Working transport: 'slab' stays.  A transport
    that raises sends EVERY rank to assemble='reduce' (collectives only, the path tests/test_gpu_rccl.py has proven on RCCL; its schedule is
    test_sharded_dice_loop_reduce above).  (A transport that fails on ONE side only leaves the peer waiting: the ring's waits carry a timeout,
    NC_P2P_CHECK_TIMEOUT, after which that rank raises -- gloo closes the pair then, so no common verdict is possible; bench.py's guard turns
    that into a printed train line.)"""
    out = str(tmp_path / 'p.npy')
    mp.spawn(_p2p_check_worker, args=(2, _free_port(), out, inject), nprocs=2, join=True)
    r = np.load(out)
    assert r.shape == (2, 3)
    assert (r[:, 0] == expect_slab).all() and (r[:, 1] == expect_slab).all() and (r[:, 2] == 1).all(), r


def test_nc_assemble_override(monkeypatch):
    from neuroclear_amd.test_dice import default_assemble
    monkeypatch.setenv('NC_ASSEMBLE', 'reduce')
    assert default_assemble(8) == 'reduce' and default_assemble(1) == 'gather'
    monkeypatch.setenv('NC_ASSEMBLE', 'gather')
    assert default_assemble(2) == 'gather'
    monkeypatch.setenv('NC_ASSEMBLE', 'slab')
    assert default_assemble(8) == 'slab' and default_assemble(8, normalize_intensity=True) == 'reduce'
    monkeypatch.setenv('NC_ASSEMBLE', 'bogus')
    with pytest.raises(ValueError):
        default_assemble(8)
