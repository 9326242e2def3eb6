"""Diced-volume inference on MI355X (reference: test_dice.py:50-157 -- the loop `for data in dataset: set_input; test;
addToStack` followed by `assemble_all`).

`diced_inference` is the loop itself, sharded over ranks the way SURVEY.md 8(e) describes: cube i belongs to rank
i % world (cubes are independent units; no data-path collective is needed to COMPUTE them).  Around the loop:
  * the weights are replicated by ONE RCCL broadcast of rank 0's packed parameter blob (28.3 MB for unet_deconv) --
    the reference's replication point is nn.DataParallel at models/networks.py:132-136;
  * assemble='reduce' (default for world > 1 when the whole volume is needed on one rank: --normalize_intensity, with_real): every rank overlap-adds its own cubes into its own padded fp32
    accumulator on its own GPU, then one RCCL reduce(sum) to rank 0 -- no per-round synchronisation, no tile traffic.
    fp32 addition order then differs from the reference's sequential index order (util/assemble_dice.py:167-173) by
    <= 1 ulp per voxel, which can flip the truncating integer cast by 1 LSB (tolerance +-1 LSB);
  * assemble='slab' (default for world > 1): rank r computes a CONTIGUOUS range of cubes (z-major numbering: two or three z-layers)
    into an accumulator that holds only those planes; one point-to-point exchange hands every rank the contributions to the z-slab
    it OWNS (1/world of the padded volume), which it finalises (/ count, x 65535, truncating cast) itself; the uint16 slabs are
    gathered on rank 0.  No rank holds or receives a whole fp32 volume (assemble='reduce' moved 7 x 3.5 GB into rank 0 for 900^3
    on 8 GPUs; this moves ~1 GB between neighbours and 1.46 GB of uint16 to rank 0).  Same +-1 LSB note as 'reduce';
  * assemble='gather': lock-step rounds, each round's tiles gathered to rank 0 and overlap-added there in index order
    -- bit-identical to the single-GPU / reference result; the verification mode.
`main()` keeps the reference's command line for the flags that matter on this path."""
import os

import numpy as np
import torch

from .data.diceImage_dataset import DiceImageDataSet
from .util.assemble_dice import Assemble_Dice, match_cube


_SIDE = {}


def _side_streams(device, n):
    """The HIP streams of the cubes in flight, created once per device: scratch buffers are cached per stream
    (ops.workspace), so fresh streams per call would mean fresh multi-GB workspaces per call."""
    key = (device.index, n)
    if key not in _SIDE:
        _SIDE[key] = [torch.cuda.Stream(device) for _ in range(n)]
    return _SIDE[key]


def sharded_cube_loop(n, rank, world, produce, consume, empty_like):
    """assemble='gather' schedule, free of device code so that it can be exercised with gloo on CPU:
    round t hands cube t*world + r to rank r; `produce(i)` returns that cube's network output (a tensor), tiles of a
    round are gathered to rank 0, which calls `consume(j, tile)` in increasing j -- the reference's summation order
    (util/assemble_dice.py:167-173).  `empty_like()` makes the placeholder a rank sends when it has no cube left."""
    import torch.distributed as dist
    rounds = (n + world - 1) // world
    for t in range(rounds):
        i = t * world + rank
        tile = produce(i) if i < n else empty_like()
        if world == 1:
            consume(i, tile)
            continue
        tiles = [torch.empty_like(tile) for _ in range(world)] if rank == 0 else None
        dist.gather(tile, tiles, dst=0)
        if rank == 0:
            for r in range(world):
                j = t * world + r
                if j < n:
                    consume(j, tiles[r])


def sharded_cube_loop_reduce(n, rank, world, produce, add_local, accumulator):
    """assemble='reduce' schedule (device-free, gloo-testable): rank r runs cubes r, r + world, ... back to back with no
    synchronisation, `add_local(i, tile)` overlap-adds into the rank's own accumulator, and ONE reduce(sum) of
    `accumulator()` lands the volume on rank 0."""
    import torch.distributed as dist
    for i in range(rank, n, world):
        add_local(i, produce(i))
    if world > 1:
        dist.reduce(accumulator(), dst=0, op=dist.ReduceOp.SUM)


def slab_plan(steps, step, roi, P0, world):
    """assemble='slab': who computes which cubes, which planes that leaves in each rank's local accumulator, and which planes of the
    padded volume each rank OWNS (finalises and ships).  Cubes are numbered z-major, so rank r's contiguous share
    [n r / world, n (r + 1) / world) covers two or three z-layers of cubes (<= 1 cube of imbalance; 729 cubes on 8 ranks: 91 or 92
    each); the owner slabs cut the padded z axis into `world` equal ranges.  Pure arithmetic: every rank computes the same plan."""
    nz, ny, nx = steps
    n = nz * ny * nx
    cubes = [(r * n // world, (r + 1) * n // world) for r in range(world)]
    local = []
    for a, b in cubes:
        local.append((0, 0) if a == b else ((a // (ny * nx)) * step, ((b - 1) // (ny * nx)) * step + roi))
    own = [(r * P0 // world, (r + 1) * P0 // world) for r in range(world)]
    return dict(cubes=cubes, local=local, own=own)


def slab_exchange(rank, world, plan, local_acc, own_acc):
    """The one exchange step of assemble='slab' (device-free: gloo on CPU, RCCL over xGMI on the GPUs): every rank sends each owner
    the planes of its local accumulator that fall into the owner's slab -- point-to-point pieces to at most three neighbours, nothing to
    a root -- and adds what it receives into `own_acc` in ascending source-rank order (its own share in its turn: deterministic).
    local_acc [local planes, P1, P2] starts at plane plan['local'][rank][0]; own_acc [owned planes, P1, P2] must be zero."""
    import torch.distributed as dist
    from .util.dist import p2p_fence
    p2p_fence(local_acc)

    def overlap(a, b):
        lo, hi = max(a[0], b[0]), min(a[1], b[1])
        return (lo, hi) if lo < hi else None
    ops, recv = [], {}
    for src in range(world):
        for dst in range(world):
            ov = overlap(plan['local'][src], plan['own'][dst])
            if ov is None or src == dst:
                continue
            if rank == src:
                piece = local_acc[ov[0] - plan['local'][src][0]:ov[1] - plan['local'][src][0]].contiguous()
                ops.append(dist.P2POp(dist.isend, piece, dst))
            elif rank == dst:
                buf = torch.empty((ov[1] - ov[0],) + tuple(own_acc.shape[1:]), dtype=own_acc.dtype, device=own_acc.device)
                recv[src] = (ov, buf)
                ops.append(dist.P2POp(dist.irecv, buf, src))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        p2p_fence(own_acc)
    o0 = plan['own'][rank][0]
    for src in range(world):
        if src == rank:
            ov = overlap(plan['local'][rank], plan['own'][rank])
            if ov is not None:
                own_acc[ov[0] - o0:ov[1] - o0] += local_acc[ov[0] - plan['local'][rank][0]:ov[1] - plan['local'][rank][0]]
        elif src in recv:
            ov, buf = recv[src]
            own_acc[ov[0] - o0:ov[1] - o0] += buf


def slab_gather(rank, world, plan, out_slab, L0, dst=0):
    """The finalised integer slabs (planes own & [0, L0)) travel to rank `dst`: 2 bytes per voxel instead of the fp32 accumulators of
    assemble='reduce'.  Returns the list of slabs in rank order on `dst`, None elsewhere.
    Wire format: BYTES.  ProcessGroupNCCL has no 16-bit integer type (torch/csrc/distributed/c10d/NCCLUtils.hpp: Char, Byte, Int, Long,
    Half, Float, Double, Bool, BFloat16, Float8*), so an int16 slab cannot cross RCCL as it is; every rank contributes ONE uint8 buffer of
    the largest slab's size (the slabs differ by the clipped tail of the last rank only) to ONE gather -- a collective, so the same call
    runs through RCCL in a world of one (tests/test_gpu_rccl.py) and the root posts no per-peer receives."""
    import torch.distributed as dist
    from .util.dist import p2p_fence
    out_slab = out_slab.contiguous()
    sizes = [max(0, min(b, L0) - min(a, L0)) for a, b in plan['own']]
    if not (dist.is_available() and dist.is_initialized()):
        return [out_slab] if rank == dst else None
    plane = int(np.prod(out_slab.shape[1:])) * out_slab.element_size()  # bytes of one finalised plane
    nbytes = max(sizes) * plane
    wire = torch.zeros(nbytes, dtype=torch.uint8, device=out_slab.device)
    wire[:sizes[rank] * plane] = out_slab.view(torch.uint8).reshape(-1)
    p2p_fence(wire)
    bufs = [torch.empty_like(wire) for _ in range(world)] if rank == dst else None
    dist.gather(wire, bufs, dst=dst)
    if rank != dst:
        return None
    tail = tuple(out_slab.shape[1:])
    return [out_slab if r == dst else bufs[r][:sizes[r] * plane].view(out_slab.dtype).reshape((sizes[r],) + tail)
            for r in range(world) if sizes[r] or r == dst]


def broadcast_parameters(net, src=0):
    """One broadcast of the packed parameter blob (state-dict order) from rank `src`; every rank then holds identical
    weights.  A no-op without an initialised process group."""
    import torch.distributed as dist
    from .util.dist import exchange_active
    if not exchange_active():
        return
    params = list(net.parameters())
    blob = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(blob, src)
    off = 0
    with torch.no_grad():
        for p in params:
            k = p.numel()
            p.copy_(blob[off:off + k].view_as(p))
            off += k


def default_assemble(world, with_real=False, normalize_intensity=False):
    """The assembly mode diced_inference picks when the caller names none: in-order on one rank; 'slab' on several -- unless something
    needs the WHOLE volume on one rank (the percentiles of --normalize_intensity, the second visual of with_real): 'slab' finalises per
    rank, so those take the one-reduce form.  Only an explicit assemble='slab' together with them raises.
    NC_ASSEMBLE=slab|reduce|gather overrides the default of world > 1: 'reduce' and 'gather' use collectives only."""
    if world <= 1:
        return 'gather'
    env = os.environ.get('NC_ASSEMBLE', '')
    if env in ('reduce', 'gather'):
        return env
    if env not in ('', 'slab'):
        raise ValueError("NC_ASSEMBLE must be 'slab', 'reduce' or 'gather'")
    return 'reduce' if (with_real or normalize_intensity) else 'slab'


_P2P_OK = {}
LAST = {'assemble': None}  # the mode the last diced_inference() of this process took (bench.py reports it)


def p2p_selfcheck(rank, world, device):
    """assemble='slab' is the one mode whose transport is point to point (`dist.batch_isend_irecv`, slab_exchange): every other exchange
    of this package is a collective.  Before the first cube of the first sharded run, a 1 KiB ring among the ranks through the same call
    (rank r sends to r + 1, receives from r - 1) says whether that transport works in THIS process group; the verdict is made common by
    an all_reduce(MIN) -- a collective -- so that every rank takes the same branch, and is remembered per (world, device).  False: the
    caller takes assemble='reduce' (collectives only; +-1 LSB like 'slab').  NC_TEST_FAIL_P2P=1 injects a failure (tests)."""
    import sys
    import torch.distributed as dist
    from .util.dist import p2p_fence
    key = (world, str(device))
    if key in _P2P_OK:
        return _P2P_OK[key]
    ok, why = 1, ''
    try:
        if os.environ.get('NC_TEST_FAIL_P2P') == '1':
            raise RuntimeError('injected point-to-point failure (NC_TEST_FAIL_P2P=1)')
        send = torch.full((256,), float(rank), dtype=torch.float32, device=device)
        recv = torch.full((256,), -1.0, dtype=torch.float32, device=device)
        p2p_fence(send)
        ops = [dist.P2POp(dist.isend, send, (rank + 1) % world), dist.P2POp(dist.irecv, recv, (rank - 1) % world)]
        import datetime
        tmo = datetime.timedelta(seconds=float(os.environ.get('NC_P2P_CHECK_TIMEOUT', '30')))
        for w in dist.batch_isend_irecv(ops):
            w.wait(tmo)  # (a peer whose transport raised never posts its half: do not wait for it for ever)
        p2p_fence(recv)
        if not bool((recv == float((rank - 1) % world)).all()):
            ok, why = 0, 'the ring delivered wrong bytes'
    except Exception as e:  # noqa: BLE001 -- whatever the backend raises: the answer is "do not use it"
        ok, why = 0, repr(e)
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    good = bool(int(flag.item()))
    if not good and (why or rank == 0):
        print("neuroclear_amd.test_dice: point-to-point self-check failed on rank %d (%s): assemble='slab' -> 'reduce'"
              % (rank, why or 'another rank reported the failure'), file=sys.stderr, flush=True)
    _P2P_OK[key] = good
    return good


def resolve_assemble(assemble, rank, world, device, with_real=False, normalize_intensity=False):
    """The mode a sharded run really takes: the caller's (or the default / NC_ASSEMBLE), with 'slab' replaced by 'reduce' when the
    point-to-point self-check fails.  Same answer on every rank."""
    import torch.distributed as dist
    if assemble is None:
        assemble = default_assemble(world, with_real, normalize_intensity)
    if assemble == 'slab' and world > 1 and dist.is_available() and dist.is_initialized() and not p2p_selfcheck(rank, world, device):
        assemble = 'reduce'
    return assemble


def diced_inference(netG, volume, opt, rank=0, world=1, max_cubes=None, assemble=None, broadcast=True, on_cube=None,
                    with_real=False):
    """volume: uint8/uint16 ndarray (original size).  Returns the assembled uint8/uint16 ndarray on rank 0.
    on_cube(fn) -> result: optional wrapper around each cube's network call (bench.py brackets it with HIP events).
    with_real: also assemble the input cubes (the reference's 'real' visual, test_dice.py without --skip_real) and return
    (fake, real); the dice -> assemble round trip of the input is the input up to 1 LSB of the truncating cast."""
    if assemble is None:
        assemble = default_assemble(world, with_real, getattr(opt, 'normalize_intensity', False))
    if assemble not in ('reduce', 'gather', 'slab'):
        raise ValueError("assemble must be 'slab', 'reduce' or 'gather'")
    if assemble == 'slab' and (with_real or getattr(opt, 'normalize_intensity', False)):
        raise NotImplementedError("assemble='slab' finalises per rank: use assemble='reduce' with --normalize_intensity (percentiles of the "
                                  "whole volume) or with_real")
    if broadcast:
        broadcast_parameters(netG, 0)  # (a no-op without a process group of > 1 rank: util.dist.exchange_active)
    ds = DiceImageDataSet(opt, volume=volume)
    assemble = resolve_assemble(assemble, rank, world, ds.device)  # 'slab' -> 'reduce' when point to point does not work in this group
    LAST['assemble'] = assemble
    n = len(ds) if max_cubes is None else min(len(ds), max_cubes)
    local_acc = assemble in ('reduce', 'slab') or rank == 0
    if with_real and assemble == 'gather' and world > 1:
        raise NotImplementedError("with_real needs assemble='reduce' when sharded")
    plan = None
    if assemble == 'slab':
        from .util import util as U
        size0 = ds.size_original()
        padded = U.padded_shape(size0, opt.dice_size[0], opt.overlap)
        steps = U.grid_steps(padded, opt.dice_size[0], opt.overlap)
        plan = slab_plan(steps, opt.dice_size[0] - opt.overlap, opt.dice_size[0], padded[0], world)
        if max_cubes is not None:  # warm-up runs: the first cubes of every rank's own range
            plan['cubes'] = [(a, min(b, a + max(1, max_cubes // world))) for a, b in plan['cubes']]
    if with_real:
        import copy
        opt = copy.copy(opt)
        opt.skip_real = False
    asm = Assemble_Dice(opt, ds.size_original(), slab=plan['local'][rank] if plan else None) if local_acc else None
    E = opt.dice_size[0] + 2 * opt.border_cut
    hm = bool(getattr(opt, 'histogram_match', False))  # the producing rank matches its own cube (it holds the input)

    # Cubes in flight (NC_INFER_STREAMS streams, default 2; round 6: each carries a BATCH of cubes, below): a batch is cut and run through the network on its stream, the
    # overlap-adds stay on the calling stream in cube order (each waits for its cube's event) -- so the result is the one-stream
    # result bit for bit, while the second stream's kernels fill the CUs the first one's kernels leave idle at their tails (the
    # persistent convolution kernels own whole CUs; a 35^3 layer fills 256 CUs 1.64 times).  Measured at 480^3: 1 stream 2.86 s,
    # 2: 2.87, 3: 2.69, 4: 2.70, 6: 3.61 (the working sets of six cubes no longer share the caches).  Not in the lock-step gather
    # rounds of world > 1 (the tiles go straight into a collective there).
    nstreams = int(os.environ.get('NC_INFER_STREAMS', '2'))
    piped = ds.device.type == 'cuda' and nstreams > 1 and (world == 1 or assemble in ('reduce', 'slab'))
    main = torch.cuda.current_stream(ds.device) if piped else None
    side = _side_streams(ds.device, nstreams) if piped else []
    for st in side:
        st.wait_stream(main)  # the volume upload and the parameter broadcast were enqueued on the calling stream
    issued = [0]

    # Round 6: NC_INFER_BATCH (default 5) cubes per network call on NC_INFER_STREAMS (default 2) streams.  The persistent 256-workgroup
    # convolution launches of the 70^3 / 35^3 levels hold 1.64-6.6 rounds of tiles for one 140^3 cube: batched launches fill their last
    # round (convolutions alone: -5 % at three cubes per call), and weight packs / statistics finalisations are shared by the batch.  In the
    # volume: 1.333 s (1 x 3) -> 1.308 (3 x 2) -> 1.295 (5 x 2) per 480^3, same-box alternation (profiles/r06_ab_infer_batch.txt).  A cube's result does not depend on its batch mates (InstanceNorm is per
    # sample) beyond fp32 rounding of the statistics' partial sums (their grouping follows the launch's tile plan): run-to-run bit-identical,
    # within 1 LSB of the one-cube-per-call order.  NC_INFER_BATCH=1: one cube per call (the reference's batch size, test_dice.py:66).
    # The batches follow the order in which THIS rank will ask for its cubes.
    nbatch = max(1, int(os.environ.get('NC_INFER_BATCH', '5')))
    if assemble == 'gather' and world > 1:
        nbatch = 1  # the verification mode keeps the reference's arithmetic cube by cube: bit-identical to a one-cube-per-call single-rank run
    if assemble == 'slab':
        my_cubes = list(range(*plan['cubes'][rank]))
    elif assemble == 'reduce':
        my_cubes = list(range(rank, n, world))
    else:
        my_cubes = [t * world + rank for t in range((n + world - 1) // world) if t * world + rank < n]
    pos_of = {c: k for k, c in enumerate(my_cubes)}
    ready = {}

    def run_batch(idx):
        xs = [ds[i]['A'] for i in idx]
        x = torch.stack(xs, 0)
        y = netG(x) if on_cube is None else on_cube(lambda: netG(x))
        outs = []
        for k in range(len(idx)):
            yk = y[k].reshape(E, E, E)
            if hm:
                yk = match_cube(yk, xs[k].unsqueeze(0), opt.dice_size[0], opt.border_cut, ds.device)
            outs.append((xs[k].unsqueeze(0), yk))
        return outs

    def produce(i):
        if i not in ready:
            k = pos_of[i]
            idx = my_cubes[k:k + nbatch]
            if not piped:
                for c, (x, y) in zip(idx, run_batch(idx)):
                    ready[c] = (x, y, None)
            else:
                st = side[issued[0] % nstreams]
                issued[0] += 1
                with torch.cuda.stream(st):
                    outs = run_batch(idx)
                    ev = torch.cuda.Event()
                    ev.record(st)
                for c, (x, y) in zip(idx, outs):
                    y.record_stream(main)
                    x.record_stream(main)
                    ready[c] = (x, y, ev)
        x, y, ev = ready.pop(i)
        if not piped:
            if with_real:
                asm.add_cube('real', x.reshape(E, E, E), i)
            return y
        return (x, y, ev)

    def add_fake(j, tile):
        if piped:
            x, tile, ev = tile
            main.wait_event(ev)
            if with_real:
                asm.add_cube('real', x.reshape(E, E, E), j)
        asm.add_cube('fake', tile, j)

    with torch.no_grad():
        if assemble == 'slab':
            a, b = plan['cubes'][rank]
            for i in range(a, b):
                add_fake(i, produce(i))
            o0, o1 = plan['own'][rank]
            own = torch.zeros((o1 - o0,) + tuple(asm.image_size[1:]), dtype=torch.float32, device=ds.device)
            slab_exchange(rank, world, plan, asm.acc['fake'], own)
            out_slab = asm.finalize_slab(own, o0, o1)
            parts = slab_gather(rank, world, plan, out_slab, asm.image_size_original[0])
            if rank != 0:
                return None
            host = torch.cat(parts, 0).cpu().numpy()
            return host.view(np.uint16) if asm.imtype == 'uint16' else host
        if assemble == 'reduce':
            sharded_cube_loop_reduce(n, rank, world, produce, add_local=add_fake, accumulator=lambda: asm.acc['fake'])
            if with_real and world > 1:
                torch.distributed.reduce(asm.acc['real'], dst=0, op=torch.distributed.ReduceOp.SUM)
        else:
            sharded_cube_loop(
                n, rank, world,
                produce=produce,
                consume=add_fake,
                empty_like=lambda: torch.zeros((E, E, E), dtype=torch.float32, device=ds.device))
    if rank != 0:
        return None
    for k in asm.count:
        asm.count[k] = asm.len_cube_queue  # warm-up runs (max_cubes) assemble a partial volume on purpose
    asm.assemble_all()
    return (asm.getDict()['fake'], asm.getDict()['real']) if with_real else asm.getDict()['fake']


def write_outputs(opt, web_dir, fake_volume, real_volume=None, gt_volume=None):
    """The tail of the reference's test script (test_dice.py:126-270): the assembled volume as a multi-page TIFF, maximum
    intensity projections, per-slice TIFFs along the three axes, and the PSNR report against a ground-truth volume --
    same directory layout and file names.  Returns the metrics dict (empty without ground truth)."""
    from .util import tiff
    from .util import util as U
    skip_real = real_volume is None
    if getattr(opt, 'save_volume', False):
        U.mkdir(web_dir + '/volumes')
        tag = 'iter-' + str(opt.load_iter) if getattr(opt, 'load_iter', 0) > 0 else 'epoch-' + str(opt.epoch)
        tiff.imsave(web_dir + '/volumes/output_volume_xy-view_' + tag + '.tif', fake_volume)
        print('Output volume is saved as a tiff file. ')
        if not skip_real:
            tiff.imsave(web_dir + '/volumes/input_volume_xy-view.tif', real_volume)
            print('Input volume is saved as a tiff file. ')
    if getattr(opt, 'save_projections', False):
        U.mkdir(web_dir + '/projections')
        ep = str(opt.epoch)
        # (the reference projects fixed sub-ranges of the output volume along y and x, test_dice.py:163-164)
        U.save_image(np.amax(fake_volume, axis=0), web_dir + '/projections/fake_xy_proj_epoch-' + ep + '.tif')
        U.save_image(np.amax(fake_volume[:, 800:1100, :], axis=1, initial=0), web_dir + '/projections/fake_xz_proj_epoch-' + ep + '.tif')
        U.save_image(np.amax(fake_volume[:, :, 200:500], axis=2, initial=0), web_dir + '/projections/fake_yz_proj_epoch-' + ep + '.tif')
        if not skip_real:
            for ax, nm in ((0, 'xy'), (1, 'xz'), (2, 'yz')):
                U.save_image(np.amax(real_volume, axis=ax), web_dir + '/projections/real_%s_proj.tif' % nm)
    if getattr(opt, 'save_slices', False):
        for nm in ('xy', 'yz', 'xz'):
            U.mkdir(web_dir + '/images/output_' + nm)
            if not skip_real:
                U.mkdir(web_dir + '/images/input_' + nm)
        for i in range(fake_volume.shape[2]):
            U.save_image(fake_volume[:, :, i], web_dir + '/images/output_yz/output_yz_%d.tif' % i)
            if not skip_real:
                U.save_image(real_volume[:, :, i], web_dir + '/images/input_yz/input_yz_%d.tif' % i)
        for i in range(fake_volume.shape[1]):
            U.save_image(fake_volume[:, i, :], web_dir + '/images/output_xz/output_xz_%d.tif' % i)
            if not skip_real:
                U.save_image(real_volume[:, i, :], web_dir + '/images/input_xz/input_xz_%d.tif' % i)
        for i in range(fake_volume.shape[0]):
            U.save_image(fake_volume[i], web_dir + '/images/output_xy/output_xy_%d.tif' % i)
            if not skip_real:
                U.save_image(real_volume[i], web_dir + '/images/input_xy/input_xy_%d.tif' % i)
    metrics = {}
    if gt_volume is not None:
        if skip_real:
            raise ValueError('the PSNR report compares input and output with the ground truth: do not pass --skip_real')
        print('Calculating PSNR for the whole image volume...')
        vols = {}
        for k, v in (('real', real_volume), ('fake', fake_volume), ('gt', gt_volume)):
            for _ in range(2):  # standardize + 8-bit normalize, applied twice as at test_dice.py:244-251
                v = U.normalize(U.standardize(v), data_type=np.uint8)
            vols[k] = v
        metrics['psnr_input_gt'] = U.get_psnr(vols['real'], vols['gt'], 2 ** 8 - 1)
        metrics['psnr_output_gt'] = U.get_psnr(vols['fake'], vols['gt'], 2 ** 8 - 1)
        bar = '---------------------------------------------------------'
        message = '\n'.join(['Experiment Name: ' + opt.name, bar, '', 'Whole_volume', bar, 'Network Input vs. Groundtruth',
                             '(psnr: %.4f) ' % metrics['psnr_input_gt'], bar, 'Network Output vs. Groundtruth',
                             '(psnr: %.4f) ' % metrics['psnr_output_gt'], bar])
        print(message)
        U.mkdir(web_dir)
        with open(os.path.join(web_dir, 'metrics.txt'), 'a') as f:
            f.write('%s\n' % message)
    return metrics


def main(argv=None):
    import argparse
    import os
    from . import models
    p = argparse.ArgumentParser(description='test_dice.py on MI355X (reference README.md:150-157)')
    p.add_argument('--dataroot', required=True)
    p.add_argument('--name', default='experiment_name')
    p.add_argument('--checkpoints_dir', default='./checkpoints')
    p.add_argument('--results_dir', default='./results/')
    p.add_argument('--gpu_ids', default='0')
    p.add_argument('--model', default='test')
    p.add_argument('--model_suffix', default='')
    p.add_argument('--netG', default='unet_deconv')
    p.add_argument('--norm', default='instance')
    p.add_argument('--init_type', default='normal')
    p.add_argument('--init_gain', type=float, default=0.02)
    p.add_argument('--input_nc', type=int, default=1)
    p.add_argument('--output_nc', type=int, default=1)
    p.add_argument('--ngf', type=int, default=64)
    p.add_argument('--image_dimension', type=int, default=3)
    p.add_argument('--dice_size', type=int, nargs='+', default=[120, 120, 120])
    p.add_argument('--overlap', type=int, default=15)
    p.add_argument('--border_cut', type=int, default=10)
    p.add_argument('--data_type', default='uint16')
    p.add_argument('--epoch', default='latest')
    p.add_argument('--load_iter', type=int, default=0)
    p.add_argument('--skip_real', action='store_true')
    p.add_argument('--no_dropout', action='store_true')
    p.add_argument('--histogram_match', action='store_true')
    p.add_argument('--normalize_intensity', action='store_true')
    p.add_argument('--sat_level', type=float, nargs='+', default=[0.25, 99.75])
    p.add_argument('--verbose', action='store_true')
    p.add_argument('--phase', default='test')
    p.add_argument('--data_name', default=None)
    p.add_argument('--dataroot_gt', default=None, help='directory with the ground-truth volume: enables the PSNR report')
    p.add_argument('--save_volume', action='store_true')
    p.add_argument('--save_projections', action='store_true')
    p.add_argument('--save_slices', action='store_true')
    opt = p.parse_args(argv)
    opt.gpu_ids = [int(g) for g in opt.gpu_ids.split(',') if int(g) >= 0]
    opt.isTrain, opt.continue_train, opt.preprocess = False, False, 'addColorChannel'
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        from .util.dist import init_process_group
        opt.gpu_ids = [int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()]
        torch.cuda.set_device(opt.gpu_ids[0])
        init_process_group(torch.device('cuda', opt.gpu_ids[0]))
    model = models.create_model(opt)
    model.setup(opt)
    from .data.diceImage_dataset import _load_volume
    names = sorted(f for f in os.listdir(opt.dataroot) if f.endswith(('.npy', '.tif', '.tiff')))
    vol = _load_volume(os.path.join(opt.dataroot, names[0]))
    gt = None
    if opt.dataroot_gt is not None:
        gnames = sorted(f for f in os.listdir(opt.dataroot_gt) if f.endswith(('.npy', '.tif', '.tiff')))
        gt = _load_volume(os.path.join(opt.dataroot_gt, gnames[0]))
    want_real = not opt.skip_real
    res = diced_inference(model.netG, vol, opt, rank, world, with_real=want_real)
    metrics = {}
    if rank == 0:
        out, real = res if want_real else (res, None)
        print('Image volume re-assembled.')
        print('re-merged image shape: {}'.format(out.shape))
        # web_dir as at test_dice.py:80-88
        base = opt.name if opt.data_name is None else opt.data_name + '_by_' + opt.name
        web_dir = os.path.join(opt.results_dir, base, '{}_{}'.format(opt.phase, opt.epoch))
        if opt.load_iter > 0:
            web_dir = '{:s}_iter{:d}'.format(web_dir, opt.load_iter)
        os.makedirs(os.path.join(web_dir, 'volumes'), exist_ok=True)
        np.save(os.path.join(web_dir, 'volumes', 'output_volume.npy'), out)
        metrics = write_outputs(opt, web_dir, out, real, gt)
        print('----Test done----')
    return metrics


if __name__ == '__main__':
    main()
