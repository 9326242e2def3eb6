"""Diced-volume inference on MI355X (reference: test_dice.py:50-157 -- the loop `for data in dataset: set_input; test;
addToStack` followed by `assemble_all`).

`diced_inference` is the loop itself, sharded over ranks the way SURVEY.md 8(e) describes: cube i belongs to rank
i % world (cubes are independent units; no data-path collective is needed to COMPUTE them).  Around the loop:
  * the weights are replicated by ONE RCCL broadcast of rank 0's packed parameter blob (28.3 MB for unet_deconv) --
    the reference's replication point is nn.DataParallel at models/networks.py:132-136;
  * assemble='reduce' (default for world > 1): every rank overlap-adds its own cubes into its own padded fp32
    accumulator on its own GPU, then one RCCL reduce(sum) to rank 0 -- no per-round synchronisation, no tile traffic.
    fp32 addition order then differs from the reference's sequential index order (util/assemble_dice.py:167-173) by
    <= 1 ulp per voxel, which can flip the truncating integer cast by 1 LSB (tolerance +-1 LSB);
  * assemble='gather': lock-step rounds, each round's tiles gathered to rank 0 and overlap-added there in index order
    -- bit-identical to the single-GPU / reference result; the verification mode.
`main()` keeps the reference's command line for the flags that matter on this path."""
import numpy as np
import torch

from .data.diceImage_dataset import DiceImageDataSet
from .util.assemble_dice import Assemble_Dice, match_cube


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


def broadcast_parameters(net, src=0):
    """One broadcast of the packed parameter blob (state-dict order) from rank `src`; every rank then holds identical
    weights.  A no-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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


def diced_inference(netG, volume, opt, rank=0, world=1, max_cubes=None, assemble=None, broadcast=True, on_cube=None):
    """volume: uint8/uint16 ndarray (original size).  Returns the assembled uint8/uint16 ndarray on rank 0.
    on_cube(fn) -> result: optional wrapper around each cube's network call (bench.py brackets it with HIP events)."""
    if assemble is None:
        assemble = 'reduce' if world > 1 else 'gather'
    if assemble not in ('reduce', 'gather'):
        raise ValueError("assemble must be 'reduce' or 'gather'")
    if broadcast and world > 1:
        broadcast_parameters(netG, 0)
    ds = DiceImageDataSet(opt, volume=volume)
    n = len(ds) if max_cubes is None else min(len(ds), max_cubes)
    local_acc = assemble == 'reduce' or rank == 0
    asm = Assemble_Dice(opt, ds.size_original()) if local_acc else None
    E = opt.dice_size[0] + 2 * opt.border_cut
    hm = bool(getattr(opt, 'histogram_match', False))  # the producing rank matches its own cube (it holds the input)

    def produce(i):
        x = ds[i]['A'].unsqueeze(0)
        y = (netG(x) if on_cube is None else on_cube(lambda: netG(x))).reshape(E, E, E)
        return match_cube(y, x, opt.dice_size[0], opt.border_cut, ds.device) if hm else y

    with torch.no_grad():
        if assemble == 'reduce':
            sharded_cube_loop_reduce(n, rank, world, produce, add_local=lambda j, tile: asm.add_cube('fake', tile, j),
                                     accumulator=lambda: asm.acc['fake'])
        else:
            sharded_cube_loop(
                n, rank, world,
                produce=produce,
                consume=lambda j, tile: asm.add_cube('fake', tile, j),
                empty_like=lambda: torch.zeros((E, E, E), dtype=torch.float32, device=ds.device))
    if rank != 0:
        return None
    asm.count['fake'] = asm.len_cube_queue  # warm-up runs (max_cubes) assemble a partial volume on purpose
    asm.assemble_all()
    return asm.getDict()['fake']


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
    opt = p.parse_args(argv)
    opt.gpu_ids = [int(g) for g in opt.gpu_ids.split(',') if int(g) >= 0]
    opt.isTrain, opt.continue_train, opt.preprocess = False, False, 'addColorChannel'
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        opt.gpu_ids = [int(os.environ.get('LOCAL_RANK', '0'))]
        torch.cuda.set_device(opt.gpu_ids[0])
        dist.init_process_group('nccl')
    model = models.create_model(opt)
    model.setup(opt)
    from .data.diceImage_dataset import _load_volume
    names = sorted(f for f in os.listdir(opt.dataroot) if f.endswith(('.npy', '.tif', '.tiff')))
    vol = _load_volume(os.path.join(opt.dataroot, names[0]))
    opt.skip_real = True
    out = diced_inference(model.netG, vol, opt, rank, world)
    if rank == 0:
        d = os.path.join(opt.results_dir, opt.name, 'volumes')
        os.makedirs(d, exist_ok=True)
        np.save(os.path.join(d, 'output_volume.npy'), out)
        print('re-merged image shape: {}'.format(out.shape))


if __name__ == '__main__':
    main()
