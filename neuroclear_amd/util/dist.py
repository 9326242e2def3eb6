"""Process-group plumbing of the N > 1 path: one process per GPU, torch.distributed over RCCL ('nccl' IS RCCL on ROCm).

The reference's multi-GPU mechanism is nn.DataParallel inside one process (models/networks.py:132-136); here every rank owns one
device and the only data exchanged is what SURVEY.md 8(e) lists (one weight broadcast, the slab exchange / integer gather of diced
inference, one gradient all-reduce per optimizer phase).

NC_DIST_BACKEND=gloo is the DRY RUN of the same schedules on a box with fewer GPUs than ranks (RCCL refuses two ranks on one
device): gloo moves CUDA tensors itself for the collectives, but its point-to-point send / recv read the buffer from the host side
without looking at the HIP stream that is still writing it -- `p2p_fence` closes that gap."""
import os

import torch


def backend_name():
    return os.environ.get('NC_DIST_BACKEND', 'nccl')


def init_process_group(device=None):
    """Join the job torch.distributed.run (or bench.py's own launcher) started.  Returns (rank, world).  A no-op for WORLD_SIZE 1."""
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1 or dist.is_initialized():
        return rank, world
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend_name() == 'nccl':
        dist.init_process_group('nccl', device_id=device)
    else:
        dist.init_process_group(backend_name())
    return rank, world


def exchange_active():
    """True when the data-parallel exchange steps (weight broadcast, gradient all-reduce, slab gather) have to run: an initialised
    process group of more than one rank -- or of ONE rank with NC_DIST_WORLD1=1, which sends the same calls through RCCL on a
    one-GPU box (every collective of a world of one is the identity, so results must equal the run without a group:
    tests/test_gpu_rccl.py).  That is the only way the RCCL side of this code can be exercised where RCCL refuses two ranks on one
    device."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('NC_DIST_WORLD1', '0') == '1'


def p2p_fence(t):
    """Call before handing a CUDA tensor to send / recv / batch_isend_irecv.  RCCL orders the transfer behind the current stream by
    itself; gloo does not (see the module docstring): drain the device first."""
    import torch.distributed as dist
    if t.is_cuda and dist.is_available() and dist.is_initialized() and dist.get_backend() == 'gloo':
        torch.cuda.synchronize(t.device)


def broadcast_buffers(model, src=0):
    """Rank `src`'s module buffers (BatchNorm running statistics under --norm batch, spectral-norm u / v) to every rank, once at setup.
    During training every rank updates them from its OWN crops (the reference's nn.DataParallel keeps replica 0's, computed over its share of
    the batch, models/networks.py:132-136 -- the same rule at a different shard size); rank 0's are the ones a checkpoint saves.  They do not
    enter the gradient exchange.  Integer buffers (num_batches_tracked: int64) travel as they are -- RCCL has Long."""
    import torch.distributed as dist
    if not exchange_active():
        return
    for name in getattr(model, 'model_names', []):
        net = getattr(model, 'net' + name, None)
        if net is None:
            continue
        for b in net.buffers():
            dist.broadcast(b, src)
