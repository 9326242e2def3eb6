"""Probe: which torch.distributed ops accept CUDA tensors on the gloo backend of this build (2 ranks on one GPU)."""
import os, sys, socket, traceback
import torch, torch.distributed as dist, torch.multiprocessing as mp

def w(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    def t(name, fn):
        try:
            fn(); torch.cuda.synchronize()
            if rank == 0: print(name, 'OK', flush=True)
        except Exception as e:
            if rank == 0: print(name, 'FAIL', repr(e)[:200], flush=True)
    x = torch.full((1000,), float(rank + 1), device=dev)
    t('broadcast', lambda: dist.broadcast(x, 0))
    t('all_reduce', lambda: dist.all_reduce(x))
    t('all_reduce_async', lambda: dist.all_reduce(x, async_op=True).wait())
    t('reduce', lambda: dist.reduce(x, 0))
    def g():
        out = [torch.empty_like(x) for _ in range(world)] if rank == 0 else None
        dist.gather(x, out, dst=0)
    t('gather', g)
    def sr():
        if rank == 1: dist.send(x, 0)
        else:
            b = torch.empty_like(x); dist.recv(b, 1)
    t('send_recv', sr)
    def b():
        ops = []
        y = torch.empty_like(x)
        ops.append(dist.P2POp(dist.isend, x, 1 - rank)); ops.append(dist.P2POp(dist.irecv, y, 1 - rank))
        for k in dist.batch_isend_irecv(ops): k.wait()
    t('batch_p2p', b)
    u = torch.arange(1000, dtype=torch.int16, device=dev)
    def sr16():
        if rank == 1: dist.send(u, 0)
        else:
            b = torch.empty_like(u); dist.recv(b, 1); assert torch.equal(b, u)
    t('send_recv_i16', sr16)
    dist.barrier(); dist.destroy_process_group()

if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(w, args=(2, port), nprocs=2, join=True)
