"""train_onecube.py on MI355X (reference: train_onecube.py:35-110): the same loop -- random index, set_input,
optimize_parameters, periodic loss print / checkpoint, update_learning_rate -- with the visualiser left out.

    python -m neuroclear_amd.train_onecube --dataroot DIR --name NAME --model axial_to_lateral_gan_apollo \\
        --preprocess randomcrop_randomflip_addColorChannel_addBatchChannel --crop_size 108 108 108 --gan_mode lsgan \\
        --init_type kaiming --norm instance --lambda_A 5 --lambda_plane 1 1 1 --lr_policy constant \\
        --randomize_projection_depth --projection_depth 10 --save_by_iter --gpu_ids 0
Under torchrun (one rank per GPU) the gradients are all-reduced over RCCL before each optimizer step."""
import os
import time

import numpy as np
import torch

from .data.singlevolume_dataset import SingleVolumeDataset
from .models import create_model
from .options import TrainOptions


def main(argv=None):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    opt = TrainOptions().parse(argv)
    if world > 1:
        from .util.dist import init_process_group
        opt.gpu_ids = [int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()]
        torch.cuda.set_device(opt.gpu_ids[0])
        init_process_group(torch.device('cuda', opt.gpu_ids[0]))
    dataset = SingleVolumeDataset(opt)
    model = create_model(opt)
    model.setup(opt)
    if world > 1:
        for o in model.optimizers:
            torch.distributed.broadcast(o.flat, 0)
        from .util.dist import broadcast_buffers
        broadcast_buffers(model)  # (--norm batch: running statistics start equal; afterwards rank-local, rank 0's are saved -- util/dist.py)
    total_iters = opt.load_iter + 1 if opt.load_iter > 0 else 0
    loaded = total_iters
    iter_data_time = time.time()
    while True:
        data = dataset[np.random.randint(0, 10)]
        iter_start_time = time.time()
        t_data = iter_start_time - iter_data_time
        total_iters += opt.batch_size
        model.set_input(data)
        model.optimize_parameters()
        if total_iters % opt.print_freq == 0 and rank == 0:
            losses = model.get_current_losses()
            t_comp = (time.time() - iter_start_time) / opt.batch_size
            print('(iters: %d, time: %.3f, data: %.3f) ' % (total_iters, t_comp, t_data) +
                  ' '.join('%s: %.3f' % kv for kv in losses.items()), flush=True)
        if total_iters % opt.save_latest_freq == 0 and rank == 0:
            model.save_networks('iter_%d' % total_iters if opt.save_by_iter else 'latest')
        model.update_learning_rate()
        iter_data_time = time.time()
        if opt.max_iters and total_iters - loaded >= opt.max_iters:
            break
    return model


if __name__ == '__main__':
    main()
