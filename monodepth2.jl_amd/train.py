"""The reference's `train()` (src/Monodepth.jl:76-197) with its hard-coded paths turned into arguments: chain the datasets, build
TrainCache / Model / ADAM, then epochs of shuffled batches -> [automasking loss] -> train_loss + gradient + update!, a disparity heat map
and the warped sources every `log_iter` iterations, a checkpoint every `save_iter`.  Data parallel when torch.distributed is initialised
(one process per GPU, `parallel.DataParallel`: every rank reads its own shard of each global batch).

Not on the measured path (bench.py feeds synthetic HBM-resident batches); the H2D copy of a batch is asynchronous from the loader's
pinned buffer, the loss is read back once per iteration like the reference's `cpu(loss)` (src/Monodepth.jl:174)."""
import os

import numpy as np

from . import data as D
from . import monodepth as md
from . import parallel as par
from . import synthetic as syn


def build_model(in_channels, lr=1e-4, seed=42, depth=18, precision="f32"):
    """src/Monodepth.jl:117-125 — ResNet trunk + depth decoder (levels 2:5) + pose decoder, Glorot weights."""
    enc = md.ResidualNetwork(depth, in_channels=in_channels)
    model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(enc.stages[-1]), lr=lr, precision=precision)
    model.load_params(syn.init_params(in_channels, seed=seed, depth=depth))
    return model


def train(datasets, log_dir, save_dir, batch_size=4, n_epochs=20, log_iter=50, save_iter=500, automasking=False,
          disparity_smoothness=1e-3, lr=1e-4, max_iters=None, seed=0, model=None, quiet=False):
    """Returns (model, list of per-iteration losses).  `datasets`: list of Depth10k / KittyDataset (same resolution and channels)."""
    import torch
    os.makedirs(log_dir, exist_ok=True); os.makedirs(save_dir, exist_ok=True)
    rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) if torch.distributed.is_initialized() else (0, 1)
    dchain = D.DChain(datasets)
    dataset = datasets[0]
    width, height = dataset.resolution
    in_channels = dchain[0].shape[1]
    parameters = md.Params(batch_size=batch_size, target_size=(width, height), disparity_smoothness=disparity_smoothness, automasking=automasking)
    max_scale, scale_levels = 5, [2, 3, 4, 5]
    scales = [1.0 / 2.0 ** (max_scale - level) for level in scale_levels]
    cache = md.TrainCache(md.SSIM(), md.Backproject(width, height), md.Project(width, height), dataset.K, dataset.invK,
                          dataset.target_id, list(dataset.source_ids), scales)
    if model is None:
        model = build_model(in_channels, lr=lr)
    model.lr = lr                                             # ADAM's eta lives in the model's handle configuration: `lr` wins over a passed-in model's
    optimizer = md.ADAM(lr)
    dp = None
    losses, it = [], 0
    for epoch in range(1, n_epochs + 1):
        # every rank draws the same permutation (same seed) and decodes only its own slice of each global batch
        loader = D.DataLoader(dchain, batch_size, shuffle=True, seed=seed + epoch, shard=(rank, world))
        for i, x_host in enumerate(loader, start=1):
            x = x_host.cuda(non_blocking=True)
            if world > 1 and dp is None:                      # first batch: the handle exists now -> attach the communicator, make the replicas identical
                dp = par.DataParallel(model, x_like=x, parameters=parameters, cache=cache)
                if rank == 0 and not quiet:                   # which gradient exchange this run uses (the library's own RCCL communicator is opt-in until a multi-GPU run has verified it)
                    print("data parallel over %d ranks: gradient exchange = %s" % (world, "libmd2hip RCCL communicator (bucketed, overlapped with the pullback)" if dp.comm == "rccl"
                                                                                   else "torch.distributed all-reduce of the library's flat gradient buffer"), flush=True)
            auto_loss = None
            if parameters.automasking:                      # computed outside AD (src/Monodepth.jl:159-164)
                auto_loss = md.automasking_loss(cache.ssim, x, x[:, cache.target_id - 1].contiguous(), cache.source_ids)
            do_visualization = (i % log_iter == 0 or i == 1) and rank == 0
            loss, disparity, warped, _ = md.train_loss(model, x, auto_loss, cache, parameters, do_visualization)
            md.gradient(model)
            if dp is not None:
                dp.allreduce_and_update()
            else:
                md.update(optimizer, model)
            loss_cpu = float(loss)
            losses.append(loss_cpu); it += 1
            if do_visualization:
                D.save_disparity(disparity[0, 0].numpy(), os.path.join(log_dir, "disp-%d-%d.png" % (epoch, i)))
                for sid, w in enumerate(warped, start=1):
                    D.save_warped(w[0].numpy(), os.path.join(log_dir, "warp-%d-%d-%d.png" % (epoch, i, sid)))
            if i % save_iter == 0 and rank == 0:
                md.save_checkpoint(model, os.path.join(save_dir, "%d-%d-%.6f.npz" % (epoch, i, loss_cpu)))
            if not quiet and rank == 0 and (i % log_iter == 0 or i == 1):
                print("epoch %d/%d  i %d/%d  loss %.6f" % (epoch, n_epochs, i, len(loader), loss_cpu), flush=True)
            if max_iters is not None and it >= max_iters:
                return model, losses
    return model, losses
