"""N>1 path on CPU: world_size-2 gloo.  Checks the exchange the DP step performs (one all-reduce(sum) of the flat
gradient buffer, then grad_scale = 1/world) against the mean of the per-shard oracle gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import md2amd


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "oracle")]
    import md2amd as m
    import oracle
    par = m.parallel
    r, w, _ = par.init_from_env("gloo")
    assert (r, w) == (rank, world)
    syn = m.synthetic
    C, H, W, GB = 1, 64, 96, 2
    theta = syn.init_params(C, seed=5, randomize_affine=True)
    flat = torch.from_numpy(theta.copy()) if rank == 0 else torch.zeros(theta.size)
    par.broadcast_params_(flat)                                  # replicas start from rank 0's weights
    assert np.array_equal(flat.numpy(), theta)
    x = syn.synthetic_triplets(GB, C, H, W, seed=9)
    lo, hi = par.shard_range(GB, rank, world)
    K, iK = syn.depth10k_intrinsics(W, H)
    cfg = oracle.make_cfg(hi - lo, C, H, W, K, iK)
    net = oracle.OracleNet(C); net.set_params(flat.numpy())
    oracle.lib().o_set_num_threads(2)
    net.train_loss(x[lo:hi], cfg, None, True)
    g = torch.from_numpy(net.get_grads())
    local = g.clone()
    par.allreduce_sum_(g)
    g *= 1.0 / par.world_size()
    np.save(os.path.join(out_dir, "local%d.npy" % rank), local.numpy()); np.save(os.path.join(out_dir, "red%d.npy" % rank), g.numpy())
    dist.barrier(); dist.destroy_process_group()


def test_allreduce_mean_of_shard_gradients(tmp_path):
    world = 2; port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    l0, l1 = np.load(tmp_path / "local0.npy"), np.load(tmp_path / "local1.npy")
    r0, r1 = np.load(tmp_path / "red0.npy"), np.load(tmp_path / "red1.npy")
    assert np.array_equal(r0, r1)                                # every replica applies the same update
    assert np.allclose(r0, (l0 + l1) / 2, rtol=1e-6, atol=1e-9)  # == mean of the per-shard gradients (per-replica BN)
    assert np.linalg.norm(l0 - l1) > 0                           # the shards really differ


# ---- the same exchange on the HIP gradient buffer: two ranks sharing the one card of a GPU box (RCCL refuses two ranks on one device, so the transport is gloo) ----
def _gpu_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root]
    import md2amd as m
    par = m.parallel; md = m.monodepth; syn = m.synthetic
    par.init_from_env("gloo")
    torch.cuda.set_device(0)
    N, C, H, W = 1, 1, 64, 96
    K, iK = syn.depth10k_intrinsics(W, H)
    params = md.Params(target_size=(W, H), batch_size=N, automasking=False)
    cache = md.TrainCache(md.SSIM(), md.Backproject(W, H), md.Project(W, H), K, iK, 2, [1, 3], [0.125, 0.25, 0.5, 1.0])
    enc = md.ResidualNetwork(18, in_channels=C)
    model = md.Model(enc, md.DepthDecoder(list(enc.stages), [2, 3, 4, 5]), md.PoseDecoder(512))
    model.load_params(syn.init_params(C, seed=5 + rank, randomize_affine=True))          # the ranks start DIFFERENT: construction must broadcast rank 0's state
    x = torch.from_numpy(syn.synthetic_triplets(world * N, C, H, W, seed=9)[rank * N:(rank + 1) * N].copy()).cuda()
    dp = par.DataParallel(model, comm="torch", x_like=x, parameters=params, cache=cache)
    np.save(os.path.join(out_dir, "p_init%d.npy" % rank), model.params())
    md.train_step(model, x, None, cache, params, do_update=False)
    np.save(os.path.join(out_dir, "g_local%d.npy" % rank), model.grads())
    par.allreduce_sum_(model.grad_buffer()); torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "g_red%d.npy" % rank), model.grads())
    for _ in range(3):
        dp.train_step(x, None, cache, params)
    np.save(os.path.join(out_dir, "p_end%d.npy" % rank), model.params())
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_data_parallel_step_on_the_hip_gradient_buffer(tmp_path):
    world = 2; port = _free_port()
    mp.spawn(_gpu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ld = lambda n: np.load(tmp_path / n)
    theta0 = md2amd.synthetic.init_params(1, seed=5, randomize_affine=True)
    assert np.array_equal(ld("p_init0.npy"), theta0) and np.array_equal(ld("p_init1.npy"), theta0)      # replicas made identical at construction
    l0, l1, r0, r1 = ld("g_local0.npy"), ld("g_local1.npy"), ld("g_red0.npy"), ld("g_red1.npy")
    assert np.array_equal(r0, r1)                                                                            # every rank holds the same reduced gradient
    assert np.linalg.norm(l0 - l1) > 0 and np.abs(r0 - (l0 + l1)).max() <= 1e-6 * np.abs(r0).max()          # == the sum of the two HIP shard gradients (ADAM scales by 1/world)
    assert np.array_equal(ld("p_end0.npy"), ld("p_end1.npy"))                                               # and the replicas stay bit-identical over further steps
    assert np.abs(ld("p_end0.npy") - theta0).max() > 1e-5
