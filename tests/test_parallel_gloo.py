"""N>1 path on CPU: world_size-2 gloo.  Checks the exchange the DP step performs (one all-reduce(sum) of the flat
gradient buffer, then grad_scale = 1/world) against the mean of the per-shard oracle gradients."""
import os
import socket

import numpy as np
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
