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


# ---- the library's gradient buckets (md2_backward all-reduces them as they complete): a pure host statement about their ranges, then the bucketed exchange on two CPU ranks
def _all_level_sets():
    import itertools
    return [c for k in range(1, 6) for c in itertools.combinations(range(1, 6), k)]


def test_gradient_buckets_tile_the_flat_buffer_exactly():
    """md2_comm_bucket_plan (host-only): for ResNet-18 and -50, C = 1 and 3 and every ascending scale_levels subset of 1:5 the up-to-four [lo, hi) slices are disjoint,
    ordered back to front (the order the pullback completes them) and cover [0, n) with no gap -- a gap would leave a slice of the gradient un-reduced, an overlap would
    sum it twice."""
    par = md2amd.parallel
    seen = set()
    for depth in (18, 50):
        for C in (1, 3):
            for levels in _all_level_sets():
                buckets, n = par.bucket_plan(2, C, 64, 96, depth, levels)
                assert 2 <= len(buckets) <= 4 and n > 0
                assert all(hi > lo for lo, hi in buckets)
                assert buckets[0][1] == n and buckets[-1][0] == 0                      # the last-produced bucket starts the buffer, the first-produced one ends it
                for (lo_a, _), (_, hi_b) in zip(buckets[:-1], buckets[1:]):
                    assert hi_b == lo_a                                                # each bucket ends exactly where the one completed before it begins
                assert sum(hi - lo for lo, hi in buckets) == n
                seen.add((depth, C, levels, n))
    assert len({n for _, _, _, n in seen}) > 20                                        # the level subsets really change the decoder (and with it the boundaries)


def _bucket_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root]
    import md2amd as m
    par = m.parallel
    par.init_from_env("gloo")
    res = []
    for depth, levels in ((18, (2, 3, 4, 5)), (50, (1, 3, 4))):
        buckets, n = par.bucket_plan(2, 1, 64, 96, depth, levels)
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(n, generator=g)
        whole = local.clone(); par.allreduce_sum_(whole)                               # the unbucketed exchange (md2_allreduce_grads)
        bucketed = local.clone()
        for lo, hi in buckets:                                                         # the bucketed one, slice by slice in completion order (what md2_backward issues)
            par.allreduce_sum_(bucketed[lo:hi])
        res.append(bool(torch.equal(whole, bucketed)) and not torch.equal(whole, local))
    np.save(os.path.join(out_dir, "ok%d.npy" % rank), np.array(res))
    dist.barrier(); dist.destroy_process_group()


def test_bucketed_allreduce_equals_whole_buffer_allreduce_on_two_cpu_ranks(tmp_path):
    world = 2; port = _free_port()
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(tmp_path / ("ok%d.npy" % r)).all()


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
