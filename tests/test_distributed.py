"""The N > 1 path on CPU: two gloo ranks shard the views (one view per rank, scene replicated) and all-reduce the
dense per-splat gradient buffer -- the same host-side protocol bench.py runs over RCCL, with the oracle standing in
for the GPU kernels (no GPU here).  Checks: sharding covers every view exactly once, the reduced gradient equals the
single-process sum over all views, and the max-over-ranks timing reduction works."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_views, out_dir):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bench import view_pose
    from conftest import make_scene
    from oracle import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle("f32")
    o.set_threads(2)
    scene = make_scene(np.random.default_rng(0), 300, log_scale=(-3.4, 0.6))  # replicated on every rank
    P, W, H = 300, 48, 32
    gbuf = torch.zeros(59 * P)
    mine = [v for v in range(n_views) if v % world == rank]  # one view per rank per step
    for v in mine:
        cam = o.lookat(*view_pose(v), width=W, height=H)
        dL = np.random.default_rng(100 + v).normal(size=(3, H, W)).astype(np.float32)
        g = o.render_backward_full(scene, cam, dL)
        flat = np.concatenate([g[k].reshape(-1) for k in ("pos", "scale", "rotq", "sh", "opacity")])
        gbuf += torch.from_numpy(flat)
    dist.all_reduce(gbuf)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), gbuf.numpy())
        np.save(os.path.join(out_dir, "tmax.npy"), t.numpy())
    np.save(os.path.join(out_dir, f"views_{rank}.npy"), np.array(mine))
    dist.barrier()
    dist.destroy_process_group()


def test_view_sharding_and_gradient_allreduce_gloo(tmp_path):
    import torch.multiprocessing as mp

    world, n_views = 2, 4
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_views, str(tmp_path)), nprocs=world, join=True)
    views = np.concatenate([np.load(tmp_path / f"views_{r}.npy") for r in range(world)])
    assert sorted(views.tolist()) == list(range(n_views))
    assert np.load(tmp_path / "tmax.npy")[0] == pytest.approx(0.2)
    # single-process reference: sum over all views
    sys.path.insert(0, ROOT)
    from bench import view_pose
    from conftest import make_scene
    from oracle import Oracle

    o = Oracle("f32")
    scene = make_scene(np.random.default_rng(0), 300, log_scale=(-3.4, 0.6))
    total = np.zeros(59 * 300, np.float64)
    for v in range(n_views):
        cam = o.lookat(*view_pose(v), width=48, height=32)
        dL = np.random.default_rng(100 + v).normal(size=(3, 32, 48)).astype(np.float32)
        g = o.render_backward_full(scene, cam, dL)
        total += np.concatenate([g[k].reshape(-1) for k in ("pos", "scale", "rotq", "sh", "opacity")])
    red = np.load(tmp_path / "reduced.npy")
    assert np.abs(red).max() > 0
    assert np.allclose(red, total, rtol=1e-4, atol=1e-5 * np.abs(total).max())
