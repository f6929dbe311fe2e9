"""The N > 1 path on CPU: two gloo ranks run the PACKAGE's view-parallel protocol
(luisacomputegaussiansplatting_amd.multi_gpu: ViewParallelTrainer + TorchCollective + shard_rows / view_of_rank -- the
very classes bench.py drives over RCCL), with a CPU stand-in for the engine only: the oracle computes a view's gradients
and a numpy restatement of lcgs_adam_step's arithmetic applies the update (no GPU here).  Checks: the views of a step
are disjoint and cover the batch; all three collective modes ("allreduce", "sharded", "sparse" -- the touched-row exchange
of round 3) leave every rank with the parameters a single process gets from the summed gradients; row ownership incl. the P mod N tail; max-over-ranks timing."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

KEYS = ("pos", "scale", "rotq", "sh", "opacity")
LR = {"pos": 1.6e-3, "sh_dc": 2.5e-2, "sh_rest": 1.25e-3, "opacity": 5e-2, "scale": 5e-3, "rot": 1e-2}
P, W, H, N_VIEWS, STEPS = 301, 48, 32, 4, 2  # P is odd: the sharded step has a one-row tail at N = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _poses():
    sys.path.insert(0, ROOT)
    from bench import view_pose

    return [view_pose(v) for v in range(N_VIEWS)]


def _raw_scene():
    from conftest import make_scene

    s = make_scene(np.random.default_rng(0), P, log_scale=(-3.4, 0.6))
    return {"pos": s["pos"], "scale": np.log(s["scale"]), "rotq": s["rotq"] * 1.3, "sh": s["sh"],
            "opacity": np.log(s["opacity"] / (1 - s["opacity"]))}


def _activate(raw):
    return {"pos": raw["pos"], "scale": np.exp(raw["scale"]),
            "rotq": raw["rotq"] / np.linalg.norm(raw["rotq"], axis=1, keepdims=True), "sh": raw["sh"],
            "opacity": 1.0 / (1.0 + np.exp(-raw["opacity"]))}


class OracleEngine:
    """CPU stand-in for multi_gpu.HipEngine: same interface, torch CPU tensors, the oracle as the kernels."""

    def __init__(self, oracle, raw_np, dL_of_view):
        import torch

        self.o = oracle
        self.raw = {k: torch.from_numpy(np.ascontiguousarray(raw_np[k], dtype=np.float32)) for k in KEYS}
        act = _activate(raw_np)
        self.activated = {k: torch.from_numpy(np.ascontiguousarray(act[k], dtype=np.float32)) for k in KEYS}
        self.activated["pos"], self.activated["sh"] = self.raw["pos"], self.raw["sh"]  # identity activation: one array
        self.m = {k: torch.zeros_like(self.raw[k]) for k in KEYS}
        self.v = {k: torch.zeros_like(self.raw[k]) for k in KEYS}
        self.dL_of_view = dL_of_view
        self.views_rendered = []
        self.touched = np.zeros(P, bool)  # rows this rank's views of the current step wrote (the HIP engine: on-screen rows)

    def forward_backward(self, cam, dL_dimg, grads, bg=(0.0, 0.0, 0.0), accumulate=False):
        import torch

        view, ocam = cam
        self.views_rendered.append(view)
        scene = {k: self.activated[k].numpy() for k in KEYS}
        g = self.o.render_backward_full(scene, ocam, self.dL_of_view(view))
        if not accumulate:
            self.touched[:] = False
        for k in KEYS:
            t = torch.from_numpy(g[k].reshape(grads[k].shape).astype(np.float32))
            self.touched |= (t.reshape(P, -1) != 0).any(1).numpy()
            if accumulate:
                grads[k].add_(t)
            else:
                grads[k].copy_(t)

    # the three device stages of the sparse exchange (csrc/kernels/comm_sparse.hip), restated on CPU tensors
    def sparse_touched_rows(self, grads, world, rank):
        rows = np.flatnonzero(self.touched).astype(np.int64)
        self.touched[:] = False  # consumed
        c = P // world
        bounds = [int(np.searchsorted(rows, o * c)) for o in range(world + 1)] + [len(rows)]
        return rows, bounds

    def sparse_pack(self, grads, rows, first, count, msg):
        import torch

        idx = rows[first:first + count]
        parts = [torch.from_numpy(idx.astype(np.int32)).view(torch.float32)]
        parts += [grads[k].reshape(P, -1)[idx].reshape(-1) for k in KEYS]
        msg.copy_(torch.cat(parts))

    def sparse_accumulate(self, grads, msg, count, rows=None):
        import torch

        idx = msg[:count].view(torch.int32).to(torch.int64)
        if rows is not None:  # (lcgs_sparse_accumulate drops rows outside the range; the protocol never sends any)
            assert bool(((idx >= rows[0]) & (idx < rows[0] + rows[1])).all())
        at = count
        for k in KEYS:
            w = grads[k].reshape(P, -1).shape[1]
            grads[k].reshape(P, -1)[idx] += msg[at:at + count * w].reshape(count, w)
            at += count * w

    def flush(self):
        pass

    # the three stages of the splat-ownership step (multi_gpu.TorchCollective.owner_step) on the oracle's
    # stage functions: the owner's per-splat half, the view rank's 2-D half, the owner's preprocess-backward
    OWNER_RECORD_FLOATS, OWNER_GRAD_FLOATS = 10, 9

    def _shard(self, span):
        first, count = span
        return {k: self.activated[k][first:first + count].numpy() for k in KEYS}

    def _project_shard(self, cam, span):
        _, ocam = cam
        s = self._shard(span)
        m, d, c2 = self.o.project(s["pos"], s["scale"], s["rotq"], ocam)
        _, _, tiles, rad = self.o.allocate_tiles(W, H, d, m, c2)
        return s, m, d, c2, tiles, rad

    def owner_records(self, cam, span, slot=0):
        import torch

        _, ocam = cam
        s, m, d, c2, tiles, _ = self._project_shard(cam, span)
        color = self.o.sh_process(np.array(ocam.position[:], np.float32), s["pos"], s["sh"])
        on = np.flatnonzero(tiles > 0)
        rec = np.concatenate([m[on], d[on, None], c2[on], color[on], s["opacity"][on, None]], axis=1).astype(np.float32)
        return torch.from_numpy((span[0] + on).astype(np.int64)), torch.from_numpy(np.ascontiguousarray(rec))

    def owner_render(self, cam, rows, rec, dL_dimg):
        import torch

        view, _ = cam
        self.views_rendered.append(view)
        n = int(rows.numel())
        if n == 0:
            return torch.zeros(0, self.OWNER_GRAD_FLOATS)
        assert bool((rows[1:] > rows[:-1]).all())  # ascending rows: equal depths blend in file order
        r = rec.numpy()
        m, d, c2, color, op = r[:, 0:2], r[:, 2], r[:, 3:6], r[:, 6:9], r[:, 9]
        bg = (0.0, 0.0, 0.0)
        mp, conic, tiles, rad = self.o.allocate_tiles(W, H, d, m, c2)
        k, v = self.o.copy_with_keys(W, H, mp, self.o.inclusive_sum(tiles), rad, d)
        ks, vs = self.o.sort_pairs(k, v)
        rng = self.o.get_ranges(ks, ((W + 15) // 16) * ((H + 15) // 16))
        _, fT, nc, _ = self.o.render_forward(W, H, bg, rng, vs, mp, conic, op, color)
        gm, gc, go, gcol = self.o.render_backward(W, H, bg, rng, vs, mp, conic, op, color, fT, nc, self.dL_of_view(view))
        return torch.from_numpy(np.concatenate([gm, gc, go[:, None], gcol], axis=1).astype(np.float32))

    def owner_backward(self, cam, span, rows, g2d, grads, accumulate=False, slot=0):
        import torch

        _, ocam = cam
        first, count = span
        s, _, _, _, _, rad = self._project_shard(cam, span)
        at = (rows.numpy() - first).astype(np.int64)
        gm, gc, gcol, go = np.zeros((count, 2), np.float32), np.zeros((count, 3), np.float32), np.zeros((count, 3), np.float32), \
            np.zeros(count, np.float32)
        g = g2d.numpy()
        gm[at], gc[at], go[at], gcol[at] = g[:, 0:2], g[:, 2:5], g[:, 5], g[:, 6:9]
        out = self.o.preprocess_backward(s, ocam, rad, gm, gc, gcol)
        out["opacity"] = go
        for k in KEYS:
            t = torch.from_numpy(out[k].reshape(grads[k][first:first + count].shape).astype(np.float32))
            if accumulate:
                grads[k][first:first + count] += t
            else:
                grads[k][first:first + count] = t

    def adam(self, grads, step, rows=None, b1=0.9, b2=0.999, eps=1e-15):
        """lcgs_adam_step (csrc/kernels/train.hip) restated: activated-space gradients -> raw-space -> Adam -> activate."""
        first, count = rows if rows is not None else (0, P)
        if count <= 0:
            return
        sl = slice(first, first + count)
        bc1, bc2 = 1.0 - b1 ** step, np.sqrt(1.0 - b2 ** step)
        for k in KEYS:
            g = grads[k][sl].numpy().astype(np.float32).reshape(count, -1)
            raw, m, v = (d[k][sl].numpy().reshape(count, -1) for d in (self.raw, self.m, self.v))
            act = self.activated[k][sl].numpy().reshape(count, -1)
            if k == "scale":
                g = g * act
            elif k == "opacity":
                g = g * act * (1 - act)
            elif k == "rotq":
                g = (g - act * (act * g).sum(1, keepdims=True)) / np.linalg.norm(raw, axis=1, keepdims=True)
            lr = np.full(g.shape[1], LR.get(k, 0.0), np.float32)
            if k == "sh":
                lr[:3], lr[3:] = LR["sh_dc"], LR["sh_rest"]
            if k == "rotq":
                lr[:] = LR["rot"]
            m[:] = b1 * m + (1 - b1) * g
            v[:] = b2 * v + (1 - b2) * g * g
            raw -= (lr / bc1) * m / (np.sqrt(v) / bc2 + eps)
            if k == "scale":
                act[:] = np.exp(raw)
            elif k == "opacity":
                act[:] = 1.0 / (1.0 + np.exp(-raw))
            elif k == "rotq":
                act[:] = raw / np.linalg.norm(raw, axis=1, keepdims=True)


def _dL(view):
    return np.random.default_rng(100 + view).normal(size=(3, H, W)).astype(np.float32)


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import luisacomputegaussiansplatting_amd.multi_gpu as mg
    from oracle import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle("f32")
    o.set_threads(2)
    cams = [(v, o.lookat(*p, width=W, height=H)) for v, p in enumerate(_poses())]
    stats = {}
    for mode in ("allreduce", "sharded", "sparse", "owner", "allreduce_2views", "sparse_2views"):
        engine = OracleEngine(o, _raw_scene(), _dL)  # the scene is replicated on every rank
        grads = {k: torch.zeros_like(engine.raw[k]) for k in KEYS}
        if mode.endswith("_2views"):  # two views per rank and optimiser step: gradients accumulate, ONE collective
            coll = mg.TorchCollective(dist, rank, world)
            trainer = mg.ViewParallelTrainer(engine, coll, cams, grads, mode=mode[:-len("_2views")], views_per_step=2)
            trainer.step(None)
            if mode.startswith("sparse"):
                stats[mode] = coll.last_stats
        elif mode in ("sparse", "owner"):
            coll = mg.TorchCollective(dist, rank, world)
            trainer = mg.ViewParallelTrainer(engine, coll, cams, grads, mode=mode)
            for _ in range(STEPS):
                trainer.step(None)
            stats[mode] = coll.last_stats
        else:
            trainer = mg.ViewParallelTrainer(engine, mg.TorchCollective(dist, rank, world), cams, grads, mode=mode)
            for _ in range(STEPS):
                trainer.step(None)
        np.savez(os.path.join(out_dir, f"{mode}_{rank}.npz"), views=np.array(engine.views_rendered),
                 **{f"act_{k}": engine.activated[k].numpy() for k in KEYS},
                 **{f"raw_{k}": engine.raw[k].numpy() for k in KEYS})
    import json

    with open(os.path.join(out_dir, f"stats_{rank}.json"), "w") as f:
        json.dump(stats, f)
    # bench.py's timing reduction: max over ranks
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(os.path.join(out_dir, "tmax.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_rows_and_view_assignment(lcgs):
    import luisacomputegaussiansplatting_amd.multi_gpu as mg

    for P_, N in ((301, 2), (6_131_954, 8), (7, 8), (0, 4), (4096, 4)):
        spans = [lcgs.api.shard_rows(P_, N, r) for r in range(N)]
        c = P_ // N
        assert spans == [(r * c, c) for r in range(N)]  # equal shards; the P mod N rows behind them are the tail
        assert P_ - c * N < N
    for N in (1, 2, 4, 8):
        for step in range(3):
            views = [mg.view_of_rank(step, r, N, 8) for r in range(N)]
            assert len(set(views)) == N  # one view per rank, no view twice in a step
        assert sorted(mg.view_of_rank(s, r, N, 8) for s in range(8 // N) for r in range(N)) == list(range(8))
    assert mg.allreduce_bus_bytes_per_gpu(6_131_954, 8) == 2 * 7 * 59 * 4 * 6_131_954 // 8
    assert mg.allreduce_bus_bytes_per_gpu(6_131_954, 1) == 0


def test_view_parallel_protocol_on_two_gloo_ranks(tmp_path, oracle):
    import torch
    import torch.multiprocessing as mp

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert np.load(tmp_path / "tmax.npy")[0] == pytest.approx(0.2)

    # single-process reference: per step, the gradients of the step's views summed, then one dense Adam step
    cams = [(v, oracle.lookat(*p, width=W, height=H)) for v, p in enumerate(_poses())]
    ref = OracleEngine(oracle, _raw_scene(), _dL)
    g = {k: torch.zeros_like(ref.raw[k]) for k in KEYS}
    for step in range(STEPS):
        total = {k: torch.zeros_like(ref.raw[k]) for k in KEYS}
        for r in range(world):
            ref.forward_backward(cams[(step * world + r) % N_VIEWS], None, g)
            for k in KEYS:
                total[k] += g[k]
        ref.adam(total, step + 1)

    res = {(m, r): np.load(tmp_path / f"{m}_{r}.npz") for m in ("allreduce", "sharded", "sparse") for r in range(world)}
    for m in ("allreduce", "sharded", "sparse"):
        views = np.stack([res[(m, r)]["views"] for r in range(world)], axis=1)  # [step, rank]
        assert views.tolist() == [[0, 1], [2, 3]]  # disjoint per step, the batch covered after STEPS steps
        for r in range(world):
            for k in KEYS:
                a, b = res[(m, r)][f"act_{k}"], ref.activated[k].numpy()
                assert np.abs(b).max() > 0
                # (sum order differs between all_reduce and the single process: fp32 round-off only)
                assert np.allclose(a, b, rtol=2e-4, atol=2e-6 * np.abs(b).max()), (m, r, k, np.abs(a - b).max())
        # every rank holds the same activated scene afterwards
        for k in KEYS:
            assert np.array_equal(res[(m, 0)][f"act_{k}"], res[(m, 1)][f"act_{k}"]), (m, k)
    # splat ownership (the prototype of DESIGN.md 7b): nothing replicated -- every rank ends with ITS rows of the scene the
    # single process gets, the other rows untouched; what it sent is 2-D records and 2-D gradients of on-screen rows only
    import luisacomputegaussiansplatting_amd.multi_gpu as mg
    import json as _json

    for r in range(world):
        got = np.load(tmp_path / f"owner_{r}.npz")
        assert got["views"].tolist() == [r, 2 + r]
        first, count = mg.owner_range(P, world, r)
        own = np.r_[first:first + count]
        other = np.setdiff1d(np.arange(P), own)
        for k in KEYS:
            for kind, refd in (("act", ref.activated), ("raw", ref.raw)):
                a, b = got[f"{kind}_{k}"][own], refd[k].numpy()[own]
                assert np.allclose(a, b, rtol=2e-4, atol=2e-6 * np.abs(refd[k].numpy()).max()), ("owner", r, kind, k)
        assert np.array_equal(got["raw_scale"][other], _raw_scene()["scale"][other].astype(np.float32))
        st = _json.load(open(tmp_path / f"stats_{r}.json"))["owner"]
        assert 0 < st["bytes_sent"] < (world - 1) * (P // world) * 59 * 4  # below even ONE half of the dense exchange
    assert sum(mg.owner_range(P, world, r)[1] for r in range(world)) == P
    # two views per rank and step: one step over all four views = one dense Adam step on the sum of their gradients
    ref2 = OracleEngine(oracle, _raw_scene(), _dL)
    total = {k: torch.zeros_like(ref2.raw[k]) for k in KEYS}
    for v in range(N_VIEWS):
        ref2.forward_backward(cams[v], None, g)
        for k in KEYS:
            total[k] += g[k]
    ref2.adam(total, 1)
    for r in range(world):
        for mode2 in ("allreduce_2views", "sparse_2views"):  # (sparse: the touched set is the UNION of the rank's two views)
            got = np.load(tmp_path / f"{mode2}_{r}.npz")
            assert got["views"].tolist() == [r, 2 + r]  # micro-step j of the step: view j * world + rank
            for k in KEYS:
                b = ref2.activated[k].numpy()
                assert np.allclose(got[f"act_{k}"], b, rtol=2e-4, atol=2e-6 * np.abs(b).max()), (mode2, r, k)
    # the sparse step's reduce half moved only touched rows: fewer bytes than the dense reduce-scatter's (N-1)/N S
    import json

    for r in range(world):
        st = json.load(open(tmp_path / f"stats_{r}.json"))["sparse"]
        dense_reduce = (world - 1) * (P // world) * 59 * 4
        gather = (world - 1) * (P // world) * 59 * 4
        assert 0 < st["touched_rows"] < P
        assert st["bytes_sent"] - gather < dense_reduce, st
    # something was learnt (the update is not a no-op)
    assert not np.allclose(ref.activated["opacity"].numpy(), _activate(_raw_scene())["opacity"].astype(np.float32))
    # sharded: raw parameters are authoritative on their owner (and on the tail) only
    sys.path.insert(0, ROOT)
    import luisacomputegaussiansplatting_amd as L

    for r in range(world):
        first, count = L.api.shard_rows(P, world, r)
        own = np.r_[first:first + count, count * world:P]
        for m in ("sharded", "sparse"):
            for k in ("scale", "opacity", "rotq"):
                a, b = res[(m, r)][f"raw_{k}"][own], ref.raw[k].numpy()[own]
                assert np.allclose(a, b, rtol=2e-4, atol=2e-6 * np.abs(b).max()), (m, r, k)
            other = np.setdiff1d(np.arange(P), own)
            assert np.array_equal(res[(m, r)]["raw_scale"][other], _raw_scene()["scale"][other].astype(np.float32))


def _worker_c5(rank, world, port, out_dir):
    """BASELINE config C5 in miniature: 8 ranks, the 8 C5 views (one per rank), one optimiser step, both collective modes."""
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import luisacomputegaussiansplatting_amd.multi_gpu as mg
    from bench import view_pose
    from oracle import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle("f32")
    o.set_threads(1)
    cams = [(v, o.lookat(*view_pose(v), width=W, height=H)) for v in range(8)]
    for mode in ("allreduce", "sharded", "sparse", "owner"):
        engine = OracleEngine(o, _raw_scene(), _dL)
        grads = {k: torch.zeros_like(engine.raw[k]) for k in KEYS}
        trainer = mg.ViewParallelTrainer(engine, mg.TorchCollective(dist, rank, world), cams, grads, mode=mode)
        trainer.step(None)
        np.savez(os.path.join(out_dir, f"c5_{mode}_{rank}.npz"), views=np.array(engine.views_rendered),
                 **{f"act_{k}": engine.activated[k].numpy() for k in KEYS})
    dist.barrier()
    dist.destroy_process_group()


def test_c5_shape_eight_ranks_eight_views_on_gloo(tmp_path, oracle):
    """The host protocol at the node's full width (the kernels' part is per-rank and identical to N = 1): 8 ranks x 1 view,
    P = 301 -> shards of 37 rows and a 5-row tail; every rank ends with the scene a single process gets from one dense Adam
    step on the sum of the eight views' gradients."""
    import torch
    import torch.multiprocessing as mp

    sys.path.insert(0, ROOT)
    from bench import view_pose

    world = 8
    mp.spawn(_worker_c5, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    cams = [(v, oracle.lookat(*view_pose(v), width=W, height=H)) for v in range(8)]
    ref = OracleEngine(oracle, _raw_scene(), _dL)
    g = {k: torch.zeros_like(ref.raw[k]) for k in KEYS}
    total = {k: torch.zeros_like(ref.raw[k]) for k in KEYS}
    for v in range(8):
        ref.forward_backward(cams[v], None, g)
        for k in KEYS:
            total[k] += g[k]
    ref.adam(total, 1)
    for mode in ("allreduce", "sharded", "sparse"):
        outs = [np.load(tmp_path / f"c5_{mode}_{r}.npz") for r in range(world)]
        assert [o_["views"].tolist() for o_ in outs] == [[r] for r in range(world)]  # one view per rank, all eight covered
        for r in range(world):
            for k in KEYS:
                b = ref.activated[k].numpy()
                assert np.allclose(outs[r][f"act_{k}"], b, rtol=3e-4, atol=3e-6 * np.abs(b).max()), (mode, r, k)
                assert np.array_equal(outs[r][f"act_{k}"], outs[0][f"act_{k}"]), (mode, r, k)  # replicas stay identical
    # splat ownership at the node's width: every rank holds ITS 37 (the last: 42) rows of that scene
    import luisacomputegaussiansplatting_amd.multi_gpu as mg

    outs = [np.load(tmp_path / f"c5_owner_{r}.npz") for r in range(world)]
    assert [o_["views"].tolist() for o_ in outs] == [[r] for r in range(world)]
    for r in range(world):
        first, count = mg.owner_range(P, world, r)
        for k in KEYS:
            b = ref.activated[k].numpy()
            assert np.allclose(outs[r][f"act_{k}"][first:first + count], b[first:first + count], rtol=3e-4,
                               atol=3e-6 * np.abs(b).max()), ("owner", r, k)
