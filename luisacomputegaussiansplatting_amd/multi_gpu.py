"""View-parallel training across the GPUs of one node: the host protocol.

One process per GPU.  The scene is replicated on every rank; a batch of views is sharded one view per rank; each rank
runs forward + backward for its view (no data-path collective), then the dense per-splat gradients are summed over
the ranks and the optimiser step is applied -- in one of two exact (f32) ways:

  mode "allreduce"   lcgs_grads_allreduce (chunked, overlapping the backward's tail) + a dense lcgs_adam_step on
                     every rank: every rank does the whole optimiser's work on identical inputs.
  mode "sharded"     lcgs_adam_step_sharded: reduce-scatter -> Adam on the rank's own rows (+ the < N tail rows) ->
                     all-gather of the refreshed activated arrays.  Same bytes on the wire, 1/N of the optimiser work.

  mode "sparse"      lcgs_adam_step_sparse: like "sharded", but the REDUCE half carries only the rows this rank's views
                     touched (a view sees 39 % of the bicycle stand-in's splats): each rank sends a row's owner the
                     touched rows of the owner's shard -- indices + 59 floats each -- the owner adds them in rank order,
                     runs Adam on its shard, and the refreshed activated rows are all-gathered.  Dense-Adam semantics
                     (rows nobody touched still decay their moments), exact up to the order of the f32 sums.

The reference has no counterpart (single device, app/main.cpp:162-163).

Three pluggable parts keep ONE protocol for the product and for its CPU tests:
  * the *engine* computes a view's gradients and applies Adam to a row range (`HipEngine` = the HIP kernels through the
    C ABI; tests/ plug in a CPU stand-in so that the protocol runs on gloo without a GPU),
  * the *collective* moves bytes (`RcclCollective` = lcgs_comm_* on RCCL over xGMI, the product path;
    `TorchCollective` = a torch.distributed process group -- gloo in the CPU tests, or nccl as a cross-check),
  * `ViewParallelTrainer` is the protocol itself: view assignment, step order, row ownership (`shard_rows`).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

from . import api

KEYS = ("pos", "scale", "rotq", "sh", "opacity")
ROW_FLOATS = {"pos": 3, "scale": 3, "rotq": 4, "sh": 48, "opacity": 1}
GRAD_FLOATS_PER_SPLAT = sum(ROW_FLOATS.values())  # 59


def view_of_rank(step: int, rank: int, world_size: int, num_views: int) -> int:
    """Which view of the batch a rank renders at a (micro-)step: consecutive views to consecutive ranks, the window
    sliding by world_size per step (every view is visited when the steps cover the batch)."""
    return (step * world_size + rank) % num_views


def allreduce_bus_bytes_per_gpu(num_gaussians: int, world_size: int) -> int:
    """Bytes one GPU sends (= receives) over xGMI for one gradient all-reduce (ring / direct algorithms alike):
    2 (N-1)/N S with S = 59 floats per splat.  The sharded step moves the same: (N-1)/N S for the reduce-scatter and
    (N-1)/N S for the all-gather."""
    if world_size <= 1:
        return 0
    s = GRAD_FLOATS_PER_SPLAT * 4 * num_gaussians
    return 2 * (world_size - 1) * s // world_size


def sparse_bus_bytes_per_gpu(num_gaussians: int, world_size: int, touched_rows: int) -> int:
    """Expected bytes one GPU SENDS per sparse step when its views touched `touched_rows` rows spread evenly over the
    shards: (N-1)/N of them travel to their owners as (1 + 59) words each, then the dense all-gather of the refreshed
    activated rows ((N-1)/N S, as in the sharded step).  (The measured figure comes from lcgs_comm_get_stats.)"""
    if world_size <= 1:
        return 0
    reduce_half = (world_size - 1) * touched_rows * (GRAD_FLOATS_PER_SPLAT + 1) * 4 // world_size
    gather_half = (world_size - 1) * (num_gaussians // world_size) * GRAD_FLOATS_PER_SPLAT * 4
    return reduce_half + gather_half


# ------------------------------------------------------------------------------------------------ collectives
class RcclCollective:
    """The product path: RCCL over xGMI through the C ABI (lcgs_comm_*), on the communicator's own HIP stream."""

    name = "rccl (lcgs_comm C ABI)"

    def __init__(self, ctx: "api.Context", rank: int, world_size: int, exchange: Optional[Callable] = None, loopback=None):
        """loopback: an api.LoopbackGroup -- N contexts on one device, one host thread per rank (rehearsals / tests)"""
        self.comm = api.Comm(ctx, rank, world_size, exchange, loopback=loopback)
        self.rank, self.world_size = rank, world_size
        self._img = None
        self._owner_async = False
        self.owner_redos = 0  # ownership steps that had to be repeated (a clipped message / truncated pairs on some rank)

    def allreduce_grads(self, grads: dict):
        self.comm.allreduce_grads(grads)

    def sharded_adam(self, engine, grads: dict, step: int):
        engine.adam_sharded(self.comm, grads, step)

    def prepare(self, mode: str):
        """collective call on trainer construction: the sparse step needs the backward passes to flag their rows"""
        self.comm.track_touched_rows(mode == "sparse")

    def sparse_adam(self, engine, grads: dict, step: int):
        engine.adam_sparse(self.comm, grads, step)
        self.last_stats = self.comm.stats()

    def owner_step(self, engine, cams, dL_dimg, grads: dict, step: int, optimise: bool = True):
        """The splat-ownership step through the C ABI: lcgs_owner_step_forward / _backward move the records and
        the 2-D gradients over RCCL point-to-point; Adam then touches this rank's own rows only."""
        import torch

        cam = cams[self.rank]
        if self._img is None or tuple(self._img.shape) != (3, cam.height, cam.width):
            self._img = torch.zeros(3, cam.height, cam.width, device=grads["pos"].device, dtype=torch.float32)
        if not self._owner_async:  # no read-back from the second step on (lcgs_owner_step_set_async / _finish)
            self.comm.owner_step_set_async(True)
            self._owner_async = True
        self.owner_redos += self.comm.owner_step(cams, self._img, dL_dimg, grads)
        if optimise:
            engine.adam(grads, step, rows=api.owner_rows(int(grads["pos"].shape[0]), self.world_size, self.rank))
        st = self.comm.stats()
        self.last_stats = {"bytes_sent": st["bytes_sent"], "bytes_received": st["bytes_received"],
                           "on_screen_rows_received": st["touched_rows"], "collective_groups": st["collective_groups"]}

    def close(self):
        self.comm.close()


class TorchCollective:
    """A torch.distributed process group as the transport: gloo on CPU tensors (the CPU tests of the protocol) or nccl
    (= RCCL) on device tensors as a cross-check of RcclCollective.  Same row ownership, same step order."""

    name = "torch.distributed"

    def __init__(self, dist, rank: int, world_size: int):
        self.dist, self.rank, self.world_size = dist, rank, world_size

    def allreduce_grads(self, grads: dict):
        for k in KEYS:
            self.dist.all_reduce(grads[k])

    def sharded_adam(self, engine, grads: dict, step: int):
        import torch

        P = int(grads["pos"].shape[0])
        first, count = api.shard_rows(P, self.world_size, self.rank)
        tail0 = count * self.world_size
        # 1. reduce-scatter (emulated where the backend has none: the sum everywhere, every rank then USES its own rows
        #    and the tail only -- the others are never read)
        for k in KEYS:
            self.dist.all_reduce(grads[k])
        # 2. Adam on the own rows and on the tail
        engine.adam(grads, step, rows=(first, count))
        engine.adam(grads, step, rows=(tail0, P - tail0))
        # 3. all-gather of the refreshed activated rows
        if count > 0:
            for k in KEYS:
                act = engine.activated[k]
                mine = act[first:first + count].contiguous()
                parts = [torch.empty_like(mine) for _ in range(self.world_size)]
                self.dist.all_gather(parts, mine)
                for r, part in enumerate(parts):
                    act[r * count:(r + 1) * count] = part

    def prepare(self, mode: str):
        pass

    def sparse_adam(self, engine, grads: dict, step: int):
        """The protocol of lcgs_adam_step_sparse (csrc/host/comm.cpp) over a process group.  The engine supplies the
        three device stages -- touched rows, pack, accumulate -- (HipEngine: the C ABI's lcgs_sparse_*; the CPU tests:
        a numpy restatement); this method is the exchange: counts by all_gather, one message per peer by send / recv."""
        import torch

        dist, N, me = self.dist, self.world_size, self.rank
        P = int(grads["pos"].shape[0])
        first, count = api.shard_rows(P, N, me)
        tail0 = count * N
        handle, owner_first = engine.sparse_touched_rows(grads, N, me)  # ascending rows; positions of the shard starts
        dev = grads["pos"].device
        mine = torch.tensor(owner_first, dtype=torch.int64, device=dev)
        table = [torch.empty_like(mine) for _ in range(N)]
        dist.all_gather(table, mine)
        table = [t.tolist() for t in table]
        rows_of = lambda src, owner: table[src][owner + 1] - table[src][owner]
        sends, recvs, ops = {}, {}, []
        for o in range(N):
            if o == me:
                continue
            n_out, n_in = rows_of(me, o), rows_of(o, me)
            if n_out > 0:
                sends[o] = torch.empty(api.sparse_message_words(n_out), dtype=torch.float32, device=dev)
                engine.sparse_pack(grads, handle, owner_first[o], n_out, sends[o])
            if n_in > 0:
                recvs[o] = torch.empty(api.sparse_message_words(n_in), dtype=torch.float32, device=dev)
        engine.flush()  # the messages are complete before the transport reads them
        for o in range(N):
            if o in sends:
                ops.append(dist.P2POp(dist.isend, sends[o], o))
            if o in recvs:
                ops.append(dist.P2POp(dist.irecv, recvs[o], o))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if P > tail0:  # the < N tail rows every rank keeps: summed densely
            for k in KEYS:
                t = grads[k][tail0:].contiguous()
                dist.all_reduce(t)
                grads[k][tail0:] = t
        for o in range(N):  # rank order: a fixed order of the f32 sums
            if o in recvs:
                engine.sparse_accumulate(grads, recvs[o], rows_of(o, me), rows=(first, count))  # only own-shard rows are accepted
        self.last_stats = {"bytes_sent": 4 * sum(int(t.numel()) for t in sends.values()),
                           "bytes_received": 4 * sum(int(t.numel()) for t in recvs.values()),
                           "touched_rows": owner_first[N + 1]}
        engine.adam(grads, step, rows=(first, count))
        engine.adam(grads, step, rows=(tail0, P - tail0))
        if count > 0:
            gathered = 0
            for k in KEYS:
                act = engine.activated[k]
                own = act[first:first + count].contiguous()
                parts = [torch.empty_like(own) for _ in range(N)]
                dist.all_gather(parts, own)
                for r, part in enumerate(parts):
                    act[r * count:(r + 1) * count] = part
                gathered += own.numel() * 4 * (N - 1)
            self.last_stats["bytes_sent"] += gathered
            self.last_stats["bytes_received"] += gathered

    def owner_step(self, engine, cams, dL_dimg, grads: dict, step: int):
        """PROTOTYPE of the splat-ownership step, verified on gloo with a CPU restatement as the engine
        (tests/test_distributed.py) and on the device through HipEngine's two-halves frame (csrc/abi_owner.cpp).  Rank o owns the rows owner_range(P, N, o) -- parameters, moments, gradients -- and
        nothing is replicated or all-gathered.  Per step, with rank v rendering view v:
          1. every owner projects ITS rows for EVERY view of the step and sends view v's rank the 2-D inputs of the rows
             that reach the screen (row index + projected mean, depth, 2-D covariance, colour, opacity: 44 bytes a row);
          2. rank v renders its view from the rows it received (ascending row order = the file's order of equal depths),
             differentiates it and returns, to every owner, the 2-D gradients of that owner's rows (40 bytes a row);
          3. every owner maps the 2-D gradients of all N views to parameter gradients of its rows, sums them in view order
             and applies Adam to its rows.
        What crosses the wire per GPU and step is (N-1)/N x (44 + 40) bytes x on-screen rows instead of 2 (N-1)/N x 236 bytes
        x ALL rows.  The engine supplies the three stages (owner_records / owner_render / owner_backward)."""
        import torch

        dist, N, me = self.dist, self.world_size, self.rank
        P = int(grads["pos"].shape[0])
        span = owner_range(P, N, me)
        dev = grads["pos"].device
        # ---- 1. my rows, every view: on-screen rows + their 2-D inputs
        if hasattr(engine, "owner_records_all"):  # (the device engine: N asynchronous projections, one read-back of the counts)
            mine = engine.owner_records_all(cams, span)
        else:
            mine = [engine.owner_records(cams[v], span, slot=v) for v in range(N)]  # [(rows [n], rec f32 [n, R])]
        counts = torch.tensor([int(m[0].numel()) for m in mine], dtype=torch.int64, device=dev)
        table = [torch.empty_like(counts) for _ in range(N)]
        dist.all_gather(table, counts)
        table = [t.tolist() for t in table]  # table[o][v]: rows owner o has on view v's screen
        R = engine.OWNER_RECORD_FLOATS
        row_dtype = mine[me][0].dtype  # (the engine's: int32 on the device)
        got_rows = {o: torch.empty(table[o][me], dtype=row_dtype, device=dev) for o in range(N)}
        got_rec = {o: torch.empty(table[o][me], R, dtype=torch.float32, device=dev) for o in range(N)}
        got_rows[me], got_rec[me] = mine[me]
        ops, sent = [], 0
        for o in range(N):
            if o == me:
                continue
            if table[me][o] > 0:  # my rows on view o's screen -> rank o
                ops += [dist.P2POp(dist.isend, mine[o][0], o), dist.P2POp(dist.isend, mine[o][1], o)]
                sent += table[me][o] * (mine[o][0].element_size() + 4 * R)
            if table[o][me] > 0:
                ops += [dist.P2POp(dist.irecv, got_rows[o], o), dist.P2POp(dist.irecv, got_rec[o], o)]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        # ---- 2. my view from everybody's rows (owner order = ascending row order), its 2-D gradients back to the owners
        rows_all = torch.cat([got_rows[o] for o in range(N)])
        rec_all = torch.cat([got_rec[o] for o in range(N)])
        g2d_all = engine.owner_render(cams[me], rows_all, rec_all, dL_dimg)  # f32 [n, G]
        G = engine.OWNER_GRAD_FLOATS
        back, at = {}, 0
        for o in range(N):
            back[o] = g2d_all[at:at + table[o][me]].contiguous()
            at += table[o][me]
        g_in = {o: torch.empty(table[me][o], G, dtype=torch.float32, device=dev) for o in range(N)}
        g_in[me] = back[me]
        ops = []
        for o in range(N):
            if o == me:
                continue
            if table[o][me] > 0:
                ops.append(dist.P2POp(dist.isend, back[o], o))
                sent += table[o][me] * 4 * G
            if table[me][o] > 0:
                ops.append(dist.P2POp(dist.irecv, g_in[o], o))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        # ---- 3. my rows: 2-D gradients of every view -> parameter gradients (summed in view order) -> Adam
        for v in range(N):
            engine.owner_backward(cams[v], span, mine[v][0], g_in[v], grads, accumulate=v > 0, slot=v)
        engine.adam(grads, step, rows=span)
        self.last_stats = {"bytes_sent": sent, "on_screen_rows_received": int(rows_all.numel())}

    def close(self):
        pass


def owner_range(num_gaussians: int, world_size: int, rank: int):
    """rows a rank OWNS in the splat-ownership step: equal contiguous shards, the P mod N tail with the last rank"""
    c = num_gaussians // world_size
    return (c * rank, c if rank < world_size - 1 else num_gaussians - c * rank)


# ------------------------------------------------------------------------------------------------ engine
class HipEngine:
    """The HIP kernels behind the protocol: lcgs_render_forward(keep_state) + lcgs_render_backward for a view, and
    lcgs_adam_step / lcgs_adam_step_sharded for the update.  Device tensors (torch owns the memory)."""

    def __init__(self, renderer: "api.Renderer", raw: Optional[dict], activated: dict, lr: Optional[dict],
                 betas=(0.9, 0.999), eps: float = 1e-15):
        """raw / lr may be None for an engine that only produces gradients (no optimiser state is allocated)."""
        import torch

        self.r, self.raw, self.activated, self.lr, self.betas, self.eps = renderer, raw, activated, lr, betas, eps
        self.m = {k: torch.zeros_like(raw[k]) for k in KEYS} if raw is not None else None
        self.v = {k: torch.zeros_like(raw[k]) for k in KEYS} if raw is not None else None
        self._img = None
        self.compact_rows = False  # N = 1 only: lcgs_render_backward_compact (row r = the frame's r-th on-screen splat)
        renderer.bind_scene(*[activated[k] for k in KEYS])

    def forward_backward(self, cam, dL_dimg, grads: dict, bg=(0.0, 0.0, 0.0), accumulate: bool = False):
        """one view: frame + its gradients into `grads` (accumulate: added to what they hold)"""
        import torch

        if self._img is None or tuple(self._img.shape) != (3, cam.height, cam.width):
            self._img = torch.empty(3, cam.height, cam.width, device=dL_dimg.device, dtype=torch.float32)
        self.r.forward(cam, self._img, bg=bg, keep_state=True, sync=False)
        self.r.backward(dL_dimg, *[grads[k] for k in KEYS], compact=self.compact_rows, accumulate=accumulate)

    def adam(self, grads: dict, step: int, rows=None):
        if self.raw is None:
            raise ValueError("this engine was built without optimiser state (raw=None)")
        if rows is None:
            self.r.adam_step(grads, self.raw, self.m, self.v, self.activated, step, self.lr, self.betas, self.eps)
            return
        first, count = rows
        if count <= 0:
            return
        sub = lambda d: {k: d[k][first:first + count] for k in KEYS}
        self.r.adam_step(sub(grads), sub(self.raw), sub(self.m), sub(self.v), sub(self.activated), step, self.lr,
                         self.betas, self.eps)

    # ---- splat ownership on the device: the frame in two halves (csrc/abi_owner.cpp)
    OWNER_RECORD_FLOATS, OWNER_GRAD_FLOATS = 12, 12

    def owner_records(self, cam, span, slot: int = 0):
        """slot: the view's index inside the step (its state is kept there until owner_backward(slot=...))"""
        return self.r.owner_project(slot, cam, span[0], span[1], keep_state=True)

    def owner_records_all(self, cams, span):
        """every view of the step (view v -> slot v) with one host synchronisation (lcgs_owner_counts)"""
        return self.r.owner_project_all(cams, span[0], span[1], keep_state=True)

    def owner_render(self, cam, rows, recs, dL_dimg, bg=(0.0, 0.0, 0.0)):
        import torch

        if self._img is None or tuple(self._img.shape) != (3, cam.height, cam.width):
            self._img = torch.empty(3, cam.height, cam.width, device=recs.device, dtype=torch.float32)
        g2d = torch.zeros(int(rows.shape[0]), self.OWNER_GRAD_FLOATS, device=recs.device, dtype=torch.float32)
        if rows.numel() == 0:
            return g2d
        self.r.owner_render(cam, rows.contiguous(), recs.contiguous(), self._img, bg=bg, keep_state=True)
        self.r.owner_render_backward(dL_dimg, g2d)
        return g2d

    def owner_backward(self, cam, span, rows, g2d, grads: dict, accumulate: bool = False, slot: int = 0):
        self.r.owner_backward(slot, g2d.contiguous(), *[grads[k] for k in KEYS], accumulate=accumulate)

    def adam_sharded(self, comm: "api.Comm", grads: dict, step: int):
        comm.adam_step_sharded(grads, self.raw, self.m, self.v, self.activated, step, self.lr, self.betas, self.eps)

    def adam_sparse(self, comm: "api.Comm", grads: dict, step: int):
        comm.adam_step_sparse(grads, self.raw, self.m, self.v, self.activated, step, self.lr, self.betas, self.eps)

    # The stages of the sparse exchange one by one, for a transport that is not RCCL-behind-the-C-ABI (TorchCollective).
    # The touched set is kept by a communicator object attached to the context: the RcclCollective's own, or -- for
    # another transport -- a world-size-1 communicator that only serves as the context's row tracker.
    _tracker = None
    _own_tracker = False

    def prepare(self, mode: str, collective):
        """called by ViewParallelTrainer: make the backward passes flag their rows when (and only when) the step is sparse"""
        if isinstance(collective, RcclCollective):
            self._tracker = collective.comm  # (RcclCollective.prepare switches the tracking itself)
            return
        if mode == "sparse" and self._tracker is None:
            self._tracker, self._own_tracker = api.Comm(self.r.ctx, 0, 1), True
        if self._tracker is not None:
            self._tracker.track_touched_rows(mode == "sparse")

    def close(self):
        if self._own_tracker and self._tracker is not None:
            self._tracker.close()
        self._tracker, self._own_tracker = None, False

    def sparse_touched_rows(self, grads: dict, world_size: int, rank: int):
        return self._tracker.sparse_touched_rows(int(grads["pos"].shape[0]), world_size)

    def sparse_pack(self, grads: dict, handle, first: int, count: int, msg):
        self._tracker.sparse_pack(grads, handle, first, count, msg)

    def sparse_accumulate(self, grads: dict, msg, count: int, rows=None):
        first, n = rows if rows is not None else (0, -1)
        self._tracker.sparse_accumulate(grads, msg, count, row_first=first, row_count=n)

    def flush(self):
        self.r.ctx.synchronize()


# ------------------------------------------------------------------------------------------------ protocol
class ViewParallelTrainer:
    """One training step per call: this rank's view -> forward + backward -> gradient collective -> optimiser."""

    def __init__(self, engine, collective, cameras: Sequence, grads: dict, mode: str = "allreduce",
                 views_per_step: int = 1):
        """views_per_step: views each rank renders (and whose gradients it accumulates) per optimiser step -- one
        collective per step, so B views per GPU amortise the gradient exchange B times."""
        if mode not in ("allreduce", "sharded", "sparse", "owner", "local"):
            raise ValueError(mode)
        if views_per_step < 1:
            raise ValueError("views_per_step must be >= 1")
        self.engine, self.coll, self.cameras, self.grads, self.mode = engine, collective, list(cameras), grads, mode
        self.views_per_step = views_per_step
        self.rank = collective.rank if collective is not None else 0
        self.world_size = collective.world_size if collective is not None else 1
        self.steps_done = 0
        if collective is not None and hasattr(collective, "prepare"):
            collective.prepare(mode)
        if collective is not None and hasattr(engine, "prepare"):
            engine.prepare(mode, collective)

    def camera_for_step(self, step: int):
        return self.cameras[view_of_rank(step, self.rank, self.world_size, len(self.cameras))]

    def step(self, dL_dimg, optimise: bool = True):
        """forward + backward of this rank's view, the collective, and (optimise=True) the Adam update."""
        if self.mode == "owner":  # splat ownership: the exchange is part of the frame itself
            if not optimise or self.views_per_step != 1:
                raise ValueError("mode 'owner': one view per rank and step, optimiser included")
            n = len(self.cameras)
            cams = [self.cameras[view_of_rank(self.steps_done, r, self.world_size, n)] for r in range(self.world_size)]
            self.steps_done += 1
            self.coll.owner_step(self.engine, cams, dL_dimg, self.grads, self.steps_done)
            return
        for j in range(self.views_per_step):  # this rank's views of the step: the first overwrites, the others add
            cam = self.camera_for_step(self.steps_done * self.views_per_step + j)
            self.engine.forward_backward(cam, dL_dimg, self.grads, accumulate=j > 0)
        self.steps_done += 1
        if self.mode == "local" or self.coll is None:
            if optimise:
                self.engine.adam(self.grads, self.steps_done)
            return
        if self.mode == "allreduce":
            self.coll.allreduce_grads(self.grads)
            if optimise:
                self.engine.adam(self.grads, self.steps_done)
        else:
            if not optimise:
                raise ValueError(f"mode '{self.mode}' fuses the collective with the optimiser step")
            if self.mode == "sparse":
                self.coll.sparse_adam(self.engine, self.grads, self.steps_done)
            else:
                self.coll.sharded_adam(self.engine, self.grads, self.steps_done)
