"""View-parallel training across the GPUs of one node: the host protocol.

One process per GPU.  The scene is replicated on every rank; a batch of views is sharded one view per rank; each rank
runs forward + backward for its view (no data-path collective), then the dense per-splat gradients are summed over
the ranks and the optimiser step is applied -- in one of two exact (f32) ways:

  mode "allreduce"   lcgs_grads_allreduce (chunked, overlapping the backward's tail) + a dense lcgs_adam_step on
                     every rank: every rank does the whole optimiser's work on identical inputs.
  mode "sharded"     lcgs_adam_step_sharded: reduce-scatter -> Adam on the rank's own rows (+ the < N tail rows) ->
                     all-gather of the refreshed activated arrays.  Same bytes on the wire, 1/N of the optimiser work.

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


# ------------------------------------------------------------------------------------------------ collectives
class RcclCollective:
    """The product path: RCCL over xGMI through the C ABI (lcgs_comm_*), on the communicator's own HIP stream."""

    name = "rccl (lcgs_comm C ABI)"

    def __init__(self, ctx: "api.Context", rank: int, world_size: int, exchange: Optional[Callable] = None):
        self.comm = api.Comm(ctx, rank, world_size, exchange)
        self.rank, self.world_size = rank, world_size

    def allreduce_grads(self, grads: dict):
        self.comm.allreduce_grads(grads)

    def sharded_adam(self, engine, grads: dict, step: int):
        engine.adam_sharded(self.comm, grads, step)

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

    def close(self):
        pass


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

    def adam_sharded(self, comm: "api.Comm", grads: dict, step: int):
        comm.adam_step_sharded(grads, self.raw, self.m, self.v, self.activated, step, self.lr, self.betas, self.eps)


# ------------------------------------------------------------------------------------------------ protocol
class ViewParallelTrainer:
    """One training step per call: this rank's view -> forward + backward -> gradient collective -> optimiser."""

    def __init__(self, engine, collective, cameras: Sequence, grads: dict, mode: str = "allreduce",
                 views_per_step: int = 1):
        """views_per_step: views each rank renders (and whose gradients it accumulates) per optimiser step -- one
        collective per step, so B views per GPU amortise the gradient exchange B times."""
        if mode not in ("allreduce", "sharded", "local"):
            raise ValueError(mode)
        if views_per_step < 1:
            raise ValueError("views_per_step must be >= 1")
        self.engine, self.coll, self.cameras, self.grads, self.mode = engine, collective, list(cameras), grads, mode
        self.views_per_step = views_per_step
        self.rank = collective.rank if collective is not None else 0
        self.world_size = collective.world_size if collective is not None else 1
        self.steps_done = 0

    def camera_for_step(self, step: int):
        return self.cameras[view_of_rank(step, self.rank, self.world_size, len(self.cameras))]

    def step(self, dL_dimg, optimise: bool = True):
        """forward + backward of this rank's view, the collective, and (optimise=True) the Adam update."""
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
                raise ValueError("mode 'sharded' fuses the collective with the optimiser step")
            self.coll.sharded_adam(self.engine, self.grads, self.steps_done)
