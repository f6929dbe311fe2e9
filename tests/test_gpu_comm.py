"""`-m gpu`: the multi-GPU side of the C ABI (lcgs_comm_*) on the one GPU a box has -- a communicator of world
size 1 takes every call through RCCL itself (run-time binding, the communicator's stream, the slice events of the chunked
all-reduce, the reduce-scatter / all-gather of the sharded step); what a rank count above 1 adds is RCCL's own business.
The N > 1 host protocol is covered on CPU (tests/test_distributed.py, same classes, gloo)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import luisacomputegaussiansplatting_amd.multi_gpu as mg
from conftest import ROOT, make_scene
from gpu_util import DEV, upload_scene

pytestmark = pytest.mark.gpu

KEYS = mg.KEYS
POSE = ([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1])
LR = {"pos": 1.6e-4, "sh_dc": 2.5e-3, "sh_rest": 1.25e-4, "opacity": 5e-2, "scale": 5e-3, "rot": 1e-3}


def _raw_act(scene):
    act = upload_scene(scene)
    raw = {"pos": act["pos"], "scale": torch.log(act["scale"]), "rotq": act["rotq"] * 1.3, "sh": act["sh"],
           "opacity": torch.log(act["opacity"] / (1 - act["opacity"]))}
    return raw, act


def _grads_like(act, fill=7.0):
    return {k: torch.full_like(act[k], fill) for k in KEYS}


def test_sliced_backward_and_chunked_allreduce_leave_the_gradients_unchanged(lcgs, oracle):
    """With a communicator attached the dense backward runs its per-splat pass as splat-range slices (the chunks of
    lcgs_grads_allreduce): same gradients as the unsliced backward (up to the order of the render-backward's float
    atomics, which differs from run to run anyway), and a world of one sums to itself."""
    rng = np.random.default_rng(5)
    P = 30000
    scene = make_scene(rng, P, log_scale=(-3.8, 0.7))
    scene["pos"][:4000] += 100.0  # a run of culled splats: some slices hold few survivors
    cam = lcgs.get_lookat_cam(*POSE, width=400, height=300)
    dL = torch.randn(3, 300, 400, device=DEV)
    d = upload_scene(scene)
    plain = lcgs.Renderer(lcgs.Context(0))
    plain.bind_scene(*[d[k] for k in KEYS])
    img = torch.zeros(3, 300, 400, device=DEV)
    plain.forward(cam, img, keep_state=True)
    g0 = _grads_like(d)
    plain.backward(dL, *[g0[k] for k in KEYS])
    plain.ctx.synchronize()

    ctx = lcgs.Context(0)
    r = lcgs.Renderer(ctx)
    r.bind_scene(*[d[k] for k in KEYS])
    comm = lcgs.Comm(ctx, 0, 1)
    for _ in range(2):  # twice: slice events are re-recorded per backward
        g1 = _grads_like(d)
        r.forward(cam, img, keep_state=True)
        r.backward(dL, *[g1[k] for k in KEYS])
        comm.allreduce_grads(g1)
        ctx.synchronize()
        torch.cuda.synchronize()
        for k in KEYS:
            a, b = g1[k].double(), g0[k].double()
            assert float((a - b).norm() / b.norm()) <= 2e-4, k  # (float-atomic order; the bar is 1e-3)
            assert torch.equal(g0[k] == 0, g1[k] == 0), k  # the same rows are written, the same stay zero
    # an all-reduce of arrays no sliced backward wrote (one chunk behind the stream's tail) works too
    g2 = {k: g0[k].clone() for k in KEYS}
    comm.allreduce_grads(g2)
    ctx.synchronize()
    torch.cuda.synchronize()
    assert all(torch.equal(g0[k], g2[k]) for k in KEYS)
    # and against the oracle (BASELINE tolerance)
    ref = oracle.render_backward_full(scene, oracle.lookat(*POSE, width=400, height=300), dL.cpu().numpy())
    for k in KEYS:
        a, b = g1[k].cpu().numpy().astype(np.float64).ravel(), ref[k].astype(np.float64).ravel()
        assert np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30) <= 1e-3, k
    comm.close()


def test_f16_transport_is_opt_in_and_rounds_once(lcgs):
    """lcgs_comm_set_transport(F16): per-attribute power-of-two scales from the (max-reduced) magnitudes, f16 on the wire.
    At world size 1 the sum is the value itself, so what comes back is each gradient rounded once to f16 at its
    attribute's scale: zeros stay exact zeros, the relative error in the norm is a few 1e-4, nothing overflows -- also
    for magnitudes far outside f16's own range and for unaligned array offsets."""
    rng = np.random.default_rng(3)
    P = 30001  # odd: the arrays carved from one flat buffer start at unaligned offsets
    ctx = lcgs.Context(0)
    comm = lcgs.Comm(ctx, 0, 1)
    flat = torch.zeros(59 * P, device=DEV)
    widths = {"pos": 3, "scale": 3, "rotq": 4, "sh": 48, "opacity": 1}
    mags = {"pos": 1e-7, "scale": 3e4, "rotq": 1.0, "sh": 1e-3, "opacity": 5e9}
    g, o = {}, 0
    for k, w in widths.items():
        g[k] = flat[o:o + w * P].view(P, w) if w > 1 else flat[o:o + P]
        o += w * P
        vals = rng.normal(size=tuple(g[k].shape)) * mags[k] * 10.0 ** rng.uniform(-3, 0, size=tuple(g[k].shape))
        vals[rng.random(size=tuple(g[k].shape)) < 0.6] = 0.0  # most rows of a view's gradient are zero
        g[k].copy_(torch.from_numpy(vals.astype(np.float32)))
    before = {k: g[k].clone() for k in KEYS}
    comm.allreduce_grads(g)  # default transport: exact
    ctx.synchronize()
    torch.cuda.synchronize()
    assert all(torch.equal(g[k], before[k]) for k in KEYS)
    comm.set_transport("f16")
    comm.allreduce_grads(g)
    ctx.synchronize()
    torch.cuda.synchronize()
    for k in KEYS:
        a, b = g[k].double(), before[k].double()
        assert torch.isfinite(a).all(), k
        assert torch.equal(g[k] == 0, before[k] == 0) or float(((g[k] == 0) & (before[k] != 0)).float().mean()) < 1e-2, k
        assert (g[k][before[k] == 0] == 0).all(), k  # zeros stay exact zeros
        rel = float((a - b).norm() / b.norm())
        assert 0.0 < rel <= 6e-4, (k, rel)  # one f16 rounding (2^-11) of each value, nothing more
    comm.set_transport("f32")
    again = {k: g[k].clone() for k in KEYS}
    comm.allreduce_grads(g)
    ctx.synchronize()
    torch.cuda.synchronize()
    assert all(torch.equal(g[k], again[k]) for k in KEYS)
    comm.close()


@pytest.mark.parametrize("P", [4097, 20000])
def test_sharded_adam_step_equals_the_dense_step(lcgs, P):
    rng = np.random.default_rng(P)
    scene = make_scene(rng, P)
    results = []
    for sharded in (False, True):
        raw, act = _raw_act(scene)
        m = {k: torch.zeros_like(raw[k]) for k in KEYS}
        v = {k: torch.zeros_like(raw[k]) for k in KEYS}
        ctx = lcgs.Context(0)
        r = lcgs.Renderer(ctx)
        comm = lcgs.Comm(ctx, 0, 1) if sharded else None
        for step in (1, 2):
            g = {k: torch.from_numpy(np.random.default_rng(step).normal(size=tuple(raw[k].shape)).astype(np.float32)).to(DEV)
                 for k in KEYS}
            if sharded:
                comm.adam_step_sharded(g, raw, m, v, act, step, LR, eps=1e-8)
            else:
                r.adam_step(g, raw, m, v, act, step, LR, eps=1e-8)
        ctx.synchronize()
        torch.cuda.synchronize()
        results.append((raw, m, v, act))
        if comm is not None:
            comm.close()
    for a, b in zip(*results):
        for k in KEYS:
            assert torch.equal(a[k], b[k]), k


def test_view_parallel_trainer_on_the_hip_engine(lcgs):
    """The package protocol end to end on the GPU: ViewParallelTrainer + HipEngine + RcclCollective (world 1), both
    collective modes, against the same steps without any collective.  (Not bit for bit: the render-backward sums with
    float atomics, so two runs differ in the last place, and Adam turns the sign of a near-zero gradient into a step of
    +-lr; the bound below is that step, and almost every element has to agree far more closely.)"""
    rng = np.random.default_rng(11)
    scene = make_scene(rng, 20000, log_scale=(-3.8, 0.7))
    cams = [lcgs.get_lookat_cam([-3 * np.cos(a), -0.5 + 3 * np.sin(a), 2.3], [0, 0, 0.5], [0, 0, 1], width=320, height=240)
            for a in (0.0, 0.4, 0.8)]
    dL = torch.randn(3, 240, 320, device=DEV)
    outs = {}
    for mode in ("local", "allreduce", "sharded", "sparse"):
        raw, act = _raw_act(scene)
        ctx = lcgs.Context(0)
        eng = mg.HipEngine(lcgs.Renderer(ctx), raw, act, LR, eps=1e-8)
        coll = None if mode == "local" else mg.RcclCollective(ctx, 0, 1)
        tr = mg.ViewParallelTrainer(eng, coll, cams, _grads_like(act, 0.0), mode=mode)
        for _ in range(4):
            tr.step(dL)
        ctx.synchronize()
        torch.cuda.synchronize()
        outs[mode] = {k: raw[k].clone() for k in KEYS}
        if coll is not None:
            coll.close()
    assert not torch.equal(outs["local"]["opacity"], _raw_act(scene)[0]["opacity"])  # it trained
    lr_of = {"pos": LR["pos"], "scale": LR["scale"], "rotq": LR["rot"], "sh": LR["sh_dc"], "opacity": LR["opacity"]}
    for mode in ("allreduce", "sharded", "sparse"):
        for k in KEYS:
            diff = (outs[mode][k] - outs["local"][k]).abs()
            assert float(diff.max()) <= 2.02 * 4 * lr_of[k], (mode, k, float(diff.max()))
            assert float((diff > 0.05 * lr_of[k]).float().mean()) < 0.01, (mode, k)


def test_comm_argument_checks(lcgs):
    ctx = lcgs.Context(0)
    comm = lcgs.Comm(ctx, 0, 1)
    with pytest.raises(lcgs.LcgsError):
        lcgs.Comm(ctx, 0, 1)  # one communicator per context
    other = lcgs.Context(0)
    z = {k: torch.zeros(s, device=DEV) for k, s in zip(KEYS, ((8, 3), (8, 3), (8, 4), (8, 48), (8,)))}
    with pytest.raises(ValueError):
        lcgs.Comm(other, 1, 2)  # world > 1 needs a way to carry the rendezvous token
    comm.allreduce_grads(z)
    ctx.synchronize()
    comm.close()
    lcgs.Comm(ctx, 0, 1).close()  # detached again: a new one may be attached


@pytest.mark.parametrize("collective", ["rccl", "torch"])
def test_bench_runs_every_multi_gpu_code_path_on_one_rank(collective):
    """bench.py with LCGS_BENCH_FORCE_DIST=1: one rank, but through the process group and the gradient collectives -- every
    N > 1 branch of the file runs (the driver launches the real N = 2, 4, 8 on a whole node)."""
    env = dict(os.environ, LCGS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--splats", "200000", "--res", "640x480", "--steps",
                          "3", "--warmup", "1", "--no-cpu-baseline", "--no-stage-path", "--no-spatial", "--collective",
                          collective], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [x for x in res.stdout.splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    fb = out["fwd_bwd"]
    assert fb["value"] > 0 and fb["without_collective"]["value"] > 0 and fb["moving_camera"]["value"] > 0
    assert fb["views_per_gpu_and_step_4"]["value"] > 0 and fb["views_per_gpu_and_step_4"]["views_per_step"] == 4
    assert fb["multi_view_step_4"]["lcgs_fit_views"]["value"] > 0 and fb["multi_view_step_4"]["one_by_one"]["value"] > 0
    assert ("rccl" in fb["collective"]) == (collective == "rccl")
    assert set(out["train_step"]) == {"allreduce", "sharded", "sparse", "owner"}, out.get("leg_errors")
    assert all(v["value"] > 0 for v in out["train_step"].values())
    assert out["train_step"]["sparse"]["touched_rows"] > 0  # (one rank: every row is its own, nothing crosses the wire)
    assert out["moving_camera"]["value"] > 0 and len(out["moving_camera"]["per_view"]) == 8
    assert "leg_errors" not in out and "error" not in out


def test_bench_still_prints_its_line_when_a_multi_gpu_leg_fails():
    """N > 1 insurance: a failure in the legs behind the forward measurement is recorded in the line (`leg_errors`), the
    forward figure is still reported and the run ends with exit code 0."""
    env = dict(os.environ, LCGS_BENCH_FORCE_DIST="1", LCGS_BENCH_INJECT_LEG_FAILURE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29534")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--splats", "100000", "--res", "320x240", "--steps",
                          "2", "--warmup", "1", "--no-cpu-baseline", "--no-stage-path", "--no-batch"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [x for x in res.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["value"] > 0 and "injected" in out["leg_errors"]["fwd_bwd / train_step"]
    assert "fwd_bwd" not in out


def test_bench_falls_back_to_torch_distributed_when_the_communicator_fails_its_selftest():
    """the first N > 1 run's other insurance (round 6): a communicator that fails lcgs_comm_selftest is not given the timed
    legs -- they run over torch.distributed's own communicator, and the line says so"""
    env = dict(os.environ, LCGS_BENCH_FORCE_DIST="1", LCGS_BENCH_INJECT_SELFTEST_FAILURE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29536")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--splats", "100000", "--res", "320x240", "--steps",
                          "2", "--warmup", "1", "--no-cpu-baseline", "--no-stage-path", "--no-batch", "--no-spatial"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([x for x in res.stdout.splitlines() if x.startswith("{")][-1])
    st = out["comm_selftest"]
    assert st["ok"] is False and st["every_rank_ok"] is False and "torch.distributed" in st["fallback"]
    assert "injected" in out["leg_errors"]["comm_selftest"]
    assert out["fwd_bwd"]["value"] > 0 and "rccl" not in out["fwd_bwd"]["collective"]
    assert out["train_step"]["allreduce"]["value"] > 0


def test_bench_with_two_ranks_on_one_gpu_through_the_torch_collective():
    """Same launch with --collective torch: the process group (gloo here) carries the gradients, so every N > 1 leg -- the
    view-parallel trainer's all-reduce and sharded steps, accumulated views, lcgs_fit_views + gradient sum -- runs end to
    end with two real ranks (rates are meaningless: gloo stages through the host)."""
    env = dict(os.environ, LCGS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--splats",
                          "100000", "--res", "320x240", "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
                          "--no-stage-path", "--no-spatial", "--leg-timeout", "300", "--collective", "torch"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [x for x in res.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and "leg_errors" not in out and "error" not in out, out.get("leg_errors")
    fb = out["fwd_bwd"]
    assert fb["value"] > 0 and fb["without_collective"]["value"] > fb["value"] and fb["moving_camera"]["value"] > 0
    assert fb["views_per_gpu_and_step_4"]["views_per_step"] == 8 and fb["multi_view_step_4"]["lcgs_fit_views"]["value"] > 0
    assert set(out["train_step"]) == {"allreduce", "sharded", "sparse", "owner"}, out.get("leg_errors")
    assert all(v["value"] > 0 for v in out["train_step"].values())
    # two real ranks: the sparse step's reduce half carried only the rows the rank's view touched
    sp = out["train_step"]["sparse"]
    P_ = 100000
    assert 0 < sp["touched_rows"] < P_
    assert sp["xgmi_bytes_sent_per_gpu"] < out["train_step"]["sharded"]["xgmi_bytes_sent_per_gpu"]


def test_bench_with_two_ranks_on_one_gpu():
    """The driver's N > 1 launch, rehearsed with two ranks on the one GPU of the box (LCGS_BENCH_BACKEND=gloo; RCCL refuses
    two ranks on one device): torchrun, per-rank views, barriers, max-over-ranks timing, ONE line from rank 0 with the
    whole-job rate -- and the gradient legs: the library's communicator cannot be created here (every rank is refused), which
    is recorded in the line (`leg_errors.comm_create`) and the legs run over torch.distributed's communicator instead of
    ending or hanging the run."""
    env = dict(os.environ, LCGS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--splats",
                          "200000", "--res", "640x480", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                          "--no-stage-path", "--no-spatial", "--leg-timeout", "300"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [x for x in res.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "view-parallel x2"
    assert out["camera_batch"]["value"] > 0 and out["moving_camera"]["value"] > 0
    # the communicator of the gradient legs: refused on a shared device -> reported, the legs measured over the process group's
    # (or, a build that allows two ranks on a device: created, self-tested and used)
    assert out["fwd_bwd"]["value"] > 0, out.get("leg_errors")
    if "leg_errors" in out:
        assert set(out["leg_errors"]) == {"comm_create"} and "ncclCommInitRank" in out["leg_errors"]["comm_create"]
        assert out["comm_selftest"]["ok"] is False and "torch.distributed" in out["comm_selftest"]["fallback"]
        assert "rccl" not in out["fwd_bwd"]["collective"] and out["train_step"]["allreduce"]["value"] > 0
    else:
        assert out["comm_selftest"]["every_rank_ok"] is True


def test_bench_says_so_when_the_first_collectives_never_return():
    """N > 1: the process group's creation and the barriers around the forward measurement are a new node's first contact with
    RCCL.  If they do not return within --start-timeout there is nothing to report -- but rank 0 prints a line that says so
    (`value` null, `error`) and the run ends non-zero instead of hanging (here: a timeout no start can meet)."""
    env = dict(os.environ, LCGS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29545", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--splats",
                          "200000", "--res", "640x480", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                          "--start-timeout", "0.02"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode != 0
    lines = [x for x in res.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["value"] is None and out["n_gpus"] == 2 and "no forward measurement within" in out["error"]


def test_allreduce_issues_the_same_collectives_whether_or_not_the_rank_ran_a_backward(lcgs):
    """Collective discipline (RCCL: mismatched counts between ranks are undefined behaviour): the number and the row
    ranges of the chunks of lcgs_grads_allreduce depend on shared values only (P, the slice count), never on whether THIS
    rank ran a sliced backward before the call -- a rank without a view in the last round of a batch, or with an empty
    frame, zero-fills its arrays and must still issue what its peers issue."""
    rng = np.random.default_rng(5)
    scene = make_scene(rng, 9000)
    act = upload_scene(scene)
    ctx = lcgs.Context(0)
    r = lcgs.Renderer(ctx)
    r.bind_scene(*[act[k] for k in KEYS])
    comm = lcgs.Comm(ctx, 0, 1)
    cam = lcgs.get_lookat_cam(*POSE, width=160, height=120)
    img = torch.zeros(3, 120, 160, device=DEV)
    g = _grads_like(act, 0.0)
    # a rank that rendered and differentiated its view (sliced backward, chunks behind the slice events)
    r.forward(cam, img, keep_state=True, sync=True)
    r.backward(torch.ones(3, 120, 160, device=DEV), *[g[k] for k in KEYS])
    comm.allreduce_grads(g)
    ctx.synchronize()
    with_backward = comm.stats()
    want = {k: g[k].clone() for k in KEYS}
    # a rank that had no view this round: zero-filled arrays, no backward (and again with stale slice events of OTHER arrays)
    z = _grads_like(act, 0.0)
    comm.allreduce_grads(z)
    ctx.synchronize()
    without = comm.stats()
    assert with_backward["collective_groups"] == without["collective_groups"] == 4  # LCGS_GRAD_SLICES default, P >= 4096
    assert with_backward["bytes_sent"] == without["bytes_sent"]
    assert all(float(z[k].abs().max()) == 0.0 for k in KEYS)
    # work enqueued on the context's stream between the backward and the all-reduce is seen by the LAST chunk at least,
    # and the sums are what the backward wrote (world size 1: the identity)
    r.forward(cam, img, keep_state=True, sync=True)
    r.backward(torch.ones(3, 120, 160, device=DEV), *[g[k] for k in KEYS])
    comm.allreduce_grads(g)
    ctx.synchronize()
    for k in KEYS:
        assert torch.allclose(g[k], want[k], rtol=1e-4, atol=1e-6 * float(want[k].abs().max()))
    small = {k: torch.zeros(s, device=DEV) for k, s in zip(KEYS, ((8, 3), (8, 3), (8, 4), (8, 48), (8,)))}
    comm.allreduce_grads(small)  # P < 4096: one chunk on every rank
    assert comm.stats()["collective_groups"] == 1
    comm.close()


@pytest.mark.parametrize("world,P", [(4, 30001), (3, 5000), (8, 6)])
def test_sparse_exchange_stages_with_virtual_ranks(lcgs, world, P):
    """The device stages of the sparse gradient exchange (csrc/kernels/comm_sparse.hip: mark -> compact -> owner bounds ->
    pack -> accumulate), with `world` VIRTUAL ranks on the one GPU: every virtual rank differentiates its own view, hands
    each owner the touched rows of the owner's shard as one message, and the owner adds the messages to its own rows in
    rank order.  Against the same sum formed densely by torch in the same order: bit-identical on every shard row, and the
    touched lists are exactly the frames' on-screen rows.  (P = 6 < world: every row is a tail row, no message at all.)"""
    rng = np.random.default_rng(world * 1000 + P)
    scene = make_scene(rng, P, log_scale=(-3.6, 0.7))
    act = upload_scene(scene)
    ctx = lcgs.Context(0)
    r = lcgs.Renderer(ctx)
    r.bind_scene(*[act[k] for k in KEYS])
    comm = lcgs.Comm(ctx, 0, 1)  # the context's row tracker (its own world is 1; the exchange's world is `world`)
    comm.track_touched_rows(True)
    W, H = 200, 152
    img = torch.zeros(3, H, W, device=DEV)
    shard = P // world
    grads, msgs, touched = [], [], []
    for rk in range(world):
        a = 0.5 * rk
        cam = lcgs.get_lookat_cam([-3 * np.cos(a), -0.5 + 3 * np.sin(a), 2.3], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
        g = _grads_like(act, 0.0)
        r.forward(cam, img, keep_state=True, sync=True)
        r.backward(torch.randn(3, H, W, device=DEV), *[g[k] for k in KEYS])
        rows_addr, first = comm.sparse_touched_rows(P, world)
        assert len(first) == world + 2 and first[0] == 0 and all(first[i] <= first[i + 1] for i in range(world + 1))
        # the list is the frame's on-screen rows, ascending; its shard segments lie where the bounds say
        vis = r.visible_rows().cpu().numpy().astype(np.int64)
        assert first[-1] == len(vis) > 0
        out = {}
        for o in range(world):
            n = first[o + 1] - first[o]
            assert np.array_equal(np.flatnonzero((vis >= o * shard) & (vis < (o + 1) * shard)), np.arange(first[o], first[o + 1]))
            m = torch.zeros(lcgs.api.sparse_message_words(n), device=DEV)
            comm.sparse_pack(g, rows_addr, first[o], n, m)
            out[o] = (m, n)
        ctx.synchronize()
        for o in range(world):
            m, n = out[o]
            assert np.array_equal(m[:n].view(torch.int32).cpu().numpy(), vis[first[o]:first[o + 1]])
        grads.append(g)
        msgs.append(out)
        touched.append(vis)
        # a second query without a backward in between: the set was consumed
        assert comm.sparse_touched_rows(P, world)[1][-1] == 0
    for o in range(world):
        own = {k: grads[o][k].clone() for k in KEYS}
        ref = {k: grads[o][k].clone() for k in KEYS}
        for s_ in range(world):
            if s_ == o:
                continue
            m, n = msgs[s_][o]
            comm.sparse_accumulate(own, m, n)
            for k in KEYS:
                ref[k][o * shard:(o + 1) * shard] += grads[s_][k][o * shard:(o + 1) * shard]
        ctx.synchronize()
        for k in KEYS:
            assert torch.equal(own[k][o * shard:(o + 1) * shard], ref[k][o * shard:(o + 1) * shard]), (o, k)
            # rows outside the owner's shard were not touched by the messages
            assert torch.equal(own[k][:o * shard], grads[o][k][:o * shard])
            assert torch.equal(own[k][(o + 1) * shard:], grads[o][k][(o + 1) * shard:])
    # a message whose indices lie outside the accepted range (corrupt, or packed for another P) is dropped row by row,
    # never written through: restricted to shard 0, a message for the last shard changes nothing; an index beyond P neither
    if world >= 2 and shard > 0:
        m, n = msgs[0][world - 1]
        if n:
            own = {k: grads[1][k].clone() for k in KEYS}
            comm.sparse_accumulate(own, m, n, row_first=0, row_count=shard)
            bad = m.clone()
            bad[:n] = torch.full((n,), P + 12345, dtype=torch.int32, device=DEV).view(torch.float32)
            comm.sparse_accumulate(own, bad, n)
            ctx.synchronize()
            for k in KEYS:
                assert torch.equal(own[k], grads[1][k]), k
    comm.close()


def test_accumulated_views_touch_the_union_of_their_rows(lcgs):
    """lcgs_render_backward starts a new touched set, lcgs_render_backward_accumulate adds to it: after two views of one
    optimiser step the list is the union of both frames' on-screen rows."""
    rng = np.random.default_rng(77)
    P = 12000
    scene = make_scene(rng, P, log_scale=(-3.6, 0.7))
    act = upload_scene(scene)
    ctx = lcgs.Context(0)
    r = lcgs.Renderer(ctx)
    r.bind_scene(*[act[k] for k in KEYS])
    comm = lcgs.Comm(ctx, 0, 1)
    comm.track_touched_rows(True)
    W, H = 160, 120
    img = torch.zeros(3, H, W, device=DEV)
    g = _grads_like(act, 0.0)
    sets = []
    for j, a in enumerate((0.0, 1.2)):
        cam = lcgs.get_lookat_cam([-3 * np.cos(a), -0.5 + 3 * np.sin(a), 2.3], [0, 0, 0.5], [0, 0, 1], width=W, height=H)
        r.forward(cam, img, keep_state=True, sync=True)
        r.backward(torch.randn(3, H, W, device=DEV), *[g[k] for k in KEYS], accumulate=j > 0)
        sets.append(set(r.visible_rows().cpu().numpy().tolist()))
    addr, first = comm.sparse_touched_rows(P, 2)
    union = np.array(sorted(sets[0] | sets[1]), np.int64)
    assert len(sets[0] ^ sets[1]) > 0 and first[-1] == len(union)
    m = torch.zeros(lcgs.api.sparse_message_words(first[-1]), device=DEV)
    comm.sparse_pack(g, addr, 0, first[-1], m)
    ctx.synchronize()
    assert np.array_equal(m[:first[-1]].view(torch.int32).cpu().numpy(), union)
    comm.close()


@pytest.mark.parametrize("P", [4097, 20000])
def test_sparse_adam_step_at_world_size_one_equals_the_dense_step(lcgs, P):
    """lcgs_adam_step_sparse through RCCL itself at world size 1 (counts all-gather, empty exchange, Adam on the own rows
    = all rows, all-gather): bit-identical to lcgs_adam_step on the same gradients."""
    rng = np.random.default_rng(P)
    scene = make_scene(rng, P)
    results = []
    for sparse in (False, True):
        raw, act = _raw_act(scene)
        m = {k: torch.zeros_like(raw[k]) for k in KEYS}
        v = {k: torch.zeros_like(raw[k]) for k in KEYS}
        ctx = lcgs.Context(0)
        r = lcgs.Renderer(ctx)
        comm = lcgs.Comm(ctx, 0, 1) if sparse else None
        if sparse:
            with pytest.raises(lcgs.LcgsError):  # tracking is opt-in and must be on before the step
                comm.adam_step_sparse(_grads_like(act, 0.0), raw, m, v, act, 1, LR, eps=1e-8)
            comm.track_touched_rows(True)
        for step in (1, 2):
            g = {k: torch.from_numpy(np.random.default_rng(step).normal(size=tuple(raw[k].shape)).astype(np.float32)).to(DEV)
                 for k in KEYS}
            if sparse:
                comm.adam_step_sparse(g, raw, m, v, act, step, LR, eps=1e-8)
                assert comm.stats()["touched_rows"] == 0  # no backward ran: nothing was flagged (and nothing needed: N = 1)
            else:
                r.adam_step(g, raw, m, v, act, step, LR, eps=1e-8)
        ctx.synchronize()
        torch.cuda.synchronize()
        results.append((raw, m, v, act))
        if comm is not None:
            comm.close()
    for a, b in zip(*results):
        for k in KEYS:
            assert torch.equal(a[k], b[k]), k


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 on one GPU: the in-process loopback transport (lcgs_comm_create_loopback) behind every collective of the C ABI.
# N contexts, one host thread each; what differs from a node is only who copies the bytes (device-to-device copies ordered
# by events instead of RCCL over xGMI): chunking, shard / tail / message arithmetic and stream ordering are the shipped code.
# ---------------------------------------------------------------------------------------------------------------------
def _run_ranks(world, rank_main):
    """rank_main(rank, side_stream) in `world` host threads; returns the per-rank results, raises what a rank raised"""
    import threading

    out, errors = [None] * world, []

    def run(me):
        try:
            side = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(side):
                out[me] = rank_main(me, side)
                side.synchronize()
        except Exception as e:  # noqa: BLE001
            import traceback

            errors.append((me, repr(e), traceback.format_exc()))

    torch.cuda.synchronize()
    threads = [threading.Thread(target=run, args=(me,)) for me in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank hangs"
    return out


@pytest.mark.parametrize("world,P", [(2, 4097), (3, 20000), (8, 4099)])
def test_allreduce_and_sharded_step_with_n_ranks_in_process(lcgs, world, P):
    """lcgs_grads_allreduce and lcgs_adam_step_sharded with N participants: every rank ends with the rank-ordered sum of
    the N gradient sets (bit for bit: the loopback adds in rank order), and the sharded step -- reduce-scatter, Adam on the
    own rows + the P mod N tail, all-gather of the activated rows -- leaves every rank with the dense step's activated
    arrays and, on its own rows and the tail, the dense step's raw parameters and moments."""
    from functools import reduce

    rng = np.random.default_rng(P + world)
    scene = make_scene(rng, P)
    gsets = [{k: torch.from_numpy(np.random.default_rng(100 * r + i).normal(size=tuple(upload_scene(scene)[k].shape)).astype(np.float32)).to(DEV)
              for i, k in enumerate(KEYS)} for r in range(world)]
    gsum = {k: reduce(lambda a, b: a + b, [gsets[r][k] for r in range(world)]) for k in KEYS}
    # the dense reference: one context, the summed gradients
    raw0, act0 = _raw_act(scene)
    m0 = {k: torch.zeros_like(raw0[k]) for k in KEYS}
    v0 = {k: torch.zeros_like(raw0[k]) for k in KEYS}
    ref = lcgs.Renderer(lcgs.Context(0))
    ref.adam_step(gsum, raw0, m0, v0, act0, 1, LR, eps=1e-8)
    ref.ctx.synchronize()
    group = lcgs.api.LoopbackGroup(world)

    def rank_main(me, side):
        ctx = lcgs.Context(0, side.cuda_stream)
        comm = lcgs.Comm(ctx, me, world, loopback=group)
        g = {k: gsets[me][k].clone() for k in KEYS}
        comm.allreduce_grads(g)
        ctx.synchronize()
        summed = {k: g[k].clone() for k in KEYS}
        raw, act = _raw_act(scene)
        m = {k: torch.zeros_like(raw[k]) for k in KEYS}
        v = {k: torch.zeros_like(raw[k]) for k in KEYS}
        g2 = {k: gsets[me][k].clone() for k in KEYS}
        comm.adam_step_sharded(g2, raw, m, v, act, 1, LR, eps=1e-8)
        ctx.synchronize()
        st = comm.stats()
        comm.close()
        return summed, raw, m, v, act, st

    res = _run_ranks(world, rank_main)
    group.close()
    for me, (summed, raw, m, v, act, st) in enumerate(res):
        first, count = lcgs.shard_rows(P, world, me)
        tail0 = count * world
        for k in KEYS:
            assert torch.equal(summed[k], gsum[k]), (me, k)
            assert torch.equal(act[k], act0[k]), (me, k)  # complete on every rank after the all-gather
            for mine, want in ((raw, raw0), (m, m0), (v, v0)):
                assert torch.equal(mine[k][first:first + count], want[k][first:first + count]), (me, k)
                assert torch.equal(mine[k][tail0:], want[k][tail0:]), (me, k)
        assert st["collective_groups"] >= 1


def test_view_parallel_trainer_with_three_ranks_in_process(lcgs):
    """ViewParallelTrainer + HipEngine + RcclCollective over the loopback, three ranks x three views, every mode: the
    rendered views' gradients are summed by the chunked all-reduce behind the sliced backward ("allreduce"), reduce-scattered
    ("sharded"), sent as touched rows to their owners ("sparse"), or never formed densely at all ("owner").  Reference: one
    context that renders the three views one after the other, accumulates and runs dense Adam.  (Bounds as in the world-size-1
    test above: float atomics make two runs differ in the last place, Adam turns a near-zero gradient's sign into +-lr.)"""
    world, steps = 3, 3
    rng = np.random.default_rng(12)
    scene = make_scene(rng, 20000, log_scale=(-3.8, 0.7))
    cams = [lcgs.get_lookat_cam([-3 * np.cos(a), -0.5 + 3 * np.sin(a), 2.3], [0, 0, 0.5], [0, 0, 1], width=320, height=240)
            for a in (0.0, 0.4, 0.8)]
    dL = torch.randn(3, 240, 320, device=DEV)
    # reference: every step renders all three views (rank r's view at step s is cams[(s * 3 + r) % 3]: all of them)
    raw_ref, act_ref = _raw_act(scene)
    ctx = lcgs.Context(0)
    eng = mg.HipEngine(lcgs.Renderer(ctx), raw_ref, act_ref, LR, eps=1e-8)
    g = _grads_like(act_ref, 0.0)
    for s in range(steps):
        for j, cam in enumerate(cams):
            eng.forward_backward(cam, dL, g, accumulate=j > 0)
        eng.adam(g, s + 1)
    ctx.synchronize()
    lr_of = {"pos": LR["pos"], "scale": LR["scale"], "rotq": LR["rot"], "sh": LR["sh_dc"], "opacity": LR["opacity"]}
    for mode in ("allreduce", "sharded", "sparse", "owner"):
        group = lcgs.api.LoopbackGroup(world)

        def rank_main(me, side, mode=mode, group=group):
            raw, act = _raw_act(scene)
            c = lcgs.Context(0, side.cuda_stream)
            e = mg.HipEngine(lcgs.Renderer(c), raw, act, LR, eps=1e-8)
            coll = mg.RcclCollective(c, me, world, loopback=group)
            tr = mg.ViewParallelTrainer(e, coll, cams, _grads_like(act, 0.0), mode=mode)
            for _ in range(steps):
                tr.step(dL)
            c.synchronize()
            out = {k: raw[k].clone() for k in KEYS}, {k: act[k].clone() for k in KEYS}
            e.close()
            coll.close()
            return out

        res = _run_ranks(world, rank_main)
        group.close()
        for me, (raw, act) in enumerate(res):
            if mode == "owner":
                first, count = lcgs.api.owner_rows(20000, world, me)
            elif mode == "allreduce":
                first, count = 0, 20000
            else:
                first, count = lcgs.shard_rows(20000, world, me)
            for k in KEYS:
                diff = (raw[k][first:first + count] - raw_ref[k][first:first + count]).abs()
                assert float(diff.max()) <= 2.02 * steps * lr_of[k], (mode, me, k, float(diff.max()))
                assert float((diff > 0.05 * lr_of[k]).float().mean()) < 0.01, (mode, me, k)
                if mode != "owner":  # every rank holds the whole refreshed scene (the ownership step replicates nothing)
                    da = (act[k] - act_ref[k]).abs()
                    scale = max(float(act_ref[k].abs().max()), 1e-6)
                    assert float((da > 1e-3 * scale).float().mean()) < 0.01, (mode, me, k)


# ---------------------------------------------------------------------------------------------------------------------
# lcgs_comm_selftest: what a communicator says about itself before anything is timed (round 6)
# ---------------------------------------------------------------------------------------------------------------------
def test_comm_selftest_through_rccl_at_world_size_one(lcgs):
    """all three phases through RCCL's own code path (the point-to-point phase sends to itself), on a context that has a
    scene of its own bound: the binding -- and the frame it renders -- are the same afterwards"""
    from conftest import make_scene

    rng = np.random.default_rng(5)
    scene = make_scene(rng, 30_000)
    r = lcgs.Renderer(lcgs.Context(0))
    r.upload_scene(scene)
    cam = lcgs.get_lookat_cam([-3, -0.5, 2.3], [0, 0, 0.5], [0, 0, 1], width=320, height=240)
    before = torch.zeros(3, 240, 320, device=DEV)
    n0 = r.forward(cam, before, sync=True)
    comm = lcgs.Comm(r.ctx, 0, 1)
    try:
        rep = comm.selftest(timeout_s=60.0)
        assert rep["ok"] and rep["allreduce_ok"] == 1 and rep["p2p_ok"] == 1 and rep["owner_step_ok"] == 1, rep
        assert rep["timed_out"] == 0 and rep["owner_max_grad_err"] <= 1e-4 and rep["world_size"] == 1, rep
        after = torch.zeros(3, 240, 320, device=DEV)
        assert r.forward(cam, after, sync=True) == n0 and torch.equal(before, after)
        assert r.verify_derived() == 0
    finally:
        comm.close()


@pytest.mark.parametrize("world", [2, 8])
def test_comm_selftest_with_n_ranks_in_process(lcgs, world):
    """the same C code with N participants over the in-process loopback: every rank's three phases pass"""
    import threading

    group = lcgs.api.LoopbackGroup(world)
    out, errors = [None] * world, []

    def rank_main(me):
        try:
            side = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(side):
                r = lcgs.Renderer(lcgs.Context(0, side.cuda_stream))
                comm = lcgs.Comm(r.ctx, me, world, loopback=group)
                out[me] = comm.selftest(timeout_s=120.0, check=False)
                comm.close()
        except Exception as e:  # noqa: BLE001
            errors.append((me, repr(e)))

    torch.cuda.synchronize()
    threads = [threading.Thread(target=rank_main, args=(me,)) for me in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank hangs"
    group.close()
    for me in range(world):
        rep = out[me]
        assert rep["ok"] and rep["rank"] == me and rep["world_size"] == world, rep
        assert rep["allreduce_ok"] == 1 and rep["p2p_ok"] == 1 and rep["owner_step_ok"] == 1 and rep["timed_out"] == 0, rep
        assert rep["owner_max_grad_err"] <= 1e-4, rep


def test_loopback_group_rejects_a_rank_taken_twice(lcgs):
    """(round-5 advisor) two communicators for one rank of a loopback group would share a mailbox; the rank is free again once
    its communicator is gone"""
    group = lcgs.api.LoopbackGroup(2)
    a, b = lcgs.Renderer(lcgs.Context(0)), lcgs.Renderer(lcgs.Context(0))
    c0 = lcgs.Comm(a.ctx, 0, 2, loopback=group)
    with pytest.raises(lcgs.LcgsError):
        lcgs.Comm(b.ctx, 0, 2, loopback=group)
    c0.close()
    c0 = lcgs.Comm(b.ctx, 0, 2, loopback=group)
    c0.close()
    group.close()
