"""The data-parallel step's collective inside the library (ABI v16: pv_dist_*, pv_ivae_dp_step, pv_ved_dp_step; csrc/pv_dist.hip,
pyroved_amd/dist.py: NativeComm).  A one-GPU box can run RCCL at world size 1 only: these tests exercise the CALL PATH (library
resolution, communicator creation through ctypes, ncclAllReduce enqueued on the compute stream between the gradient launches and
the optimizer launch, the trainer and bench.py routes) — not the wire.  The sharding arithmetic itself is covered with two ranks
over gloo (tests/test_host_cpu.py, test_gpu_parity.py::test_trainer_data_parallel_two_ranks_one_gpu).

What is sharded: trainers/svi.py:104-113 of the reference (`self.svi.step(x)`), whose loss is a SUM over the data plate
(models/ivae.py:177,215)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def comm(gpu_device):
    from pyroved_amd import dist as pvdist
    c = pvdist.native_comm(torch.device("cuda", 0))
    assert (c.rank, c.world) == (0, 1) and "rccl" in c.library
    return c


def test_native_allreduce_world1_is_identity_on_the_current_stream(comm):
    t = torch.randn(152079, device="cuda")
    ref = t.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                 # the collective goes wherever the caller's current stream is
        t.mul_(2.0)
        comm.allreduce_sum_(t)
        t.mul_(0.5)
    side.synchronize()
    assert torch.equal(t, ref)
    with pytest.raises(Exception):
        comm.allreduce_sum_(torch.zeros(4))       # host memory: no CPU fallback


@pytest.mark.parametrize("kind", ["ivae_f3", "ivae_f2", "jivae", "convenc"])
def test_dp_step_world1_is_bit_identical_to_the_two_call_step(comm, kind):
    """pv_ivae_dp_step (loss_and_grads -> ncclAllReduce -> pv_adam_step_hist in one enqueue) against loss_and_grads() +
    adam_step() on the same inputs: at world size 1 the all-reduce is the identity, so parameters, Adam moments, zeroed
    gradients and the history slot must agree bit for bit over several steps."""
    import pyroved_amd as pv
    dims = (28, 28)
    if kind == "jivae":
        mk = lambda: pv.models.jiVAE((28, 28), 2, 10, ["r"], seed=1, device="cuda")
        b, fused = 32, 2
    elif kind == "convenc":
        dims = (32, 32)
        def mk():
            m = pv.models.iVAE((32, 32), 2, ["r", "t", "s"], seed=1, device="cuda")
            m.set_encoder(pv.nets.convEncoderNet((32, 32), latent_dim=m.z_dim, hidden_dim=[(32,), (64, 64)]))
            return m
        b, fused = 16, 2
    else:
        mk = lambda: pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
        b, fused = 256, int(kind[-1])
    ma, mb = mk(), mk()
    ea, eb = ma.engine(fused=fused), mb.engine(fused=fused)
    g = torch.Generator().manual_seed(3)
    hist_a, hist_b = torch.zeros(4, 4, device="cuda"), torch.zeros(4, 4, device="cuda")
    for i in range(4):
        x = torch.rand(b, *dims, generator=g).cuda()
        eps = torch.randn(b, ma.z_dim, generator=g).cuda()
        ea.loss_and_grads(x, eps)
        ea.adam_step_hist(hist_a[i])
        eb.loss_and_grads(x, eps, step=True, comm=comm, hist_out=hist_b[i])
    torch.cuda.synchronize()
    assert ea.adam_t == eb.adam_t == 4
    assert torch.equal(hist_a, hist_b) and torch.isfinite(hist_a).all() and (hist_a[:, 0] > 0).all()
    assert torch.equal(ea.flat, eb.flat) and torch.equal(ea.m, eb.m) and torch.equal(ea.v, eb.v)
    assert torch.equal(ea.grad[:ea.n_flat], eb.grad[:eb.n_flat]) and not ea.grad[:ea.n_flat].any()


def test_ved_dp_step_world1_is_bit_identical(comm):
    import pyroved_amd as pv
    mk = lambda: pv.models.VED((32, 32), (64,), hidden_dim_e=[(32,), (64, 64)], hidden_dim_d=[(64, 64), (32,)], seed=1, device="cuda")
    ma, mb = mk(), mk()
    ea, eb = ma.engine(fused=2), mb.engine(fused=2)
    g = torch.Generator().manual_seed(5)
    ha, hb = torch.zeros(3, 4, device="cuda"), torch.zeros(3, 4, device="cuda")
    for i in range(3):
        x = torch.rand(16, 1, 32, 32, generator=g).cuda()
        y = torch.rand(16, 1, 64, generator=g).cuda()
        eps = torch.randn(16, ma.z_dim, generator=g).cuda()
        ea.loss_and_grads(x, eps, 1.0, y)
        ea.adam_step_hist(ha[i])
        eb.loss_and_grads(x, eps, 1.0, y, step=True, comm=comm, hist_out=hb[i])
    torch.cuda.synchronize()
    assert torch.equal(ha, hb) and torch.isfinite(ha).all()
    assert torch.equal(ea.flat, eb.flat) and torch.equal(ea.m, eb.m) and torch.equal(ea.v, eb.v)


def test_dp_step_rejects_what_it_cannot_reduce(comm):
    import ctypes as C
    import pyroved_amd as pv
    from pyroved_amd import _abi
    m = pv.models.iVAE((28, 28), 2, ["r"], seed=1, device="cuda")
    eng = m.engine(fused=2)
    x, eps = torch.rand(8, 28, 28).cuda(), torch.randn(8, m.z_dim).cuda()
    with pytest.raises(ValueError):
        eng.loss_and_grads(x, eps, comm=comm)                      # comm without step=True
    with pytest.raises(ValueError):
        eng.loss_and_grads(x, eps, step=True, comm=comm, scalars_out=torch.zeros(4, device="cuda"))   # scalars outside the bucket
    assert _abi.lib().pv_dist_allreduce_sum(None, _abi.ptr(eng.grad), 4, _abi.current_stream()) == -1   # PV_EINVAL: no communicator
    r, w = C.c_int32(-1), C.c_int32(-1)
    assert _abi.lib().pv_dist_comm_info(comm.handle, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (0, 1)


def test_trainer_over_nccl_world1_uses_the_native_collective(gpu_device):
    """SVItrainer under an initialised nccl process group (world size 1 here) takes the native route — its history and weights
    must equal the plain single-process trainer's, bit for bit."""
    code = r'''
import json, os, sys, torch
import torch.distributed as td
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
sys.path.insert(0, %r)
import pyroved_amd as pv
torch.cuda.set_device(0)
def run(dist):
    if dist:
        td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    m = pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
    x = torch.rand(512, 28, 28, generator=torch.Generator().manual_seed(0))
    loader = pv.utils.init_dataloader(x, batch_size=128)
    tr = pv.trainers.SVItrainer(m, seed=1)
    native = getattr(tr, "_comm", None) is not None
    for _ in range(2):
        tr.step(loader)
    w = float(sum(p.double().sum() for p in m.state_dict().values()))
    if dist:
        td.destroy_process_group()
    return tr.loss_history["training_loss"], w, native
a = run(False)
class Forced:            # world size 1 never shards: make the trainer believe in two ranks' worth of code path
    pass
import pyroved_amd.dist as pvdist
real_world = pvdist.world
pvdist.world = lambda group=None: (0, 2) if td.is_initialized() else real_world(group)
pvdist.shard_bounds = lambda n, rank, world: (0, n)
pvdist.sync_replicas = lambda *a_, **k_: None
b = run(True)
print("RESULT " + json.dumps({"single": a[:2], "nccl": b[:2], "native": b[2], "single_native": a[2]}))
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert d["native"] is True and d["single_native"] is False
    np.testing.assert_array_equal(d["nccl"][0], d["single"][0])
    assert d["nccl"][1] == d["single"][1]


def test_bench_multirank_path_over_nccl_world1(gpu_device):
    """bench.py's N > 1 code path (process group over the nccl backend = RCCL, replica sync, the per-step collective, barriers,
    max-over-ranks timing) at world size 1 — the only form a one-GPU box can run it in: the line must parse, say which
    collective ran (the library's), carry its event-timed duration and the single-GPU ELBO."""
    env = dict(os.environ, PV_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29563")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--repeats", "2", "--no-configs",
           "--no-legs", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["collective"].startswith("rccl-native"), d["collective"]
    assert d["allreduce_ms"] > 0 and d["fp32_class"]["allreduce_ms"] > 0
    assert abs(d["fp32_class"]["loss_per_image_step0"] - 544.5358) < 0.02
    # the conv config's weak-scaling leg rides along as at N > 1: VED's one-call data-parallel step (pv_ved_dp_step) through bench.py
    assert d["c5_weak"]["value"] > 0 and d["c5_weak"]["allreduce_ms"] > 0, d.get("c5_weak")


def test_dp_step_captured_in_a_hipgraph_replays_bit_identically(comm):
    """pv_ivae_dp_step only ENQUEUES on the caller's stream (gradient launches, RCCL's all-reduce kernel, the optimizer launch): the
    whole data-parallel step can be captured in a hipGraph.  A captured-and-replayed step must equal the eager step bit for bit
    (scripts/dp_graph_capture.py is the stand-alone form)."""
    import pyroved_amd as pv
    mk = lambda: pv.models.iVAE((28, 28), 2, ["r", "t"], seed=1, device="cuda")
    ma, mb = mk(), mk()
    ea, eb = ma.engine(fused=3), mb.engine(fused=3)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(256, 28, 28, generator=g).cuda()
    eps = torch.randn(256, ma.z_dim, generator=g).cuda()
    ha, hb = torch.zeros(4, device="cuda"), torch.zeros(4, device="cuda")
    ea.loss_and_grads(x, eps, step=True, comm=comm, hist_out=ha)        # one eager step each: module load, RCCL's first call
    eb.loss_and_grads(x, eps, step=True, comm=comm, hist_out=hb)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            eb.loss_and_grads(x, eps, step=True, comm=comm, hist_out=hb)    # (adam_step = 2 is baked into the captured launch)
    torch.cuda.synchronize()
    eb.adam_t -= 1                       # capture does not execute
    graph.replay(); eb.adam_t += 1
    ea.loss_and_grads(x, eps, step=True, comm=comm, hist_out=ha)
    torch.cuda.synchronize()
    assert torch.equal(ea.flat, eb.flat) and torch.equal(ha, hb) and torch.isfinite(ha).all()
