"""
CPU tests of the host side: the C-ABI library loads and exports every symbol the header
declares (no compute calls), the reference-API mirror behaves like the reference's own shape /
type tests (tests/test_models.py:50-97 of the reference), the product refuses to compute on the
CPU, and the trainer's host logic — data order, eps stream, evaluate semantics, and the
data-parallel shard / all-reduce path with world_size 2 over gloo — reproduces the reference
trainer's numbers when driven with a stand-in engine.
"""
import ctypes as C
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, check_digest

import pyroved_amd as pv
from pyroved_amd import _abi, dist as pvdist
from oracle import svi_oracle as orc
from _oracle_engine import OracleEngine

tt = torch.tensor


# ------------------------------------------------------------------------------- C ABI
def header_functions():
    src = open(os.path.join(ROOT, "include", "pyroved_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pv_[a-z_0-9]+)\s*\(", src)))


def test_abi_exports_every_header_symbol():
    names = header_functions()
    assert "pv_ivae_loss_and_grads" in names and "pv_adam_step" in names and len(names) >= 11
    assert sorted(_abi.SIGNATURES.keys()) == names, "ctypes binding and header disagree"
    lib = _abi.lib()           # raises if the .so is missing or incomplete
    for n in names:
        assert hasattr(lib, n)
    assert lib.pv_version() == _abi.PV_ABI_VERSION


def test_plan_struct_matches_header_layout():
    """sizeof/offsets of the ctypes mirror vs the C struct, through a compiled probe."""
    import subprocess
    import tempfile
    probe = r'''
#include <stdio.h>
#include <stddef.h>
#include "pyroved_amd.h"
int main() {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(pv_layer), sizeof(pv_ivae_plan), offsetof(pv_ivae_plan, enc),
         offsetof(pv_ivae_plan, out), offsetof(pv_ivae_plan, params), offsetof(pv_ivae_plan, scalars),
         offsetof(pv_ivae_plan, adam_step));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(probe)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o",
                               os.path.join(d, "p")])
        out = subprocess.check_output([os.path.join(d, "p")]).split()
    got = [int(v) for v in out]
    P = _abi.pv_ivae_plan
    want = [C.sizeof(_abi.pv_layer), C.sizeof(P), P.enc.offset, P.out.offset, P.params.offset, P.scalars.offset,
            P.adam_step.offset]
    assert got == want


def test_workspace_query_runs_without_gpu():
    """pv_ivae_workspace_bytes is pure host arithmetic: callable on the CPU box."""
    lib = _abi.lib()
    assert lib.pv_linear_workspace_bytes(200704, 128, 128) > 0
    assert lib.pv_linear_workspace_bytes(-1, 128, 128) < 0


def _headline_plan(batch, fused):
    """BASELINE configs[1] by hand (offsets are irrelevant to the sizing arithmetic)."""
    p = _abi.pv_ivae_plan()
    p.batch, p.n_pix, p.coord_dim, p.z_dim, p.latent_dim = batch, 784, 2, 5, 2
    p.has_r = p.has_t = 1
    p.lik, p.sigmoid_out, p.fused = _abi.LIK["bernoulli"], 1, fused
    def layer(i, o, act):
        l = _abi.pv_layer()
        l.in_dim, l.out_dim, l.act, l.b_off = i, o, _abi.ACT[act], 0
        return l
    p.n_enc = 2
    p.enc[0], p.enc[1] = layer(784, 128, "tanh"), layer(128, 128, "tanh")
    p.head = layer(128, 10, None)
    p.fc_coord, p.fc_latent = layer(2, 128, "tanh"), layer(2, 128, None)
    p.n_dec = 2
    p.dec[0], p.dec[1] = layer(128, 128, "tanh"), layer(128, 128, "tanh")
    p.out = layer(128, 1, None)
    return p


def test_workspace_by_purpose():
    """pv_ivae_workspace_bytes_for: the fused training step needs a small fraction of what a layered decode of the same
    batch needs, the fused forward-only decode (round 3) a few KB per image; PV_WS_ALL (= pv_ivae_workspace_bytes) covers
    all three."""
    lib = _abi.lib()
    for fused in (2, 3):
        p = _headline_plan(4096, fused)
        step, enc, dec = (lib.pv_ivae_workspace_bytes_for(C.byref(p), w) for w in (1, 2, 3))
        allb = lib.pv_ivae_workspace_bytes(C.byref(p))
        assert min(step, enc, dec) > 0 and allb == max(step, enc, dec) == lib.pv_ivae_workspace_bytes_for(C.byref(p), 0)
        assert enc < 64 << 20
        # step: ~0.1 MB per image (per-row outputs + per-sample partials), not the layered path's ~1.9 MB
        assert step < 4096 * 200_000
        # decode on the fused kernel: hz + transform parameters + the weight images — no (B N) x 128 activations
        assert dec < 4096 * 1024 + (1 << 20)
        p.fused = 0
        dec_layered = lib.pv_ivae_workspace_bytes_for(C.byref(p), 3)
        assert step < dec_layered // 4 and dec < dec_layered // 100
        p.fused = fused
    assert lib.pv_ivae_workspace_bytes_for(C.byref(p), 4) < 0
    p.fused = 0
    assert lib.pv_ivae_workspace_bytes_for(C.byref(p), 1) > 4096 * 1_000_000


def test_viz_mosaic_and_plots():
    """utils.viz: the mosaic layout is make_grid's (padding before every cell + a closing border, nrow per row,
    pad_value fill; pyroved/utils/viz.py:16-18, 66-69) and the three plot helpers run headless."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from pyroved_amd.utils import viz
    imgs = torch.arange(5 * 3 * 4, dtype=torch.float32).reshape(5, 3, 4) + 1
    g = viz.tile_images(imgs, 2, padding=2, pad_value=-1)
    assert g.shape == (3 * 5 + 2, 2 * 6 + 2)
    assert (g[:2] == -1).all() and (g[:, :2] == -1).all() and (g[-2:] == -1).all()
    assert torch.equal(g[2:5, 2:6], imgs[0]) and torch.equal(g[2:5, 8:12], imgs[1])
    assert torch.equal(g[7:10, 2:6], imgs[2]) and torch.equal(g[12:15, 2:6], imgs[4])
    assert (g[12:15, 8:12] == -1).all()                    # the empty sixth cell
    assert torch.equal(viz.tile_images(imgs[:, None], 2, 2, -1), g)
    with pytest.raises(AssertionError):
        viz.plot_img_grid(torch.zeros(4, 4), 2)
    viz.plot_img_grid(torch.rand(4, 6, 6), 2, extent=[torch.tensor(-1.), torch.tensor(1.), torch.tensor(-2.), torch.tensor(2.)])
    viz.plot_spect_grid(torch.rand(4, 1, 9), 2, ylim=[0, 1])
    viz.plot_grid_traversal(torch.rand(9, 6, 6), 3, (6, 6), 2)
    assert len(plt.get_fignums()) == 3
    plt.close("all")


def test_trainer_epoch_order_and_noise_match_a_dataloader_pass():
    """SVItrainer's epoch prologue (trainers/svi.py: _epoch_batches, _draw_eps_epoch, the background permutation for
    the next epoch) against what `for data in loader: eps = torch.empty(b, z).normal_()` does: same minibatch order,
    same eps, same state of the global CPU generator afterwards — for shuffled / sequential / drop_last / ragged
    loaders over consecutive epochs, with and without a correctly predicted prefetch."""
    from torch.utils.data import DataLoader, TensorDataset
    from pyroved_amd.trainers.svi import SVItrainer

    class _M:
        z_dim = 5
    tr = SVItrainer.__new__(SVItrainer)
    tr.model, tr.rng = _M(), "cpu"
    x = torch.rand(1003, 4)
    hits = 0
    for kw in (dict(batch_size=100, shuffle=True), dict(batch_size=100, shuffle=False),
               dict(batch_size=64, shuffle=True, drop_last=True), dict(batch_size=7, shuffle=True),
               dict(batch_size=2048, shuffle=True)):
        loader = DataLoader(TensorDataset(x), **kw)
        torch.manual_seed(11)
        ref, ref_eps = [], []
        for ep in range(3):                                    # consecutive epochs on one generator stream
            b_ = [b for b in DataLoader(range(1003), batch_sampler=loader.batch_sampler)]
            ref.append(b_)
            ref_eps.append(torch.cat([torch.empty(len(b), 5).normal_() for b in b_]))
        st = torch.get_rng_state()
        torch.manual_seed(11)
        tr._perm_job = None
        for ep in range(3):
            job = getattr(tr, "_perm_job", None)
            got = tr._epoch_batches(loader, 1003)
            if job is not None:
                hits += int(bool(job[2]) and kw["shuffle"] and torch.equal(torch.cat(got)[:len(job[2][0])][:10], job[2][0][:10]))
            eps = tr._draw_eps_epoch([len(b) for b in got])
            if kw["shuffle"]:
                tr._prefetch_perm(1003)
            assert len(got) == len(ref[ep]) and all(torch.equal(a, b) for a, b in zip(got, ref[ep])), (kw, ep)
            assert torch.equal(eps, ref_eps[ep]), (kw, ep)
        assert torch.equal(st, torch.get_rng_state()), kw
        if tr._perm_job is not None:
            tr._perm_job[0].join()
    assert hits >= 6                                            # the predicted permutations were the ones used
    # a wrong prediction (someone drew from the generator in between) is discarded
    loader = DataLoader(TensorDataset(x), batch_size=100, shuffle=True)
    torch.manual_seed(5)
    tr._prefetch_perm(1003)
    torch.rand(3)
    st0 = torch.get_rng_state()
    got = tr._epoch_batches(loader, 1003)
    torch.set_rng_state(st0)
    want = [b for b in DataLoader(range(1003), batch_sampler=loader.batch_sampler)]
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    # custom sampler: the generic path
    from torch.utils.data import SubsetRandomSampler
    l3 = DataLoader(TensorDataset(x), batch_size=50, sampler=SubsetRandomSampler(range(0, 1003, 2)))
    torch.manual_seed(2)
    want = [b for b in DataLoader(range(1003), batch_sampler=l3.batch_sampler)]
    torch.manual_seed(2)
    got = tr._epoch_batches(l3, 1003)
    assert all(torch.equal(a, b) for a, b in zip(got, want))


def test_gp_helper_matches_closed_forms():
    """utils.gp_model (the torch restatement of pyro.contrib.gp's GPRegression + RBF behind predict_on_latent; Pyro is
    absent, so this is pinned to the textbook formulas only): the loss is -log N(y | 0, K + (noise + jitter) I), the
    predictive mean K_*f (K_ff + ..)^-1 y, variances non-negative and ~0 at noiseless training points for tiny noise,
    and one Adam(lr=0.005) step moves each log-parameter by 0.005."""
    from torch.distributions import MultivariateNormal
    from pyroved_amd.utils.gp import GPRegression, gp_model
    torch.manual_seed(0)
    X, y = torch.randn(7, 3), torch.randn(7)
    g = GPRegression(X, y)
    K = g.kernel(X) + (1e-6 + 1.0) * torch.eye(7)
    want = -MultivariateNormal(torch.zeros(7), covariance_matrix=K).log_prob(y)
    assert abs(float(g.loss()) - float(want)) < 1e-4
    with torch.no_grad():
        mean, var = g(X)
        np.testing.assert_allclose(mean.numpy(), (g.kernel(X) @ torch.linalg.solve(K, y)).numpy(), atol=1e-5)
        assert (var >= 0).all()
        _, cov = g(X, full_cov=True)
        np.testing.assert_allclose(torch.diagonal(cov).numpy(), var.numpy(), atol=1e-5)
        g.log_noise.fill_(-12.0)
        m2, v2 = g(X)
        np.testing.assert_allclose(m2.numpy(), y.numpy(), atol=2e-3)
        assert float(v2.max()) < 1e-3
    t = gp_model(3, X, y, gp_iterations=1)
    for q in t.parameters():
        assert abs(abs(float(q)) - 0.005) < 1e-6
    t3 = gp_model(3, X, y, gp_iterations=3)
    assert float(t3.loss()) < float(GPRegression(X, y).loss())


# ------------------------------------------------------------------------------- API mirror
@pytest.mark.parametrize("invariances, coord_exp", [(None, 0), (['t'], 1)])
def test_base_vae_1d(invariances, coord_exp):
    m = pv.models.baseVAE((8,), invariances, device="cpu")
    assert m.coord == coord_exp


@pytest.mark.parametrize("invariances, coord_exp",
                         [(None, 0), (['r'], 1), (['t'], 2), (['s'], 1), (['r', 's', 't'], 4)])
def test_base_vae_2d(invariances, coord_exp):
    m = pv.models.baseVAE((8, 8), invariances, device="cpu")
    assert m.coord == coord_exp


@pytest.mark.parametrize("invariances", [['r'], ['s'], ['r', 't']])
def test_base_vae_1d_exception(invariances):
    with pytest.raises(ValueError):
        pv.models.baseVAE((8,), invariances, device="cpu")


def test_split_latent_shapes():
    z = torch.randn(5, 3)
    m = pv.models.baseVAE((8,), ['t'], device="cpu")
    phi, dx, sc, zc = m._split_latent(z)
    assert phi is None and sc is None and dx.shape == (5, 1) and zc.shape == (5, 2)
    m = pv.models.baseVAE((8, 8), ['r', 't', 's'], device="cpu")
    z = torch.randn(5, 6)
    phi, dx, sc, zc = m._split_latent(z)
    assert phi.shape == (5,) and dx.shape == (5, 2) and sc.shape == (5,) and zc.shape == (5, 2)
    o = orc.split_latent(orc.Config((8, 8), 2, ['r', 't', 's']), z)
    for a, b in zip((phi, dx, sc, zc), o):
        assert torch.equal(a, b)


@pytest.mark.parametrize("invariances, n_params", [(['r'], 151559), (['r', 't'], 152075)])
def test_ivae_parameter_inventory(invariances, n_params):
    """parameter count and state_dict keys of the 28x28 models."""
    m = pv.models.iVAE((28, 28), 2, invariances, device="cpu")
    assert sum(p.numel() for p in m.parameters()) == n_params
    assert list(m.state_dict().keys()) == [
        'encoder_z.fc_layers.0.weight', 'encoder_z.fc_layers.0.bias', 'encoder_z.fc_layers.2.weight',
        'encoder_z.fc_layers.2.bias', 'encoder_z.fc11.weight', 'encoder_z.fc11.bias', 'encoder_z.fc12.weight',
        'encoder_z.fc12.bias', 'decoder.coord_latent.fc_coord.weight', 'decoder.coord_latent.fc_coord.bias',
        'decoder.coord_latent.fc_latent.weight', 'decoder.fc_layers.0.weight', 'decoder.fc_layers.0.bias',
        'decoder.fc_layers.2.weight', 'decoder.fc_layers.2.bias', 'decoder.out.weight', 'decoder.out.bias']
    assert m.z_dim == 2 + m.coord and m.c_dim == 0 and m.ndim == 2
    assert m.grid.shape == (784, 2)


def test_vanilla_uses_fc_decoder_and_samplers():
    m = pv.models.iVAE((8, 8), 2, None, device="cpu")
    assert isinstance(m.decoder, pv.nets.fcDecoderNet) and m.decoder.out.weight.shape == (64, 128)
    with pytest.raises(KeyError):
        pv.utils.get_sampler("poisson")
    assert pv.utils.get_sampler("gaussian", decoder_sig=0.3).decoder_sig == pytest.approx(0.3)


def test_save_load_weights_roundtrip(tmp_path):
    m = pv.models.iVAE((8, 8), 2, ['r'], seed=1, device="cpu")
    m.save_weights(str(tmp_path / "w"))
    m2 = pv.models.iVAE((8, 8), 2, ['r'], seed=7, device="cpu")
    assert not torch.equal(m.decoder.out.weight, m2.decoder.out.weight)
    m2.load_weights(str(tmp_path / "w.pt"))
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_no_cpu_fallback():
    """The product refuses to run without a HIP device instead of silently computing on the CPU."""
    m = pv.models.iVAE((8, 8), 2, ['r'], seed=1, device="cpu")
    with pytest.raises(_abi.PvError):
        pv.trainers.SVItrainer(m)
    with pytest.raises(_abi.PvError):
        m.encode(torch.rand(3, 8, 8))
    with pytest.raises(_abi.PvError):
        pv.utils.transform_coordinates(torch.zeros(2, 4, 2))
    with pytest.raises(_abi.PvError):
        m.encoder_z(torch.rand(3, 8, 8))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pyroved_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(dirpath, f)


# ------------------------------------------------------------------------------- trainer host logic
def _trainer_run(name, engine_factory, **kw):
    gold = load_golden(name)
    inv = str(gold["meta.invariances"])
    data_dim = tuple(int(v) for v in gold["meta.data_dim"])
    train, test = torch.from_numpy(gold["train"]), torch.from_numpy(gold["test"])
    batch = int(gold["meta.batch"])
    train_loader = pv.utils.init_dataloader(train, batch_size=batch)
    test_loader = pv.utils.init_dataloader(test, batch_size=batch)
    model = pv.models.iVAE(data_dim, 2, list(inv) if inv else None, seed=1, device="cpu")
    cfg = orc.Config(data_dim=data_dim, latent_dim=2, invariances=list(inv) if inv else None)
    eng = engine_factory(model, cfg)
    trainer = pv.trainers.SVItrainer(model, seed=1, engine=eng, device="cpu", **kw)
    for _ in range(int(gold["meta.epochs"])):
        if int(gold["meta.with_test"]):
            trainer.step(train_loader, test_loader)
        else:
            trainer.step(train_loader)
    return gold, trainer, eng


@pytest.mark.parametrize("name", ["epochs_8x8_rts", "epochs_8x8_r_notest", "epochs_8x8_none"])
def test_trainer_host_logic_reproduces_reference_epochs(name, capsys):
    """SVItrainer's loop (data order, eps draws, evaluate-steps-the-optimizer, normalisation by
    len(dataset), history, print format) around a stand-in engine == the reference's trainer."""
    gold, trainer, eng = _trainer_run(name, OracleEngine)
    np.testing.assert_allclose(trainer.loss_history["training_loss"], gold["epochs.training_loss"], rtol=1e-5)
    np.testing.assert_allclose(trainer.loss_history["test_loss"], gold["epochs.test_loss"], rtol=1e-5)
    for key in eng.o.p:
        check_digest(eng.o.p[key], gold, "final." + key, rtol=1e-4, atol=1e-7, what=name)
    trainer.print_statistics()
    out = capsys.readouterr().out.strip()
    if len(gold["epochs.test_loss"]):
        assert out == 'Epoch: {} Training loss: {:.4f}, Test loss: {:.4f}'.format(
            trainer.current_epoch, trainer.loss_history["training_loss"][-1], trainer.loss_history["test_loss"][-1])
    else:
        assert out == 'Epoch: {} Training loss: {:.4f}'.format(
            trainer.current_epoch, trainer.loss_history["training_loss"][-1])


@pytest.mark.parametrize("name", ["epochs_8x8_rts", "epochs_8x8_r_notest"])
def test_device_feed_equals_loader_iteration(name):
    """The device-resident data feed (default) and plain iteration of the caller's DataLoader consume the RNG streams
    identically: same minibatch order, same eps, same loss history (both equal the reference's, test above)."""
    _, fed, _ = _trainer_run(name, OracleEngine, device_feed=True)
    _, plain, _ = _trainer_run(name, OracleEngine, device_feed=False)
    assert fed._feed_cache is not None and plain._feed_cache is None
    assert fed.loss_history == plain.loss_history


def test_device_feed_falls_back_on_custom_loaders():
    """Anything but a plain TensorDataset loader is iterated as given."""
    class Loader:
        def __init__(self, x):
            self.dataset = x
            self.x = x
        def __iter__(self):
            for i in range(0, len(self.x), 2):
                yield (self.x[i:i + 2],)
    model = pv.models.iVAE((8, 8), 2, ["r"], seed=1, device="cpu")
    cfg = orc.Config(data_dim=(8, 8), latent_dim=2, invariances=["r"])
    tr = pv.trainers.SVItrainer(model, seed=1, engine=OracleEngine(model, cfg), device="cpu")
    tr.step(Loader(torch.rand(5, 8, 8)))
    assert tr._feed_cache is None and len(tr.loss_history["training_loss"]) == 1


def test_inference_batches_equal_a_nonshuffling_loader():
    """encode / decode / predict walk their input with utils.iter_batches instead of a DataLoader (which indexes and collates
    sample by sample): same batch boundaries, order and contents as init_dataloader(..., shuffle=False) (base.py:129,154)."""
    from pyroved_amd.utils import init_dataloader, iter_batches
    x, y = torch.rand(1037, 5, 3), torch.rand(1037, 2)
    for bs in (100, 1, 64, 1037, 2000):
        a = list(init_dataloader(x, y, shuffle=False, batch_size=bs))
        b = list(iter_batches(x, y, batch_size=bs))
        assert len(a) == len(b)
        for p_, q_ in zip(a, b):
            assert torch.equal(p_[0], q_[0]) and torch.equal(p_[1], q_[1])
    assert len(list(iter_batches(x))) == 11                         # the reference's default batch size of 100
    # ... and the global CPU generator ends in the same state (iter(DataLoader) draws a base seed even without shuffling)
    torch.manual_seed(5); list(init_dataloader(x, shuffle=False, batch_size=64)); a = torch.get_rng_state()
    torch.manual_seed(5); list(iter_batches(x, batch_size=64)); b = torch.get_rng_state()
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        list(iter_batches(x, y[:5]))


def test_shard_bounds_cover_batch():
    for n in (0, 1, 5, 7, 256, 257):
        for w in (1, 2, 3, 8):
            spans = [pvdist.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, name, q):
    import torch.distributed as td
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        gold, trainer, eng = _trainer_run(name, OracleEngine)
        q.put((rank, trainer.loss_history, {k: v.detach().numpy().copy() for k, v in eng.o.p.items()}))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("name", ["epochs_8x8_rts", "epochs_8x8_r_notest"])
def test_data_parallel_world2_gloo_matches_single_process(name):
    """Two ranks over gloo, each computing its contiguous shard of every global minibatch (incl. the
    odd-sized and last partial batches), ONE all-reduce of [grads | scalars] per step: the loss history
    and the final weights equal the single-process reference run."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    gold = load_golden(name)
    res.sort(key=lambda t: t[0])
    for rank, hist, params in res:
        np.testing.assert_allclose(hist["training_loss"], gold["epochs.training_loss"], rtol=2e-5)
        np.testing.assert_allclose(hist["test_loss"], gold["epochs.test_loss"], rtol=2e-5)
        for key, p in params.items():
            check_digest(torch.from_numpy(p), gold, "final." + key, rtol=2e-4, atol=1e-7,
                         what="%s rank %d" % (name, rank))
    for key in res[0][2]:     # replicas stay in lock-step
        assert np.array_equal(res[0][2][key], res[1][2][key]), key


# ------------------------------------------------------------------------------- auxSVItrainer host logic
def _aux_trainer_run(name):
    from conftest import ssmeta_of, ss_build
    from _oracle_ss_engine import OracleSSEngine
    gold = load_golden(name)
    meta = ssmeta_of(gold)
    model = ss_build(meta, "cpu")
    model.engine = lambda **kw: eng                      # model.classifier / regressor go through the stand-in too
    cfg = orc.Config(data_dim=meta["data_dim"], latent_dim=meta["latent_dim"], invariances=meta["invariances"],
                     c_dim=meta["dim"])
    eng = OracleSSEngine(model, cfg, meta["task"])
    trainer = pv.trainers.auxSVItrainer(model, task=meta["task"], seed=1, engine=eng, device="cpu")
    xu, xs, ys = (torch.from_numpy(gold[k]) for k in ("xu", "xs", "ys"))
    lu, ls, lv = pv.utils.init_ssvae_dataloaders(xu[:meta["n_u"]], (xs[:meta["n_s"]], ys[:meta["n_s"]]),
                                                 (xs[:meta["n_s"]], ys[:meta["n_s"]]), batch_size=meta["batch_s"])
    for _ in range(meta["epochs"]):
        trainer.step(lu, ls, lv, scale_factor=meta["beta"], aux_loss_multiplier=meta["mult"])
    return gold, trainer, eng


@pytest.mark.parametrize("name", ["sscls_8x8_rt_k3", "ssreg_8x8_rt_c2", "sscls_8x8_rts_k4_sf"])
def test_aux_trainer_host_logic_reproduces_reference_epochs(name):
    """auxSVItrainer's loop around a stand-in engine == the reference trainer: loader interleaving, the guide's noise
    stream (one (K, B, z) draw for unlabeled classification batches; label noise first for regression), two optimizer
    steps per call, the test metric."""
    gold, trainer, _ = _aux_trainer_run(name)
    np.testing.assert_allclose(trainer.history["training_loss"], gold["epochs.training_loss"], rtol=2e-5)
    np.testing.assert_allclose([float(v) for v in trainer.history["test"]], gold["epochs.test"], rtol=1e-4, atol=1e-7)


def _aux_dp_worker(rank, world, port, name, q):
    import torch.distributed as td
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        gold, trainer, eng = _aux_trainer_run(name)
        q.put((rank, trainer.history["training_loss"], [float(v) for v in trainer.history["test"]],
               {k: v.detach().numpy().copy() for k, v in eng.o.p.items()}))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("name", ["sscls_8x8_rt_k3", "ssreg_8x8_rts_c2_sf"])
def test_aux_trainer_data_parallel_world2_gloo(name):
    """Two ranks over gloo: every compute_loss call shards its (global) batch by samples — the K enumerated passes of
    an unlabeled classification batch stay with their sample — and all-reduces [gradients | loss] once per SVI step; the
    histories equal the single-process reference run and the replicas stay in lock-step."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_aux_dp_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    gold = load_golden(name)
    res.sort(key=lambda t: t[0])
    for rank, tr, te, params in res:
        np.testing.assert_allclose(tr, gold["epochs.training_loss"], rtol=5e-5)
        np.testing.assert_allclose(te, gold["epochs.test"], rtol=1e-3, atol=1e-6)
    for key in res[0][3]:
        assert np.array_equal(res[0][3][key], res[1][3][key]), key


def test_roctx_ranges_resolve_without_a_link_dependency():
    """The ABI entry points carry roctx ranges (csrc/pv_common.h PV_RANGE) whose two symbols are looked up at run time: the
    library must load where no marker library exists (nothing in its NEEDED list names one) and find the markers when asked
    to (PV_ROCTX=1) — checked in a fresh process, the lookup being once per process."""
    import subprocess
    import sys
    from pyroved_amd import _abi
    needed = subprocess.run(["readelf", "-d", _abi.LIB_PATH], capture_output=True, text=True).stdout
    assert "roctx" not in needed.lower()
    code = ("import ctypes as C, os; os.environ['PV_ROCTX'] = '%s'; "
            "print(C.CDLL(%r).pv_debug_roctx_state())" % ("%s", _abi.LIB_PATH))
    have = any(os.path.exists(os.path.join("/opt/rocm/lib", n)) for n in ("librocprofiler-sdk-roctx.so", "libroctx64.so"))
    off = subprocess.run([sys.executable, "-c", code % "0"], capture_output=True, text=True)
    assert off.returncode == 0 and off.stdout.strip() == "0", off.stderr
    on = subprocess.run([sys.executable, "-c", code % "1"], capture_output=True, text=True)
    assert on.returncode == 0 and on.stdout.strip() == ("1" if have else "0"), on.stderr


def test_dist_layer_resolves_the_process_rccl_and_refuses_a_second_one():
    """ABI v16: the collective lives in the library but RCCL is not a link dependency — pv_dist_load resolves ncclAllReduce & co. in
    the RCCL shared object the process already holds (PyTorch's own), once; without a communicator every entry point is PV_EINVAL
    (no compute call here: there is no GPU)."""
    import ctypes as C
    import subprocess
    from pyroved_amd import dist as pvdist
    path = pvdist._loaded_rccl_path()
    assert path and "rccl" in os.path.basename(path)
    lib = _abi.lib()
    assert lib.pv_dist_load(path.encode()) == 0
    assert lib.pv_dist_library().decode() == path
    assert lib.pv_dist_load(path.encode()) == 0                       # idempotent
    assert lib.pv_dist_load(b"/nonexistent/librccl.so") == -1         # a second library in one process: PV_EINVAL
    assert lib.pv_dist_allreduce_sum(None, None, 4, None) == -1
    assert lib.pv_ivae_dp_step(None, None, None, None) == -1
    assert lib.pv_dist_comm_info(None, None, None) == -1
    # libpyroved_amd.so itself must not link RCCL (a single-GPU user never loads it)
    out = subprocess.run(["readelf", "-d", _abi.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in out.lower()
    # a path that does not load is an error of its own kind (fresh process: nothing resolved yet)
    code = ("import sys; sys.path.insert(0, %r); from pyroved_amd import _abi; "
            "print(_abi.lib().pv_dist_load(b'/nonexistent/librccl.so'))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.stdout.strip() == "-3", (r.stdout, r.stderr[-500:])      # PV_ECOLL
