"""
The scenarios of the reference's own test-suite that do not need Pyro's trace machinery (tests/test_models.py:427-666,
tests/test_trainers.py, tests/test_conv.py, tests/test_utils.py of pyroVED), restated against this package on the GPU:
the same constructor calls, argument forms and sizes, the same shape / no-NaN / weights-changed assertions.  (The
`*_sites_*` tests of the reference inspect Pyro traces of model()/guide(); Pyro is not a dependency here.)
"""
from copy import deepcopy as dc

import numpy as np
import pytest
import torch

import pyroved_amd as pv
from pyroved_amd import models, nets, trainers, utils

pytestmark = pytest.mark.gpu
tt = torch.tensor
INV5 = [None, ['r'], ['s'], ['t'], ['r', 't', 's']]
VED_DIMS = [((8,), (8, 8)), ((8, 8), (8,)), ((8,), (8,)), ((8, 8), (8, 8))]


@pytest.fixture(autouse=True)
def _headless(gpu_device):
    import matplotlib
    matplotlib.use("Agg")
    yield
    import matplotlib.pyplot as plt
    plt.close("all")


def weights_equal(m1, m2):
    return all(np.array_equal(p1.detach().cpu().numpy(), p2.detach().cpu().numpy()) for p1, p2 in zip(m1.values(), m2.values()))


def n_coord(invariances, ndim=2):
    if invariances is None:
        return 0
    return len(invariances) + (1 if 't' in invariances and ndim == 2 else 0)


# ------------------------------------------------------------------ models: encode / decode / manifold / weights
@pytest.mark.parametrize("data_dim", [(2, 8), (2, 8, 8), (3, 8), (3, 8, 8)])
def test_basevae_encode_x(data_dim):
    x = torch.randn(*data_dim)
    vae = models.base.baseVAE(data_dim[1:], None)
    vae.set_encoder(nets.fcEncoderNet(data_dim[1:], 2, 0))
    encoded = vae._encode(x)
    assert encoded[:, :2].shape == (data_dim[0], 2) and encoded[:, 2:].shape == (data_dim[0], 2)


def test_basevae_encode_xy():
    x = torch.randn(2, 64)
    y = utils.to_onehot(torch.tensor([0, 2]), 3)
    vae = models.base.baseVAE((64,), None)
    vae.set_encoder(nets.fcEncoderNet((64,), 2, 3))
    encoded = vae._encode(x, y)
    assert encoded[:, :2].shape == (2, 2) and encoded[:, 2:].shape == (2, 2)


@pytest.mark.parametrize("invariances", [None, ['r'], ['s'], ['r', 't', 's']])
def test_basevae_decode_x(invariances):
    data_dim = (3, 8, 8)
    coord = n_coord(invariances)
    z = torch.randn(data_dim[0], 2)
    vae = models.base.baseVAE(data_dim[1:], invariances)
    vae.coord = coord
    vae.grid = utils.generate_grid(data_dim[1:]).to(vae.device)
    dnet = nets.sDecoderNet if 0 < coord < 5 else nets.fcDecoderNet
    vae.set_decoder(dnet(data_dim[1:], 2))
    decoded = vae._decode(z)
    assert decoded.squeeze().shape == data_dim


@pytest.mark.parametrize("vae_model", [models.jiVAE, models.ssiVAE])
@pytest.mark.parametrize("invariances", [None, ['r'], ['s'], ['r', 't', 's']])
def test_joint_and_ss_decode(vae_model, invariances):
    data_dim = (38, 8)
    model = vae_model(data_dim, 2, 3, invariances=invariances)
    decoded = model.decode(torch.tensor([0.0, 0.0]).unsqueeze(0), utils.to_onehot(torch.tensor(0).unsqueeze(0), 3))
    assert decoded.squeeze().shape == data_dim


@pytest.mark.parametrize("data_dim, invariances", [((8, 8), None), ((8, 8), ['r']), ((8, 8), ['s']), ((8, 8), ['r', 't', 's']),
                                                   ((8,), None), ((8,), ['t'])])
def test_ivae_decode(data_dim, invariances):
    model = models.iVAE(data_dim, invariances=invariances)
    assert model.decode(torch.tensor([0.0, 0.0]).unsqueeze(0)).squeeze().shape == data_dim


@pytest.mark.parametrize("input_dim, output_dim", VED_DIMS)
def test_ved_decode_predict_encode_manifold(input_dim, output_dim):
    model = models.VED(input_dim, output_dim)
    assert model.decode(torch.tensor([0.0, 0.0]).unsqueeze(0)).squeeze().shape == output_dim
    x = torch.randn(2, 1, *input_dim)
    prediction, _ = model.predict(x)
    assert prediction.squeeze().shape == (2, *output_dim)
    encoded = model.encode(x)
    assert encoded[0].shape == (2, 2) and encoded[0].shape == encoded[1].shape
    assert model.manifold2d(4, plot=True).squeeze().shape == (16, *output_dim)


def test_ved_predict_matches_its_definition():
    """VED.predict (models/ved.py:198-216): mean and standard deviation over 30 decoded draws z ~ N(z_mu, z_sig) per input.
    The 30 draws of a batch are decoded in ONE call; the same draws (same seed: predict's only generator use is the loop's
    base seed and one rsample per batch) decoded one by one give the same numbers."""
    model = models.VED((16, 16), (24,), seed=3)
    x = torch.randn(5, 1, 16, 16, generator=torch.Generator().manual_seed(1))
    torch.manual_seed(11)
    mu, sd = model.predict(x, batch_size=3)
    eng = model.engine()
    torch.manual_seed(11)
    torch.empty((), dtype=torch.int64).random_()
    ref_mu, ref_sd = [], []
    for lo in (0, 3):
        z_mu, z_sig = eng.encode(x[lo:lo + 3].to(eng.device, torch.float32))
        zs = torch.distributions.Normal(z_mu.cpu(), z_sig.cpu()).rsample(sample_shape=(30,))
        y = torch.stack([eng.decode(z.to(eng.device)).cpu() for z in zs])
        ref_mu.append(y.mean(0)); ref_sd.append(y.std(0))
    assert mu.shape[0] == 5 and sd.shape == mu.shape and (sd >= 0).all()
    np.testing.assert_allclose(mu.numpy(), torch.cat(ref_mu).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sd.numpy(), torch.cat(ref_sd).numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("invariances", [None, ['r'], ['s'], ['r', 't', 's']])
def test_conditional_ivae_decode(invariances):
    model = models.iVAE((8, 8), c_dim=3, invariances=invariances)
    y = utils.to_onehot(torch.tensor(0).unsqueeze(0), 3)
    assert model.decode(torch.tensor([0.0, 0.0]).unsqueeze(0), y).squeeze().shape == (8, 8)


@pytest.mark.parametrize("invariances", INV5)
def test_encode_shapes(invariances):
    x = torch.randn(3, 8, 8)
    coord = n_coord(invariances)
    enc = models.iVAE((8, 8), 2, invariances=invariances).encode(x)
    assert enc[0].shape == (3, coord + 2) and enc[0].shape == enc[1].shape
    enc = models.jiVAE((8, 8), 2, 3, invariances=invariances).encode(x)
    assert enc[0].shape == enc[1].shape == (3, coord + 2) and enc[2].shape == (3,)
    enc = models.ssiVAE((8, 8), 2, 5, invariances=invariances).encode(x.reshape(3, 64))
    assert enc[0].shape == enc[1].shape == (3, coord + 2) and enc[2].shape == (3,)
    if invariances in (None, ['t']):
        c1 = 0 if invariances is None else 1
        enc = models.iVAE((8,), 2, invariances=invariances).encode(torch.randn(3, 8))
        assert enc[0].shape == (3, c1 + 2) and enc[0].shape == enc[1].shape


@pytest.mark.parametrize("num_classes", [0, 2, 3])
@pytest.mark.parametrize("invariances", INV5)
def test_ivae_manifold2d(invariances, num_classes):
    model = models.iVAE((8, 8), c_dim=num_classes, invariances=invariances)
    y = utils.to_onehot(torch.tensor(0).unsqueeze(0), num_classes) if num_classes > 0 else None
    assert model.manifold2d(4, y, plot=True).squeeze().shape == (16, 8, 8)


@pytest.mark.parametrize("vae_model", [models.jiVAE, models.ssiVAE])
@pytest.mark.parametrize("invariances", INV5)
def test_joint_and_ss_manifold2d(vae_model, invariances):
    model = vae_model((8, 8), 2, 3, invariances=invariances)
    assert model.manifold2d(4, plot=True).squeeze().shape == (16, 8, 8)


@pytest.mark.parametrize("invariances", INV5)
def test_save_load_basevae(invariances, tmp_path):
    coord = n_coord(invariances)
    vae = models.base.baseVAE((8, 8), invariances)
    vae.set_encoder(nets.fcEncoderNet((8, 8), 2 + coord, 0))
    dnet = nets.sDecoderNet if 0 < coord < 5 else nets.fcDecoderNet
    vae.set_decoder(dnet((8, 8), 2, 0))
    weights_init = dc(vae.state_dict())
    vae.save_weights(str(tmp_path / "my_weights"))
    vae.load_weights(str(tmp_path / "my_weights.pt"))
    assert weights_equal(vae.state_dict(), weights_init)


# ------------------------------------------------------------------ trainers
@pytest.mark.parametrize("invariances", INV5)
def test_svi_trainer_ivae(invariances):
    train_loader = utils.init_dataloader(torch.randn(5, 8, 8), batch_size=2)
    test_loader = utils.init_dataloader(torch.randn(5, 8, 8), batch_size=2)
    vae = models.iVAE((8, 8), 2, invariances)
    trainer = trainers.SVItrainer(vae)
    before = dc(vae.state_dict())
    for _ in range(2):
        trainer.step(train_loader, test_loader)
    assert not torch.isnan(tt(trainer.loss_history["training_loss"])).any()
    assert not weights_equal(before, vae.state_dict())
    trainer.print_statistics()


@pytest.mark.parametrize("invariances", INV5)
def test_svi_trainer_jivae(invariances):
    train_loader = utils.init_dataloader(torch.randn(6, 8, 8), batch_size=2)
    vae = models.jiVAE((8, 8), 2, 3, invariances)
    trainer = trainers.SVItrainer(vae, enumerate_parallel=True)
    before = dc(vae.state_dict())
    for _ in range(2):
        trainer.step(train_loader)
    assert not torch.isnan(tt(trainer.loss_history["training_loss"])).any()
    assert not weights_equal(before, vae.state_dict())


@pytest.mark.parametrize("task, c_dim", [("classification", 3), ("regression", 1), ("regression", 2)])
@pytest.mark.parametrize("invariances", INV5)
def test_auxsvi_trainer(task, c_dim, invariances):
    train_unsup = torch.randn(5, 64)
    train_sup = train_unsup + .1 * torch.randn_like(train_unsup)
    if task == "classification":
        labels = utils.to_onehot(torch.randint(0, 3, (5,)), 3)
        vae = models.ssiVAE((8, 8), 2, 3, invariances)
        trainer = trainers.auxSVItrainer(vae)
    else:
        labels = torch.randn(5, c_dim)
        vae = models.ss_reg_iVAE((8, 8), 2, c_dim, invariances)
        trainer = trainers.auxSVItrainer(vae, task="regression")
    lu, ls, lv = utils.init_ssvae_dataloaders(train_unsup, (train_sup, labels), (train_sup, labels), batch_size=2)
    before = dc(vae.state_dict())
    for _ in range(2):
        trainer.step(lu, ls, lv)
    assert not torch.isnan(tt(trainer.history["training_loss"])).any()
    assert not weights_equal(before, vae.state_dict())
    trainer.print_statistics()


@pytest.mark.parametrize("invariances", INV5)
def test_auxsvi_trainer_swa(invariances):
    train_unsup = torch.randn(5, 64)
    train_sup = train_unsup + .1 * torch.randn_like(train_unsup)
    labels = utils.to_onehot(torch.randint(0, 3, (5,)), 3)
    lu, ls, _ = utils.init_ssvae_dataloaders(train_unsup, (train_sup, labels), (train_sup, labels), batch_size=2)
    vae = models.ssiVAE((8, 8), 2, 3, invariances)
    trainer = trainers.auxSVItrainer(vae)
    for _ in range(3):
        trainer.step(lu, ls)
        trainer.save_running_weights("encoder_y")
    final = dc(vae.encoder_y.state_dict())
    trainer.average_weights("encoder_y")
    assert not weights_equal(final, vae.encoder_y.state_dict())


@pytest.mark.parametrize("input_dim, output_dim", VED_DIMS)
def test_svi_trainer_ved(input_dim, output_dim):
    loader = utils.init_dataloader(torch.randn(5, 1, *input_dim), torch.randn(5, 1, *output_dim), batch_size=2)
    vae = models.VED(input_dim, output_dim)
    trainer = trainers.SVItrainer(vae)
    before = dc(vae.state_dict())
    for _ in range(2):
        trainer.step(loader)
    assert not torch.isnan(tt(trainer.loss_history["training_loss"])).any()
    assert not weights_equal(before, vae.state_dict())


# ------------------------------------------------------------------ nets.conv (tests/test_conv.py of the reference)
@pytest.mark.parametrize("hidden_dim, bnorm, nbnorm", [([(8,)], True, 1), ([(8,)], False, 0),
                                                       ([(8,), (16, 16)], True, 3), ([(8,), (16, 16)], False, 0)])
def test_feature_extractor_bnorm(hidden_dim, bnorm, nbnorm):
    c = nets.FeatureExtractor(2, conv_filters=hidden_dim, batchnorm=bnorm)
    assert len([k for k in c.state_dict().keys() if 'running_mean' in k]) == nbnorm


@pytest.mark.parametrize("activation, expected", [("relu", torch.nn.ReLU), ("lrelu", torch.nn.LeakyReLU),
                                                  ("softplus", torch.nn.Softplus), ("tanh", torch.nn.Tanh)])
def test_feature_extractor_activation(activation, expected):
    conv_ = nets.FeatureExtractor(2, conv_filters=[(8, 8)], activation=activation)
    assert sum(isinstance(c2, expected) for c1 in conv_.children() for c2 in c1.children()) == 2


@pytest.mark.parametrize("dim, expected", [(1, torch.nn.Conv1d), (2, torch.nn.Conv2d), (3, torch.nn.Conv3d)])
def test_feature_extractor_dim(dim, expected):
    conv_ = nets.FeatureExtractor(dim, conv_filters=[(8, 8)])
    assert sum(isinstance(c2, expected) for c1 in conv_.children() for c2 in c1.children()) == 2


@pytest.mark.parametrize("pool_last", [True, False])
@pytest.mark.parametrize("dim, size", [(1, [8]), (2, [8, 8]), (3, [8, 8, 8])])
def test_feature_extractor_forward(dim, size, pool_last):
    data = torch.randn(2, 1, *size).cuda()
    out = nets.FeatureExtractor(dim, conv_filters=[(8, 8)], pool_last=pool_last).cuda()(data)
    assert sum(out.size(i + 2) for i in range(dim)) == sum(size) / (2 if pool_last else 1)


@pytest.mark.parametrize("dim, size", [(1, [8]), (2, [8, 8]), (3, [8, 8, 8])])
def test_upsample_block(dim, size):
    data = torch.randn(2, 2, *size).cuda()
    out = nets.UpsampleBlock(dim, 2, 2, mode="nearest").cuda()(data)
    assert sum(out.size(i + 2) for i in range(dim)) == sum(size) * 2


@pytest.mark.parametrize("in_channels, out_channels", [(8, 8), (8, 4), (4, 8)])
def test_upsampleblock_change_number_of_channels(in_channels, out_channels):
    out = nets.UpsampleBlock(2, in_channels, out_channels).cuda()(torch.randn(4, in_channels, 8, 8).cuda())
    assert out.size(1) == out_channels


@pytest.mark.parametrize("latent_dim", [1, 2, 5])
@pytest.mark.parametrize("input_channels", [1, 2, 3])
@pytest.mark.parametrize("input_dim", [(8,), (8, 8), (8, 8, 8)])
def test_conv_encoder_output(input_dim, input_channels, latent_dim):
    x = torch.randn(5, input_channels, *input_dim).cuda()
    encoder = nets.convEncoderNet(input_dim, latent_dim, input_channels, hidden_dim=[(8,), (8, 8)]).cuda()
    z1, z2 = encoder(x)
    assert z1.shape == z2.shape == (5, latent_dim)


@pytest.mark.parametrize("latent_dim", [1, 2, 5])
@pytest.mark.parametrize("output_channels", [1, 2, 3])
@pytest.mark.parametrize("output_dim", [(8,), (8, 8), (8, 8, 8)])
def test_conv_decoder_output(latent_dim, output_dim, output_channels):
    z = torch.randn(5, latent_dim).cuda()
    decoder = nets.convDecoderNet(latent_dim, output_dim, output_channels, hidden_dim=[(8, 8), (8,)]).cuda()
    assert decoder(z).shape == (5, output_channels, *output_dim)


# ------------------------------------------------------------------ GP helper (tests/test_utils.py, test_models.py:650-666)
def test_gp_model_output_shape():
    encoded_X, y = torch.randn(5, 3), torch.randn(5)
    gpr = utils.gp_model(3, encoded_X, y)
    with torch.no_grad():
        predictions, _ = gpr(encoded_X)
    assert predictions.shape == y.shape


def test_ivae_predict_on_latent():
    train_data, gp_labels, d = torch.randn(10, 5, 5), torch.randint(0, 2, (10,)), 12
    vae = models.iVAE((5, 5), latent_dim=2, invariances=None, seed=0)
    (z, z_decoded), predictions = vae.predict_on_latent(train_data, gp_labels, 1, d, plot=False)
    assert isinstance(z, torch.Tensor) and isinstance(predictions, torch.Tensor)
    assert z_decoded.dim() == 3 and predictions.dim() == 1 and z_decoded.shape == (d * d, 5, 5)
    (_, _), p2 = vae.predict_on_latent(train_data, gp_labels, 1, 4, plot=True)
    assert p2.shape == (16,)
