"""CPU checks of the convolutional plugin nets (muax/nn.py:47-56,118-395 mirrored as torch modules): haiku
'SAME' geometry, NHWC contract, valid-count average pooling, per-channel min-max normalisation."""
import numpy as np
import torch

import muax_amd as mx


def _conv_same_numpy(x, w, stride):
    """Naive NHWC / HWIO convolution with TensorFlow 'SAME' padding."""
    B, H, W, Ci = x.shape
    k, _, _, Co = w.shape
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph, pw = max((Ho - 1) * stride + k - H, 0), max((Wo - 1) * stride + k - W, 0)
    xp = np.zeros((B, H + ph, W + pw, Ci), np.float64)
    xp[:, ph // 2:ph // 2 + H, pw // 2:pw // 2 + W] = x
    y = np.zeros((B, Ho, Wo, Co))
    for i in range(Ho):
        for j in range(Wo):
            patch = xp[:, i * stride:i * stride + k, j * stride:j * stride + k]
            y[:, i, j] = np.tensordot(patch, w, axes=([1, 2, 3], [0, 1, 2]))
    return y


def test_hk_conv2d_same_padding_matches_naive_convolution():
    rng = np.random.default_rng(0)
    for H, W, k, stride in ((8, 8, 3, 2), (7, 9, 3, 2), (6, 6, 3, 1), (5, 5, 1, 1), (84, 84, 3, 2)):
        conv = mx.nn.HkConv2D(5, k, stride, generator=torch.Generator().manual_seed(1))
        x = rng.normal(size=(2, H, W, 3)).astype(np.float32)
        with torch.no_grad():
            y = conv(torch.from_numpy(x)).numpy()
        ref = _conv_same_numpy(x, conv.w.detach().numpy().astype(np.float64), stride)
        assert y.shape == ref.shape == (2, -(-H // stride), -(-W // stride), 5)
        assert np.allclose(y, ref, atol=1e-5)
        assert abs(float(conv.w.std()) - 0.88 / np.sqrt(k * k * 3)) < 0.5 / np.sqrt(k * k * 3)


def test_avg_pool_same_divides_by_valid_count_and_normalize2d():
    x = torch.arange(2 * 5 * 5 * 1, dtype=torch.float32).reshape(2, 5, 5, 1)
    y = mx.nn.avg_pool_same(x)  # 5 -> 3, windows clipped at the border
    assert y.shape == (2, 3, 3, 1)
    assert float(y[0, 0, 0, 0]) == float(x[0, 0:2, 0:2, 0].mean())  # corner window holds 4 valid pixels
    assert float(y[0, 1, 1, 0]) == float(x[0, 1:4, 1:4, 0].mean())
    assert torch.allclose(mx.nn.avg_pool_same(torch.ones(1, 21, 21, 3)), torch.ones(1, 11, 11, 3))
    s = torch.rand(3, 4, 4, 6) * 5 - 1
    n = mx.nn.min_max_normalize2d(s)
    assert torch.allclose(n.amin((1, 2)), torch.zeros(3, 6)) and torch.allclose(n.amax((1, 2)), torch.ones(3, 6))
    flat = torch.ones(1, 2, 2, 1)
    assert torch.equal(mx.nn.min_max_normalize2d(flat), torch.zeros(1, 2, 2, 1))  # scale < 1e-5 -> +1e-5


def test_resnet_trio_contract_atari_shape():
    g = torch.Generator().manual_seed(0)
    net = mx.nn.create_muzero_network(lambda e: mx.nn.ResNetRepresentation(8, generator=g),
                                      lambda a, f: mx.nn.ResNetPrediction(a, f, generator=g),
                                      lambda e, a, f: mx.nn.ResNetDynamic(a, f, output_channels=16, generator=g),
                                      embedding_dim=8, num_actions=18, full_support_size=21)
    obs = torch.randint(0, 256, (2, 84, 84, 4)).float()
    with torch.no_grad():
        s = net.representation_fn(obs)
        v, lg = net.prediction_fn(s)
        r, ns = net.dynamic_fn(s, torch.tensor([0, 17]))
    assert s.shape == ns.shape == (2, 6, 6, 16) and v.shape == r.shape == (2, 21) and lg.shape == (2, 18)
    assert float(ns.min()) == 0.0 and float(ns.max()) == 1.0
    assert not mx.nn.is_default_mlp_trio(net)
    b1 = mx.nn.ResidualConvBlockV1(4, 1, False)
    b2 = mx.nn.ResidualConvBlockV2(6, 2, True)
    with torch.no_grad():
        x = torch.randn(2, 8, 8, 4)
        assert b1(x).shape == (2, 8, 8, 4) and float(b1(x).min()) >= 0.0
        assert b2(x).shape == (2, 4, 4, 6)
    ln = mx.nn.HkLayerNorm()
    with torch.no_grad():
        y = ln(torch.randn(3, 4, 4, 5) * 3 + 2)
    assert torch.allclose(y.mean((1, 2, 3)), torch.zeros(3), atol=1e-5) and torch.allclose(
        y.var((1, 2, 3), unbiased=False), torch.ones(3), atol=1e-3)


def test_ez_trio_contract_and_factories():
    """muax/nn.py:180-309 (EfficientZero-style nets) and the factory functions muax/nn.py:398-451: geometry of the
    encoder (84 -> 42 -> 21 -> 11 -> 6), both residual-block flavours, the raw-action plane of EZDynamic, the
    LayerNorm heads and the VarianceScaling init of their last layer."""
    g = torch.Generator().manual_seed(0)
    nn_ = mx.nn
    rep = nn_._init_ez_representation_func(lambda e: nn_.EZRepresentation(e, generator=g), 16)
    pred = nn_._init_ez_prediction_func(lambda a, f, sc: nn_.EZPrediction(a, f, sc, generator=g), 18, 21, 0.1)
    dyn = nn_._init_ez_dynamic_func(lambda e, a, f, sc: nn_.EZDynamic(e, a, f, sc, generator=g), 16, 18, 21, 0.1)
    obs = torch.randint(0, 256, (2, 84, 84, 4)).float()
    with torch.no_grad():
        s = rep(obs)
        v, lg = pred(s)
        r, ns = dyn(s, torch.tensor([0, 17]))
        r2, ns2 = dyn(s, torch.tensor([5, 17]))
    assert s.shape == ns.shape == (2, 6, 6, 16) and v.shape == r.shape == (2, 21) and lg.shape == (2, 18)
    assert not torch.equal(ns[0], ns2[0]) and torch.equal(ns[1], ns2[1])  # the action plane matters, per sample
    w = pred.v_func.out.w
    assert abs(float(w.std()) - np.sqrt(0.1 / 32)) < 0.4 * np.sqrt(0.1 / 32) and float(pred.v_func.out.b.abs().max()) == 0.0
    assert pred.v_func.fc.b is None and tuple(pred.v_func.fc.w.shape) == (6 * 6 * 16, 32)
    v1 = nn_.EZStateEncoder(8, use_v2=False, generator=g)
    with torch.no_grad():
        assert v1(obs).shape == (2, 6, 6, 8) and float(v1(obs).min()) >= 0.0  # v1 blocks end in a relu
        d1 = nn_.EZDynamic(8, 4, 21, 1.0, use_v2=False, generator=g)
        r1, n1 = d1(v1(obs), torch.tensor([1, 2]))
    assert n1.shape == (2, 6, 6, 8) and r1.shape == (2, 21)
    # the ResNet factories (muax/nn.py:435-451) build the modules config 4 uses
    net = mx.MZNetwork(nn_._init_resnet_representation_func(lambda input_channels: nn_.ResNetRepresentation(input_channels, generator=g), 8),
                       nn_._init_resnet_prediction_func(lambda a, f, c: nn_.ResNetPrediction(a, f, c, generator=g), 18, 21, 16),
                       nn_._init_resnet_dynamic_func(lambda a, f, c: nn_.ResNetDynamic(a, f, output_channels=c, generator=g), 18, 21, 16))
    with torch.no_grad():
        s = net.representation_fn(obs)
        assert s.shape == (2, 6, 6, 16) and net.prediction_fn(s)[1].shape == (2, 18)
        assert net.dynamic_fn(s, torch.tensor([3, 4]))[1].shape == (2, 6, 6, 16)
    # a MuZero on the EZ nets initialises and exposes its sub-networks (the search itself needs the GPU)
    m = mx.MuZero(rep, pred, dyn, device="cpu")
    m.init(0, np.zeros((1, 84, 84, 4), np.float32))
    assert m.representation(obs.numpy()).shape == (2, 6, 6, 16)
