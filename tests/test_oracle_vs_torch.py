"""Second opinion on the oracle's op semantics and vjps: torch-CPU autograd.

torch is used here ONLY as an independent differentiator for the numpy oracle (CPU tests);
it is not part of the product.  The mapping between lasagne's true convolution and
torch's cross-correlation is the filter flip of SURVEY.md Appendix A.1/A.2.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from oracle import ops, step  # noqa: E402

rng = np.random.RandomState(0)


def t(a, grad=True):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, requires_grad=grad)


def flipt(w):
    return torch.flip(w, dims=(2, 3))


@pytest.mark.parametrize("k,s,pad,H", [(5, 1, 2, 9), (3, 2, 1, 8), (3, 1, 1, 7), (2, 1, 0, 2), (3, 2, 1, 7)])
def test_conv2d(k, s, pad, H):
    x, W, b = rng.randn(2, 3, H, H + 1), rng.randn(4, 3, k, k), rng.randn(4)
    y = ops.conv2d_fwd(x, W, b, s, pad)
    xt, Wt, bt = t(x), t(W), t(b)
    yt = F.conv2d(xt, flipt(Wt), bt, stride=s, padding=pad)
    assert np.allclose(y, yt.detach().numpy(), atol=1e-12)
    dy = rng.randn(*y.shape)
    yt.backward(t(dy, False))
    dx, dW, db = ops.conv2d_vjp(x, W, dy, s, pad)
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-11)
    assert np.allclose(dW, Wt.grad.numpy(), atol=1e-11)
    assert np.allclose(db, bt.grad.numpy(), atol=1e-11)


@pytest.mark.parametrize("k,s,H", [(2, 1, 1), (2, 2, 5), (3, 2, 4)])
def test_deconv2d(k, s, H):
    x, W, b = rng.randn(2, 3, H, H), rng.randn(3, 5, k, k), rng.randn(5)
    y = ops.deconv2d_fwd(x, W, b, s)
    xt, Wt, bt = t(x), t(W), t(b)
    yt = F.conv_transpose2d(xt, flipt(Wt), bt, stride=s)
    assert y.shape == tuple(yt.shape)
    assert np.allclose(y, yt.detach().numpy(), atol=1e-12)
    dy = rng.randn(*y.shape)
    yt.backward(t(dy, False))
    dx, dW, db = ops.deconv2d_vjp(x, W, dy, s)
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-11)
    assert np.allclose(dW, Wt.grad.numpy(), atol=1e-11)
    assert np.allclose(db, bt.grad.numpy(), atol=1e-11)


@pytest.mark.parametrize("shape", [(4, 6), (4, 5, 3, 3), (4, 7, 1, 1)])
def test_batchnorm(shape):
    x, beta, gamma = rng.randn(*shape), rng.randn(shape[1]), rng.randn(shape[1])
    y, mu, inv = ops.bn_train_fwd(x, beta, gamma)
    xt, bt, gt = t(x), t(beta), t(gamma)
    yt = F.batch_norm(xt, None, None, gt, bt, training=True, eps=ops.BN_EPS)
    assert np.allclose(y, yt.detach().numpy(), atol=1e-10)
    dy = rng.randn(*shape)
    yt.backward(t(dy, False))
    dx, dbeta, dgamma = ops.bn_train_vjp(x, gamma, mu, inv, dy)
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-9)
    assert np.allclose(dbeta, bt.grad.numpy(), atol=1e-10)
    assert np.allclose(dgamma, gt.grad.numpy(), atol=1e-10)


@pytest.mark.parametrize("shape", [(4, 5, 3, 3), (2, 8, 16, 16), (3, 7, 2, 1)])
def test_instancenorm(shape):
    """oracle/ops.py in_fwd / in_vjp (no reference call site; north_star names the op) against torch's instance_norm"""
    x, beta, gamma = rng.randn(*shape), rng.randn(shape[1]), rng.randn(shape[1])
    y, mu, inv = ops.in_fwd(x, beta, gamma)
    xt, bt, gt = t(x), t(beta), t(gamma)
    yt = F.instance_norm(xt, None, None, gt, bt, use_input_stats=True, eps=ops.BN_EPS)
    assert np.allclose(y, yt.detach().numpy(), atol=1e-10)
    assert mu.shape == shape[:2] and inv.shape == shape[:2]
    dy = rng.randn(*shape)
    yt.backward(t(dy, False))
    dx, dbeta, dgamma = ops.in_vjp(x, gamma, mu, inv, dy)
    assert np.allclose(dx, xt.grad.numpy(), atol=1e-9)
    assert np.allclose(dbeta, bt.grad.numpy(), atol=1e-10)
    assert np.allclose(dgamma, gt.grad.numpy(), atol=1e-10)


def test_pools_and_upsamples():
    x = rng.randn(2, 3, 8, 8)
    xt = t(x)
    y = ops.maxpool_fwd(x, 2)
    yt = F.max_pool2d(xt, 2)
    assert np.allclose(y, yt.detach().numpy())
    dy = rng.randn(*y.shape)
    yt.backward(t(dy, False))
    assert np.allclose(ops.maxpool_vjp(x, y, dy, 2), xt.grad.numpy())
    xt = t(x)
    yt = F.avg_pool2d(xt, 4)
    assert np.allclose(ops.avgpool_fwd(x, 4), yt.detach().numpy())
    dy = rng.randn(*yt.shape)
    yt.backward(t(dy, False))
    assert np.allclose(ops.avgpool_vjp(x.shape, dy, 4), xt.grad.numpy())
    xt = t(x)
    yt = F.interpolate(xt, scale_factor=2, mode='nearest')
    assert np.allclose(ops.upscale_nearest_fwd(x), yt.detach().numpy())
    dy = rng.randn(*yt.shape)
    yt.backward(t(dy, False))
    assert np.allclose(ops.upscale_nearest_vjp(dy), xt.grad.numpy())


@pytest.mark.parametrize("hw", [(1, 1), (2, 2), (5, 3), (8, 8)])
def test_bilinear_closed_form_equals_theano_algorithm(hw):
    x = rng.randn(2, 3, *hw)
    a = ops.bilinear_up2_fwd(x)
    b = ops.bilinear_theano_literal(x, 2)
    assert a.shape == b.shape == (2, 3, 2 * hw[0], 2 * hw[1])
    assert np.allclose(a, b, atol=1e-12)
    # adjoint test <Ax, g> == <x, A^T g>
    g = rng.randn(*a.shape)
    assert np.isclose((a * g).sum(), (x * ops.bilinear_up2_vjp(g)).sum())


@pytest.mark.parametrize("case", [(2, 3, 4, 5, 7), (1, 2, 3, 2, 2), (1, 2, 2, 2, 3), (2, 4, 5, 8, 6), (1, 3, 2, 16, 16)])
def test_bilinear_conv_on_the_coarse_grid(case):
    """BilinearUpsample2DLayer(2) -> 3x3 'same' Conv2DLayer (p2p.py:204-267) == four collapsed coarse convolutions of the
    zero-extended input + the convolution of a frame of two fine rows and two fine columns (oracle/ops.py
    bilinear_conv_*; the statement csrc/conv_bilinear.hip is tested against): forward, and both gradients through the
    decomposition, against the literal up-sample + convolution -- odd sizes, 2 x 2 maps, both borders."""
    N, C, K, n1, n2 = case
    rng = np.random.RandomState(sum(case))
    x, W, b = rng.randn(N, C, n1, n2), rng.randn(K, C, 3, 3), rng.randn(K)
    ref = ops.conv2d_fwd(ops.bilinear_up2_fwd(x), W, b, 1, 1)
    main = ops.bilinear_conv_main(x, W)                         # [N, 4, K, n1, n2]
    Fr = ops.bilinear_conv_frame(x)
    assert np.count_nonzero(np.abs(Fr[:, :, 1:-2, 1:-2]).sum((0, 1))) == 0          # only the frame lines
    y = np.zeros_like(ref)
    for pq in range(4):
        y[:, :, (pq >> 1)::2, (pq & 1)::2] = main[:, pq]
    yF = ops.corr2d_fwd(Fr, ops._flip(W), 1, 0)
    touched = np.abs(yF).sum((0, 1)) > 1e-13
    edge = np.zeros_like(touched)
    edge[[0, -2, -1], :] = True
    edge[:, [0, -2, -1]] = True
    assert not np.any(touched & ~edge)                          # the frame reaches rows / columns 0, 2n-2, 2n-1 only
    assert np.allclose(y + yF + b.reshape(1, -1, 1, 1), ref, atol=1e-12)
    # collapsed taps: 9 / 6 / 6 / 4 non-zero
    Wc = ops.bilinear_conv_collapse(ops._flip(W))
    assert [int(np.count_nonzero(np.abs(w).sum((0, 1)))) for w in Wc] == [9, 6, 6, 4]
    # gradients through the decomposition == gradients of the literal form
    dy = rng.randn(*ref.shape)
    dxu, dW_ref, _ = ops.conv2d_vjp(ops.bilinear_up2_fwd(x), W, dy, 1, 1)
    dx_ref = ops.bilinear_up2_vjp(dxu)
    dWc = [ops.corr2d_bwd_weight(x, dy[:, :, (pq >> 1)::2, (pq & 1)::2], 1, 1, 3, 3) for pq in range(4)]
    dW_main = ops._flip(ops.bilinear_conv_expand(dWc))
    dW_frame = ops._flip(ops.corr2d_bwd_weight(Fr, dy, 1, 0, 3, 3))
    assert np.allclose(dW_main + dW_frame, dW_ref, atol=1e-10)
    dx_main = sum(ops.corr2d_bwd_input(dy[:, :, (pq >> 1)::2, (pq & 1)::2], Wc[pq], 1, 1, n1, n2) for pq in range(4))
    eps = 1e-6                                                  # the frame is linear in x: its vjp by a directional probe
    v = rng.randn(*x.shape)
    lhs = ((dx_ref - dx_main) * v).sum()
    rhs = (ops.corr2d_fwd(ops.bilinear_conv_frame(v), ops._flip(W), 1, 0) * dy).sum()
    assert abs(lhs - rhs) <= 1e-9 * (1 + abs(rhs))


def test_optimizers_against_torch_formulas():
    p, g = rng.randn(50), rng.randn(50)
    acc = np.abs(rng.randn(50))
    pn, an = ops.rmsprop_step(p, g, acc, 1e-2)
    an_ref = 0.9 * acc + 0.1 * g * g
    assert np.allclose(an, an_ref) and np.allclose(pn, p - 1e-2 * g / np.sqrt(an_ref + 1e-6))
    # adam: lasagne folds the bias correction into the step size (eps is NOT rescaled, unlike
    # torch.optim.Adam), so compare with torch run at the equivalent per-step eps
    m = v = np.zeros(50)
    pcur, tt = p, 0
    pt_ = p.copy()
    mt = vt = np.zeros(50)
    for k in range(1, 4):
        pcur, m, v, tt = ops.adam_step(pcur, g, m, v, tt, 1e-3)
        mt = 0.9 * mt + 0.1 * g
        vt = 0.999 * vt + 0.001 * g * g
        eps_equiv = 1e-8 / np.sqrt(1 - 0.999 ** k)
        pt_ = pt_ - 1e-3 * (mt / (1 - 0.9 ** k)) / (np.sqrt(vt / (1 - 0.999 ** k)) + eps_equiv)
    assert tt == 3 and np.allclose(pcur, pt_, atol=1e-12)


def _torch_step(state, Z, X, Y):
    """The whole train_fn forward + 4 gradient roots in torch, written independently."""
    cfg = state['cfg']
    P = {k: [t(a, True) for a in state['params'][k[0]][k[1]]] for k in step.NET_ORDER}
    z, x, y = t(Z, False), t(X, False), t(Y, False)

    def conv(h, W, b, s, p):
        return F.conv2d(h, flipt(W), b, stride=s, padding=p)

    def deconv(h, W, b, s):
        return F.conv_transpose2d(h, flipt(W), b, stride=s)

    def bn(h, beta, gamma):
        return F.batch_norm(h, None, None, gamma, beta, training=True, eps=1e-4)

    def bil(h):
        def ax(v, d):
            n = v.shape[d]
            idx = torch.clamp(torch.arange(n) + 1, max=n - 1)
            odd = 0.5 * (v + v.index_select(d, idx))
            return torch.stack([v, odd], dim=d + 1).reshape(*v.shape[:d], 2 * n, *v.shape[d + 1:])
        return ax(ax(h, 2), 3)

    g = cfg['gen_dcgan']
    q = P[('dcgan', 'gen')]
    h = z @ q[0] + q[1]
    h = bn(h, q[2], q[3]).reshape(-1, g['nch'], g['initial_size'], g['initial_size'])
    i = 6
    for _ in g['div']:
        h = F.leaky_relu(bn(conv(h, q[i], q[i + 1], 1, g['h'] // 2), q[i + 2], q[i + 3]), 0.2)
        h = F.interpolate(h, scale_factor=2, mode='nearest')
        i += 6
    gz = torch.sigmoid(conv(h, q[i], q[i + 1], 1, g['h'] // 2))

    def D(inp):
        d = cfg['disc_dcgan']
        q = P[('dcgan', 'disc')]
        h, i = inp, 0
        for _ in d['div']:
            h = F.max_pool2d(F.leaky_relu(conv(h, q[i], q[i + 1], 1, d['h'] // 2), 0.2), 2)
            i += 2
        h = F.relu(conv(h, q[i], q[i + 1], 1, d['h'] // 2))
        return h.mean(dim=(2, 3)).reshape(-1, 1)

    def U(inp):
        q = P[('p2p', 'gen')]
        L = int(np.log2(cfg['in_shp']))
        skips, h, i = {}, inp, 0
        for l in range(1, L):
            c = bn(conv(h, q[i], q[i + 1], 2, 1), q[i + 2], q[i + 3])
            skips[l] = c
            h = F.leaky_relu(c, 0.01)
            i += 6
        h = F.leaky_relu(bn(conv(h, q[i], q[i + 1], 1, 0), q[i + 2], q[i + 3]), 0.01)
        i += 6
        dd = bn(deconv(h, q[i], q[i + 1], 1), q[i + 2], q[i + 3])
        i += 6
        h = F.leaky_relu(torch.cat([dd, skips[L - 1]], 1), 0.01)
        for l in range(L - 2, 0, -1):
            if cfg['gen_p2p']['bilinear_upsample']:
                dd = conv(bil(h), q[i], q[i + 1], 1, 1)
            else:
                dd = deconv(h, q[i], q[i + 1], 2)
            dd = bn(dd, q[i + 2], q[i + 3])
            i += 6
            h = F.leaky_relu(torch.cat([dd, skips[l]], 1), 0.01)
        return torch.tanh(deconv(h, q[i], q[i + 1], 2))

    def Dp(a, b):
        q = P[('p2p', 'disc')]
        h, i = torch.cat([a, b], 1), 0
        for _ in cfg['disc_p2p']['mul_factor']:
            h = F.leaky_relu(conv(h, q[i], q[i + 1], 2, 1), 0.01)
            i += 2
        return conv(h, q[i], q[i + 1], 2, 1)

    d_real, d_fake = D(x), D(gz)
    ux = U(x)
    p_real, p_fake = Dp(x, y), Dp(x, ux)
    losses = {
        ('dcgan', 'gen'): ((d_fake - 1) ** 2).mean(),
        ('dcgan', 'disc'): ((d_real - 1) ** 2).mean() + (d_fake ** 2).mean(),
        ('p2p', 'gen'): ((p_fake - 1) ** 2).mean() + cfg['alpha'] * (ux - y).abs().mean(),
        ('p2p', 'disc'): ((p_real - 1) ** 2).mean() + (p_fake ** 2).mean(),
    }
    sp = step.specs(cfg)
    grads = {}
    for key in step.NET_ORDER:
        tr = [q for q, f in zip(P[key], sp[key].trainable) if f]
        grads[key] = [v.numpy() for v in torch.autograd.grad(losses[key], tr, retain_graph=True)]
    five = [losses[('dcgan', 'gen')], losses[('dcgan', 'disc')], ((p_fake - 1) ** 2).mean(),
            (ux - y).abs().mean(), losses[('p2p', 'disc')]]
    return [float(v) for v in five], grads, gz.detach().numpy(), ux.detach().numpy()


@pytest.mark.parametrize("bilinear", [True, False])
def test_full_step_losses_and_grads_vs_torch(bilinear):
    cfg = step.default_cfg(in_shp=32, latent_dim=24,
                           gen_dcgan=dict(nch=16, div=[2, 2, 4]),
                           disc_dcgan=dict(nch=16, div=[4, 2, 2]),
                           gen_p2p=dict(nf=4, bilinear_upsample=bilinear),
                           disc_p2p=dict(nf=4, mul_factor=[1, 2]))
    st = step.init_state(cfg, seed=3, dtype=np.float64)
    Z, X, Y = step.synthetic_batch(3, cfg, seed=5, dtype=np.float64)
    ref_losses, ref_grads, gz, ux = _torch_step(st, Z, X, Y)
    fw = step.forward(st, Z, X, Y)
    assert np.allclose(step.losses_of(fw), ref_losses, rtol=1e-10)
    assert np.allclose(fw['gz'].v, gz, atol=1e-12) and np.allclose(fw['ux'].v, ux, atol=1e-12)
    grads = step.gradients(fw, st)
    for key in step.NET_ORDER:
        for a, b in zip(grads[key], ref_grads[key]):
            assert a.shape == b.shape
            assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(b) + 1e-13, key  # (bias before BN: grad == 0)
