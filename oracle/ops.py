"""Numpy restatement of the Theano/Lasagne ops on the gan-heightmaps hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity numerics unpinned).

Every op is a pair ``fwd`` / ``vjp`` written from the published semantics of the
third-party library the reference calls; the reference call site
that relies on it is cited next to each function (paths under /root/reference).

The convolution follows Theano's CPU path (CorrMM): per-image im2col followed by one
GEMM, and col2im for the input gradient -- the op-level contract the HIP kernels replace
(traceback naming CorrMM_gradInputs in
lasagne/notebooks/gaussian_blur.ipynb:600-610).
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

# --------------------------------------------------------------------------------------
# cross-correlation core (im2col + GEMM), used by conv2d / deconv2d below
# --------------------------------------------------------------------------------------


def _im2col(xp_n, kh, kw, s, Ho, Wo):
    """xp_n: one padded image [C, Hp, Wp] -> [C*kh*kw, Ho*Wo] (row order c, a, b)."""
    C = xp_n.shape[0]
    win = sliding_window_view(xp_n, (kh, kw), axis=(1, 2))[:, ::s, ::s][:, :Ho, :Wo]
    return np.ascontiguousarray(win.transpose(0, 3, 4, 1, 2)).reshape(C * kh * kw, Ho * Wo)


def out_size(n, k, s, pad):
    return (n + 2 * pad - k) // s + 1


def corr2d_fwd(x, Wc, s, pad):
    """y[n,co,i,j] = sum_{c,a,b} xpad[n,c,i*s+a,j*s+b] * Wc[co,c,a,b]."""
    N, C, H, W_ = x.shape
    Co, C2, kh, kw = Wc.shape
    assert C == C2
    Ho, Wo = out_size(H, kh, s, pad), out_size(W_, kw, s, pad)
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))) if pad else x
    Wm = Wc.reshape(Co, -1)
    out = np.empty((N, Co, Ho, Wo), x.dtype)
    for n in range(N):
        out[n] = (Wm @ _im2col(xp[n], kh, kw, s, Ho, Wo)).reshape(Co, Ho, Wo)
    return out


def corr2d_bwd_weight(x, dy, s, pad, kh, kw):
    """dWc[co,c,a,b] = sum_{n,i,j} dy[n,co,i,j] * xpad[n,c,i*s+a,j*s+b]."""
    N, C, H, W_ = x.shape
    Co, Ho, Wo = dy.shape[1:]
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))) if pad else x
    dWm = np.zeros((Co, C * kh * kw), x.dtype)
    for n in range(N):
        dWm += dy[n].reshape(Co, -1) @ _im2col(xp[n], kh, kw, s, Ho, Wo).T
    return dWm.reshape(Co, C, kh, kw)


def corr2d_bwd_input(dy, Wc, s, pad, H, W_):
    """Adjoint of corr2d_fwd w.r.t. x (col2im of Wc^T dy), x of spatial size H x W_."""
    N, Co, Ho, Wo = dy.shape
    _, C, kh, kw = Wc.shape
    Wm = Wc.reshape(Co, -1)
    dxp = np.zeros((N, C, H + 2 * pad, W_ + 2 * pad), dy.dtype)
    for n in range(N):
        colg = (Wm.T @ dy[n].reshape(Co, -1)).reshape(C, kh, kw, Ho, Wo)
        for a in range(kh):
            for b in range(kw):
                dxp[n, :, a:a + s * Ho:s, b:b + s * Wo:s] += colg[:, a, b]
    return dxp[:, :, pad:pad + H, pad:pad + W_] if pad else dxp


def _flip(W):
    return W[:, :, ::-1, ::-1]


# --------------------------------------------------------------------------------------
# Conv2DLayer  (architectures/dcgan.py:22,32,42,50; architectures/p2p.py:20-21)
# Lasagne default flip_filters=True => TRUE convolution: the filter is flipped.
# --------------------------------------------------------------------------------------


def resolve_pad(pad, k):
    if pad == 'same':
        assert k % 2 == 1, "pad='same' needs an odd filter"
        return k // 2
    if pad == 'valid':
        return 0
    return int(pad)


def conv2d_fwd(x, W, b, stride=1, pad=0):
    y = corr2d_fwd(x, _flip(W), stride, pad)
    return y + b[None, :, None, None]


def conv2d_vjp(x, W, dy, stride=1, pad=0):
    """-> (dx, dW, db)."""
    kh, kw = W.shape[2:]
    dx = corr2d_bwd_input(dy, _flip(W), stride, pad, x.shape[2], x.shape[3])
    dW = np.ascontiguousarray(_flip(corr2d_bwd_weight(x, dy, stride, pad, kh, kw)))
    return dx, dW, dy.sum(axis=(0, 2, 3))


# --------------------------------------------------------------------------------------
# Deconv2DLayer == TransposedConv2DLayer (architectures/p2p.py:23-24 via :197,272)
# W[Cin,Cout,kh,kw]; defaults crop=0, flip_filters=False; it is the exact adjoint of the
# true convolution above with W read as [Cout_fwd=Cin, Cin_fwd=Cout].
#   out[n,co,i*s+a,j*s+b] += x[n,ci,i,j] * W[ci,co,k-1-a,k-1-b];  out size (in-1)*s+k-2*crop
# --------------------------------------------------------------------------------------


def deconv2d_fwd(x, W, b, stride=1, crop=0):
    kh, kw = W.shape[2:]
    H = (x.shape[2] - 1) * stride + kh - 2 * crop
    W_ = (x.shape[3] - 1) * stride + kw - 2 * crop
    y = corr2d_bwd_input(x, _flip(W), stride, crop, H, W_)
    return y + b[None, :, None, None]


def deconv2d_vjp(x, W, dy, stride=1, crop=0):
    kh, kw = W.shape[2:]
    dx = corr2d_fwd(dy, _flip(W), stride, crop)
    dW = np.ascontiguousarray(_flip(corr2d_bwd_weight(dy, x, stride, crop, kh, kw)))
    return dx, dW, dy.sum(axis=(0, 2, 3))


# --------------------------------------------------------------------------------------
# DenseLayer (architectures/dcgan.py:16): y = x @ W + b, W[in, units]
# --------------------------------------------------------------------------------------


def dense_fwd(x, W, b):
    return x @ W + b


def dense_vjp(x, W, dy):
    return dy @ W.T, x.T @ dy, dy.sum(axis=0)


# --------------------------------------------------------------------------------------
# BatchNormLayer (architectures/dcgan.py:17,23,44; every BatchNormLayer in p2p.py)
# axes = all but 1, epsilon=1e-4, alpha=0.1, biased variance, running average of the
# INVERSE std (not of the variance).
# --------------------------------------------------------------------------------------

BN_EPS = 1e-4
BN_ALPHA = 0.1


def _bn_axes(x):
    return (0,) + tuple(range(2, x.ndim))


def _bshape(x):
    return (1, -1) + (1,) * (x.ndim - 2)


def bn_train_fwd(x, beta, gamma):
    """-> (y, mu, inv) with the batch statistics used."""
    ax = _bn_axes(x)
    mu = x.mean(axis=ax)
    var = x.var(axis=ax)
    inv = 1.0 / np.sqrt(var + x.dtype.type(BN_EPS))
    sh = _bshape(x)
    y = (x - mu.reshape(sh)) * (gamma * inv).reshape(sh) + beta.reshape(sh)
    return y, mu, inv


def bn_train_vjp(x, gamma, mu, inv, dy):
    """-> (dx, dbeta, dgamma); gradients flow through mu and var."""
    ax = _bn_axes(x)
    sh = _bshape(x)
    xhat = (x - mu.reshape(sh)) * inv.reshape(sh)
    dbeta = dy.sum(axis=ax)
    dgamma = (dy * xhat).sum(axis=ax)
    m = x.size // x.shape[1]
    dx = (gamma * inv).reshape(sh) * (dy - (dbeta / m).reshape(sh) - xhat * (dgamma / m).reshape(sh))
    return dx, dbeta, dgamma


def bn_running_update(mean, inv_std, mu, inv):
    a = mean.dtype.type(BN_ALPHA)
    return (1 - a) * mean + a * mu, (1 - a) * inv_std + a * inv


def bn_infer_fwd(x, beta, gamma, mean, inv_std):
    sh = _bshape(x)
    return (x - mean.reshape(sh)) * (gamma * inv_std).reshape(sh) + beta.reshape(sh)


# --------------------------------------------------------------------------------------
# InstanceNorm (BASELINE north_star names it beside BatchNorm; NO reference call site: the reference's architecture files
# build BatchNormLayer only, architectures/p2p.py:146-268 -- parity of this op is pinned on torch.nn.functional.instance_norm,
# tests/test_oracle_vs_torch.py).  BatchNormLayer's conventions with the axes (2, 3): biased variance, eps inside the root.
# --------------------------------------------------------------------------------------
def in_fwd(x, beta, gamma, eps=BN_EPS):
    """-> (y, mu[N, C], inv[N, C])"""
    mu = x.mean(axis=(2, 3))
    var = x.var(axis=(2, 3))
    inv = 1.0 / np.sqrt(var + x.dtype.type(eps))
    y = (x - mu[:, :, None, None]) * (gamma[None, :] * inv)[:, :, None, None] + beta[None, :, None, None]
    return y, mu, inv


def in_vjp(x, gamma, mu, inv, dy):
    """-> (dx, dbeta[C], dgamma[C]); gradients flow through mu and var of every instance"""
    xhat = (x - mu[:, :, None, None]) * inv[:, :, None, None]
    db = dy.sum(axis=(2, 3))                        # per instance
    dg = (dy * xhat).sum(axis=(2, 3))
    m = x.shape[2] * x.shape[3]
    dx = (gamma[None, :] * inv)[:, :, None, None] * (dy - (db / m)[:, :, None, None] - xhat * (dg / m)[:, :, None, None])
    return dx, db.sum(axis=0), dg.sum(axis=0)


# --------------------------------------------------------------------------------------
# nonlinearities (lasagne.nonlinearities)
#   LeakyRectify(a) is computed as 0.5(1+a)x + 0.5(1-a)|x|, gradient of |x| is sgn(x)
#   => slope at exactly 0 is 0.5(1+a).  leaky_rectify is the a=0.01 instance (all of
#   p2p.py); LeakyRectify(0.2) in dcgan.py:24,45.
# --------------------------------------------------------------------------------------


def lrelu_fwd(x, a):
    a = x.dtype.type(a)
    return 0.5 * (1 + a) * x + 0.5 * (1 - a) * np.abs(x)


def lrelu_vjp(x, a, dy):
    a = x.dtype.type(a)
    return dy * (0.5 * (1 + a) + 0.5 * (1 - a) * np.sign(x))


def relu_fwd(x):
    return np.maximum(x, 0)


def relu_vjp(x, dy):
    return dy * (x > 0)


def sigmoid_fwd(x):
    return 1.0 / (1.0 + np.exp(-x))


def sigmoid_vjp_from_out(y, dy):
    return dy * y * (1 - y)


def tanh_fwd(x):
    return np.tanh(x)


def tanh_vjp_from_out(y, dy):
    return dy * (1 - y * y)


# --------------------------------------------------------------------------------------
# Upscale2DLayer(scale_factor=2), mode 'repeat' (architectures/dcgan.py:31)
# --------------------------------------------------------------------------------------


def upscale_nearest_fwd(x, f=2):
    return x.repeat(f, axis=2).repeat(f, axis=3)


def upscale_nearest_vjp(dy, f=2):
    N, C, H, W_ = dy.shape
    return dy.reshape(N, C, H // f, f, W_ // f, f).sum(axis=(3, 5))


# --------------------------------------------------------------------------------------
# BilinearUpsample2DLayer -> theano bilinear_upsampling(ratio=2) (architectures/layers.py:22-26)
# Closed form, separable per axis: out[2m] = x[m]; out[2m+1] = (x[m] + x[min(m+1,n-1)])/2
# (edge-replicated input, stride-2 transposed conv with kernel [.5,1,.5], crop 3).
# bilinear_theano_literal() below transcribes that algorithm step by step; the closed
# form is tested against it.
# --------------------------------------------------------------------------------------


def _bilin_axis_fwd(x, axis):
    n = x.shape[axis]
    nxt = np.take(x, np.minimum(np.arange(n) + 1, n - 1), axis=axis)
    even = x
    odd = (x + nxt) * x.dtype.type(0.5)
    st = np.stack([even, odd], axis=axis + 1)
    shp = list(x.shape)
    shp[axis] = 2 * n
    return st.reshape(shp)


def bilinear_up2_fwd(x):
    return _bilin_axis_fwd(_bilin_axis_fwd(x, 2), 3)


def _bilin_axis_vjp(g, axis):
    n = g.shape[axis] // 2
    shp = list(g.shape)
    shp[axis:axis + 1] = [n, 2]
    g2 = g.reshape(shp)
    ge = np.take(g2, 0, axis=axis + 1)
    go = np.take(g2, 1, axis=axis + 1)
    dx = ge + go * g.dtype.type(0.5)
    # odd sample 2m+1 also reads x[min(m+1,n-1)]
    idx = np.minimum(np.arange(n) + 1, n - 1)
    half = go * g.dtype.type(0.5)
    dx = np.moveaxis(dx, axis, 0).copy()
    half = np.moveaxis(half, axis, 0)
    np.add.at(dx, idx, half)
    return np.moveaxis(dx, 0, axis)


def bilinear_up2_vjp(g):
    return _bilin_axis_vjp(_bilin_axis_vjp(g, 3), 2)


# ---- BilinearUpsample2DLayer(2) -> 3x3 'same' Conv2DLayer on the COARSE grid (architectures/p2p.py:204-267 with
# architectures/layers.py:13-26).  Test infrastructure for csrc/conv_bilinear.hip; the algebra, per axis, fine index -1 .. 2n:
#   U  = Theano's operator (above) with the convolution's ring of zeros,
#   U0 = the natural operator on x extended by zeros: u0[2m] = x[m], u0[2m+1] = (x[m] + x[m+1]) / 2 for m = -1 .. n,
#   U x = U0 x + D x,  (D x)[-1] = -x[0] / 2,  (D x)[2n-1] = +x[n-1] / 2, zero elsewhere
# so that  conv(U x U^T) = conv(U0 x U0^T) + conv(F),  F = (D x) U^T + U0 (x D^T): four zero-padded coarse convolutions with
# collapsed taps (one per output parity) plus a frame of two fine rows and two fine columns.
_BL_COEF = (np.array([[.5, 0, 0], [.5, 1, .5], [0, 0, .5]]),      # even outputs: [coarse tap r][fine tap a]
            np.array([[0, 0, 0], [1, .5, 0], [0, .5, 1]]))        # odd outputs: coarse offsets 0, +1


def bilinear_conv_collapse(Wcorr):
    """correlation taps [K, C, 3, 3] of the fine convolution -> [4][K, C, 3, 3] coarse correlation taps, class = 2 p + q
    (25 of the 36 non-zero)"""
    return [np.einsum('ra,sb,kcab->kcrs', _BL_COEF[p], _BL_COEF[q], Wcorr) for p in (0, 1) for q in (0, 1)]


def bilinear_conv_expand(dWc):
    """the transposed tap map: gradients of the four collapsed tap sets -> gradient of the fine correlation taps"""
    return sum(np.einsum('ra,sb,kcrs->kcab', _BL_COEF[pq >> 1], _BL_COEF[pq & 1], dWc[pq]) for pq in range(4))


def _bl_u0_axis(v, axis):
    """U0 along ``axis``: n samples -> fine positions -1 .. 2n (2n + 2 values)"""
    v = np.moveaxis(v, axis, -1)
    n = v.shape[-1]
    z = np.zeros_like(v[..., :1])
    vp = np.concatenate([z, v, z], -1)
    out = np.zeros(v.shape[:-1] + (2 * n + 2,), v.dtype)
    out[..., 1::2] = vp[..., 1:]
    out[..., 0::2] = 0.5 * (vp[..., :-1] + vp[..., 1:])
    return np.moveaxis(out, -1, axis)


def _bl_u_axis(v, axis):
    """U along ``axis``: Theano's operator embedded in -1 .. 2n (zero ring)"""
    pad = [(0, 0)] * v.ndim
    pad[axis] = (1, 1)
    return np.pad(_bilin_axis_fwd(v, axis), pad)


def bilinear_conv_frame(x):
    """F [N, C, 2 n1 + 2, 2 n2 + 2]: U x U^T - U0 x U0^T, non-zero on fine rows -1, 2 n1 - 1 and columns -1, 2 n2 - 1"""
    return _bl_u_axis(_bl_u_axis(x, 2) - _bl_u0_axis(x, 2), 3) + _bl_u0_axis(_bl_u_axis(x, 3) - _bl_u0_axis(x, 3), 2)


def bilinear_conv_main(x, W):
    """conv3x3(U0 x U0^T) as the four collapsed coarse convolutions, parity-planar [N, 4, K, n1, n2] (no bias)"""
    Wc = bilinear_conv_collapse(_flip(W))
    return np.stack([corr2d_fwd(x, Wc[pq], 1, 1) for pq in range(4)], 1)


def bilinear_theano_literal(x, ratio=2):
    """Literal transcription of theano.tensor.nnet.abstract_conv.bilinear_upsampling
    (use_1D_kernel=True): replicate the border once, transposed-conv each axis with the
    normalised 1-D bilinear kernel at stride ``ratio``, crop ``pad`` from each side."""
    N, C, h, w = x.shape
    k1 = np.concatenate([np.arange(1, ratio + 1), np.arange(ratio - 1, 0, -1)]).astype(x.dtype) / ratio
    pad = 2 * ratio - (ratio - 1) // 2 - 1
    xr = x.reshape(N * C, 1, h, w)
    xr = np.concatenate([xr[:, :, :1], xr, xr[:, :, -1:]], axis=2)
    xr = np.concatenate([xr[:, :, :, :1], xr, xr[:, :, :, -1:]], axis=3)

    def tconv_axis(a, axis):
        n = a.shape[axis]
        full = (n - 1) * ratio + len(k1)
        shp = list(a.shape)
        shp[axis] = full
        out = np.zeros(shp, a.dtype)
        for t, kv in enumerate(k1):
            sl = [slice(None)] * a.ndim
            sl[axis] = slice(t, t + ratio * (n - 1) + 1, ratio)
            out[tuple(sl)] += a * kv
        sl = [slice(None)] * a.ndim
        target = ratio * (n - 2)
        sl[axis] = slice(pad, pad + target)
        return out[tuple(sl)]

    up = tconv_axis(tconv_axis(xr, 2), 3)
    return up.reshape(N, C, h * ratio, w * ratio)


# --------------------------------------------------------------------------------------
# MaxPool2DLayer(pool_size=2) (architectures/dcgan.py:47): stride = pool, no pad,
# ignore_border=True.  Gradient goes to EVERY position equal to the window max.
# Pool2DLayer(mode='average_inc_pad') (architectures/dcgan.py:52): plain mean (no pad).
# --------------------------------------------------------------------------------------


def maxpool_fwd(x, p=2):
    N, C, H, W_ = x.shape
    Ho, Wo = H // p, W_ // p
    return x[:, :, :Ho * p, :Wo * p].reshape(N, C, Ho, p, Wo, p).max(axis=(3, 5))


def maxpool_vjp(x, y, dy, p=2):
    N, C, H, W_ = x.shape
    Ho, Wo = y.shape[2:]
    dx = np.zeros_like(x)
    yu = y.repeat(p, axis=2).repeat(p, axis=3)
    gu = dy.repeat(p, axis=2).repeat(p, axis=3)
    xs = x[:, :, :Ho * p, :Wo * p]
    dx[:, :, :Ho * p, :Wo * p] = np.where(xs == yu, gu, 0)
    return dx


def avgpool_fwd(x, p):
    N, C, H, W_ = x.shape
    Ho, Wo = H // p, W_ // p
    return x[:, :, :Ho * p, :Wo * p].reshape(N, C, Ho, p, Wo, p).mean(axis=(3, 5))


def avgpool_vjp(x_shape, dy, p):
    N, C, H, W_ = x_shape
    Ho, Wo = dy.shape[2:]
    dx = np.zeros(x_shape, dy.dtype)
    dx[:, :, :Ho * p, :Wo * p] = dy.repeat(p, axis=2).repeat(p, axis=3) / (p * p)
    return dx


# --------------------------------------------------------------------------------------
# objectives (pix2pix.py:102-121)
# --------------------------------------------------------------------------------------


def squared_error_mean(a, target):
    """lasagne.objectives.squared_error(a, t).mean()  (pix2pix.py:103,107-108)"""
    d = a - a.dtype.type(target)
    return (d * d).mean(), 2.0 * d / d.size


def bce_mean(p, target):
    """lasagne.objectives.binary_crossentropy(p, t).mean()  (pix2pix.py:105)"""
    t = p.dtype.type(target)
    loss = -(t * np.log(p) + (1 - t) * np.log(1 - p))
    return loss.mean(), (-(t / p) + (1 - t) / (1 - p)) / p.size


def l1_mean(a, b):
    """T.abs_(a - b).mean()  (pix2pix.py:115); d|x|/dx = sgn(x)."""
    d = a - b
    return np.abs(d).mean(), np.sign(d) / d.size


def l2_mean(a, b):
    d = a - b
    return (d * d).mean(), 2.0 * d / d.size


# --------------------------------------------------------------------------------------
# optimisers (lasagne.updates; chosen at experiments.py:116-117,
# default pix2pix.py:30)
# --------------------------------------------------------------------------------------


def rmsprop_step(p, g, acc, lr, rho=0.9, eps=1e-6):
    """acc <- rho*acc + (1-rho) g^2 ; p <- p - lr * g / sqrt(acc + eps)   (eps INSIDE sqrt)"""
    dt = p.dtype.type
    acc_new = dt(rho) * acc + dt(1 - rho) * g * g
    p_new = p - dt(lr) * g / np.sqrt(acc_new + dt(eps))
    return p_new, acc_new


def adam_step(p, g, m, v, t_prev, lr, b1=0.9, b2=0.999, eps=1e-8):
    """t = t_prev+1; a_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p <- p - a_t*m/(sqrt(v)+eps)"""
    dt = p.dtype.type
    t = t_prev + 1
    a_t = dt(lr) * np.sqrt(dt(1) - dt(b2) ** dt(t)) / (dt(1) - dt(b1) ** dt(t))
    m_new = dt(b1) * m + dt(1 - b1) * g
    v_new = dt(b2) * v + dt(1 - b2) * g * g
    p_new = p - a_t * m_new / (np.sqrt(v_new) + dt(eps))
    return p_new, m_new, v_new, t


# --------------------------------------------------------------------------------------
# init (lasagne.init.GlorotUniform gain 1)
# --------------------------------------------------------------------------------------


def glorot_uniform(rng, shape, dtype=np.float32):
    n1, n2 = shape[0], shape[1]
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    std = np.sqrt(2.0 / ((n1 + n2) * rf))
    a = np.sqrt(3.0) * std
    return rng.uniform(-a, a, size=shape).astype(dtype)


# --------------------------------------------------------------------------------------
# DropoutLayer(p, rescale=True): the mask is this build's own counter-based hash (ghm_dropout in
# csrc/elementwise.hip) -- Theano's MRG_RandomStreams cannot be reproduced; lasagne semantics otherwise
# (architectures/p2p.py:200-223, dcgan.py:25-26): y = x * mask / (1 - p), identity when deterministic
# --------------------------------------------------------------------------------------


def _lowbias32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x = (x * np.uint32(0x7feb352d)).astype(np.uint32)
    x ^= x >> np.uint32(15); x = (x * np.uint32(0x846ca68b)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def dropout_mask(shape, p, key, step):
    with np.errstate(over='ignore'):
        idx = np.arange(int(np.prod(shape)), dtype=np.uint32)
        h = _lowbias32((_lowbias32(idx ^ np.uint32(key & 0xffffffff)) +
                        np.uint32((step * 0x9e3779b9) & 0xffffffff)).astype(np.uint32))
    u = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).reshape(shape)

