"""The four networks of the hot path as plain forward functions (TEST INFRASTRUCTURE ONLY).

Each function follows the reference architecture file line by line (cited inline, paths
under /root/reference) and is written independently of the product's layer-graph in
gan_heightmaps_amd/, so a wrong slope / order / shape on either side shows up as a parity
failure.  Parameters are flat lists in lasagne ``get_all_param_values`` order (SURVEY.md
Appendix B): Dense/Conv/Deconv [W, b]; BatchNorm [beta, gamma, mean, inv_std].

Generalisations beyond the reference (needed for BASELINE configs 1 and 5 and for
CPU-sized tests; identical to the reference at in_shp = nch = 512):
  * default_discriminator: final average pool is in_shp // 2**len(div) instead of
    nch // 2**len(div) (architectures/dcgan.py:51 uses nch as a stand-in for in_shp).
  * g_unet: any power-of-two in_shp >= 4 (reference asserts 512, architectures/p2p.py:137);
    level l has nf*min(2**(l-1), 8) channels, which reproduces 64..512 at 512 px.
"""
import numpy as np

from . import ops
from . import tape as T


class ParamSpec:
    def __init__(self):
        self.names, self.shapes, self.kinds = [], [], []

    def add(self, name, shape, kind):
        self.names.append(name)
        self.shapes.append(tuple(int(s) for s in shape))
        self.kinds.append(kind)

    def conv(self, name, co, ci, k):
        self.add(name + ".W", (co, ci, k, k), 'W')
        self.add(name + ".b", (co,), 'b')

    def deconv(self, name, ci, co, k):
        self.add(name + ".W", (ci, co, k, k), 'W')
        self.add(name + ".b", (co,), 'b')

    def bn(self, name, c):
        for kind in ('beta', 'gamma', 'mean', 'inv_std'):
            self.add(name + "." + kind, (c,), kind)

    @property
    def trainable(self):
        return [k in ('W', 'b', 'beta', 'gamma') for k in self.kinds]

    def count(self, trainable_only=False):
        tr = self.trainable
        return sum(int(np.prod(s)) for s, t in zip(self.shapes, tr) if t or not trainable_only)

    def init(self, rng, dtype=np.float32):
        """GlorotUniform W, zero b / beta / mean, one gamma / inv_std (lasagne defaults),
        drawn in layer-construction order from ``rng``."""
        out = []
        for shape, kind in zip(self.shapes, self.kinds):
            if kind == 'W':
                out.append(ops.glorot_uniform(rng, shape, dtype))
            elif kind in ('gamma', 'inv_std'):
                out.append(np.ones(shape, dtype))
            else:
                out.append(np.zeros(shape, dtype))
        return out


class _Cursor:
    """Walks a flat param list; records BN batch statistics for the running update."""

    def __init__(self, nodes):
        self.nodes = nodes
        self.i = 0
        self.bn_stats = {}      # index of 'mean' param -> (mu, inv)

    def take(self, n):
        out = self.nodes[self.i:self.i + n]
        self.i += n
        return out


def _bn(cur, x, deterministic):
    beta, gamma, mean, inv_std = cur.take(4)
    if deterministic:
        return T.bn_infer(x, beta, gamma, mean.v, inv_std.v)
    y, mu, inv = T.bn_train(x, beta, gamma)
    cur.bn_stats[cur.i - 2] = (mu, inv)
    return y


# --------------------------------------------------------------------------------------
# DCGAN generator: architectures/dcgan.py:14-33
# --------------------------------------------------------------------------------------


def dcgan_gen_spec(latent_dim, is_a_grayscale, nch=512, h=5, initial_size=4,
                   div=(2, 2, 4, 4, 8, 8, 16)):
    sp = ParamSpec()
    sp.add("dense.W", (latent_dim, nch * initial_size * initial_size), 'W')      # :16
    sp.add("dense.b", (nch * initial_size * initial_size,), 'b')
    sp.bn("dense_bn", nch * initial_size * initial_size)                          # :17
    prev = nch
    for i, d in enumerate(div):
        n = nch // d                                                              # :19 (py2 int /)
        sp.conv("g_conv%d" % (i + 1), n, prev, h)                                 # :22
        sp.bn("g_bn%d" % (i + 1), n)                                              # :23
        prev = n
    sp.conv("g_out", 1 if is_a_grayscale else 3, prev, h)                         # :32
    return sp


def dcgan_gen_fwd(P, z, nch=512, h=5, initial_size=4, div=(2, 2, 4, 4, 8, 8, 16),
                  bilinear_upsample=False, deterministic=False):
    """P: list of tape nodes in spec order; z: node [B, latent]. -> (out node, cursor)"""
    cur = _Cursor(P)
    W, b = cur.take(2)
    x = T.dense(z, W, b)                                                          # :16 linear
    x = _bn(cur, x, deterministic)                                                # :17 axes=(0,)
    x = T.reshape(x, (-1, nch, initial_size, initial_size))                       # :18
    for _ in div:
        W, b = cur.take(2)
        x = T.conv2d(x, W, b, 1, h // 2)                                          # :22 pad='same'
        x = _bn(cur, x, deterministic)                                            # :23
        x = T.lrelu(x, 0.2)                                                       # :24
        x = T.bilinear_up2(x) if bilinear_upsample else T.upscale_nearest(x, 2)   # :27-31
    W, b = cur.take(2)
    x = T.sigmoid(T.conv2d(x, W, b, 1, h // 2))                                   # :32
    return x, cur


# --------------------------------------------------------------------------------------
# DCGAN discriminator: architectures/dcgan.py:35-58
# --------------------------------------------------------------------------------------


def dcgan_disc_spec(in_shp, is_a_grayscale, nch=512, h=5, div=(8, 4, 4, 2, 2, 1, 1), bn=False):
    sp = ParamSpec()
    prev = 1 if is_a_grayscale else 3
    for i, d in enumerate(div):
        n = nch // d
        sp.conv("d_conv%d" % (i + 1), n, prev, h)                                 # :42
        if bn:
            sp.bn("d_bn%d" % (i + 1), n)                                          # :44
        prev = n
    sp.conv("d_out", 1, prev, h)                                                  # :50
    return sp


def dcgan_disc_fwd(P, x, in_shp, h=5, div=(8, 4, 4, 2, 2, 1, 1), bn=False,
                   nonlinearity='sigmoid', pool_mode='max', deterministic=False):
    cur = _Cursor(P)
    for _ in div:
        W, b = cur.take(2)
        x = T.conv2d(x, W, b, 1, h // 2)                                          # :42
        if bn:
            x = _bn(cur, x, deterministic)                                        # :44
        x = T.lrelu(x, 0.2)                                                       # :45
        x = T.maxpool(x, 2) if pool_mode == 'max' else T.avgpool(x, 2)            # :46-49
    W, b = cur.take(2)
    x = T.relu(T.conv2d(x, W, b, 1, h // 2))          # :50  NO nonlinearity kwarg => lasagne default rectify
    red = in_shp // (2 ** len(div))                   # :51  (reference: nch // 2**len(div); equal at 512)
    x = T.avgpool(x, red)                                                         # :52
    x = T.reshape(x, (-1, 1))                                                     # :55
    return T.act(x, nonlinearity), cur                                            # :56


# --------------------------------------------------------------------------------------
# pix2pix U-Net generator: architectures/p2p.py:126-276
# --------------------------------------------------------------------------------------


def _unet_levels(in_shp):
    L = int(np.log2(in_shp))
    assert 2 ** L == in_shp and L >= 2
    return L


def _unet_ch(nf, l):
    return nf * min(2 ** (l - 1), 8)


def unet_spec(in_shp, is_a_grayscale, is_b_grayscale, nf=64, bilinear_upsample=False):
    L = _unet_levels(in_shp)
    sp = ParamSpec()
    prev = 1 if is_a_grayscale else 3
    for l in range(1, L):                               # conv1..conv8 at 512 (:145-190)
        c = _unet_ch(nf, l)
        sp.conv("conv%d" % l, c, prev, 3)
        sp.bn("conv%d_bn" % l, c)
        prev = c
    sp.conv("conv%d" % L, nf * 8, prev, 2)              # conv9: k=2 valid (:193)
    sp.bn("conv%d_bn" % L, nf * 8)
    sp.deconv("dconv1", nf * 8, nf * 8, 2)              # :197  k=2 s=1
    sp.bn("dconv1_bn", nf * 8)
    prev = nf * 8 + _unet_ch(nf, L - 1)
    j = 2
    for l in range(L - 2, 0, -1):                       # dconv2..dconv8 at 512 (:205-268)
        c = _unet_ch(nf, l)
        if bilinear_upsample:
            sp.conv("dconv%d" % j, c, prev, 3)          # Convolution(.., s=1) after bilinear
        else:
            sp.deconv("dconv%d" % j, prev, c, 2)        # Deconvolution k=2 s=2
        sp.bn("dconv%d_bn" % j, c)
        prev = 2 * c
        j += 1
    sp.deconv("dconv%d" % j, prev, 1 if is_b_grayscale else 3, 2)   # dconv9 (:272)
    return sp


def unet_fwd(P, x, in_shp, act='tanh', bilinear_upsample=False, deterministic=False):
    L = _unet_levels(in_shp)
    cur = _Cursor(P)
    skips = {}
    for l in range(1, L):
        W, b = cur.take(2)
        c = T.conv2d(x, W, b, 2, 1)                     # Convolution k=3 s=2 'same' (:20-21)
        c = _bn(cur, c, deterministic)
        skips[l] = c                                    # skip takes the post-BN PRE-activation tensor
        x = T.lrelu(c, 0.01)                            # leaky_rectify == LeakyRectify(0.01)
    W, b = cur.take(2)
    c = T.conv2d(x, W, b, 1, 0)                         # :193 k=2 s=1 valid
    c = _bn(cur, c, deterministic)
    x = T.lrelu(c, 0.01)
    W, b = cur.take(2)
    d = T.deconv2d(x, W, b, 1)                          # :197
    d = _bn(cur, d, deterministic)
    x = T.lrelu(T.concat([d, skips[L - 1]]), 0.01)      # :202-203 concat THEN leaky_rectify
    for l in range(L - 2, 0, -1):
        W, b = cur.take(2)
        if bilinear_upsample:
            d = T.conv2d(T.bilinear_up2(x), W, b, 1, 1)  # :208-209
        else:
            d = T.deconv2d(x, W, b, 2)                   # :206
        d = _bn(cur, d, deterministic)
        x = T.lrelu(T.concat([d, skips[l]]), 0.01)
    W, b = cur.take(2)
    out = T.deconv2d(x, W, b, 2)                        # :272
    return T.act(out, act), cur                         # :275


# --------------------------------------------------------------------------------------
# PatchGAN discriminator: architectures/p2p.py:278-292
# --------------------------------------------------------------------------------------


def patchgan_spec(in_shp, is_a_grayscale, is_b_grayscale, nf=32, mul_factor=(1, 2, 4, 8), bn=False):
    sp = ParamSpec()
    prev = (1 if is_a_grayscale else 3) + (1 if is_b_grayscale else 3)
    for i, m in enumerate(mul_factor):
        sp.conv("pd_conv%d" % (i + 1), nf * m, prev, 3)
        if bn:
            sp.bn("pd_bn%d" % (i + 1), nf * m)
        prev = nf * m
    sp.conv("pd_out", 1, prev, 3)
    return sp


def patchgan_fwd(P, a, b_img, act='sigmoid', mul_factor=(1, 2, 4, 8), bn=False, deterministic=False):
    cur = _Cursor(P)
    x = T.concat([a, b_img])                            # :281
    for _ in mul_factor:
        W, b = cur.take(2)
        x = T.conv2d(x, W, b, 2, 1)                     # :285
        x = T.lrelu(x, 0.01)                            # :286
        if bn:
            x = _bn(cur, x, deterministic)              # :287-288 (BN AFTER the nonlinearity)
    W, b = cur.take(2)
    x = T.conv2d(x, W, b, 2, 1)                         # :289
    return T.act(x, act), cur                           # :290
