"""pix2pix U-Net generator / PatchGAN discriminator with the reference's signatures
(/root/reference/architectures/p2p.py: helpers :20-27, g_unet_256 :29-122, g_unet :126-276,
discriminator :278-292, discriminator2 :294-308, fake_* :314-325).

The U-Net is generated from a level table instead of being unrolled: level l (1-based) works at
in_shp / 2**l pixels with nf * min(2**(l-1), 8) channels, which is 64,128,256,512,512,... at nf=64.
"""
import math

from ..layers import (InputLayer, Conv2DLayer, Deconv2DLayer, BatchNormLayer, NonlinearityLayer, ConcatLayer,
                      DropoutLayer)
from ..nonlinearities import leaky_rectify, linear, sigmoid, tanh
from .layers import BilinearUpsample2DLayer


def Convolution(layer, f, k=3, s=2, border_mode='same', **kwargs):
    return Conv2DLayer(layer, num_filters=f, filter_size=(k, k), stride=(s, s), pad=border_mode, nonlinearity=linear)


def Deconvolution(layer, f, k=2, s=2, **kwargs):
    return Deconv2DLayer(layer, num_filters=f, filter_size=(k, k), stride=(s, s), nonlinearity=linear)


def concatenate_layers(layers, **kwargs):
    return ConcatLayer(layers, axis=1)


def _channels(in_shp):
    return lambda nf, level: nf * min(2 ** (level - 1), 8)


def _unet(in_shp, is_a_grayscale, is_b_grayscale, nf, act, dropout_p, num_repeats, bilinear_upsample):
    depth = int(round(math.log(in_shp, 2)))
    if 2 ** depth != in_shp or depth < 2:
        raise ValueError("in_shp must be a power of two >= 4")
    width = _channels(in_shp)
    x = InputLayer((None, 1 if is_a_grayscale else 3, in_shp, in_shp))
    skips = {}
    # encoder: 3x3 stride-2 conv -> BN -> leaky_rectify(0.01); the skip keeps the post-BN PRE-activation tensor
    for level in range(1, depth):
        pre = BatchNormLayer(Convolution(x, width(nf, level)))
        skips[level] = pre
        x = NonlinearityLayer(pre, nonlinearity=leaky_rectify)
        for _ in range(num_repeats):
            x = NonlinearityLayer(BatchNormLayer(Convolution(x, width(nf, level), s=1, k=3)), nonlinearity=leaky_rectify)
    # bottleneck: 2x2 valid conv to 1x1, then 2x2 stride-1 transposed conv back to 2x2
    x = NonlinearityLayer(BatchNormLayer(Convolution(x, nf * 8, k=2, s=1, border_mode='valid')),
                          nonlinearity=leaky_rectify)
    up = BatchNormLayer(Deconvolution(x, nf * 8, k=2, s=1))
    drops_left = 3
    if dropout_p > 0:
        up = DropoutLayer(up, p=dropout_p)
    drops_left -= 1
    x = NonlinearityLayer(concatenate_layers([up, skips[depth - 1]]), nonlinearity=leaky_rectify)
    # decoder: x2 (transposed conv k2 s2, or Theano-bilinear + 3x3 conv) -> BN -> concat skip -> leaky_rectify
    for level in range(depth - 2, 0, -1):
        if bilinear_upsample:
            up = Convolution(BilinearUpsample2DLayer(x, 2), width(nf, level), s=1)
        else:
            up = Deconvolution(x, width(nf, level))
        up = BatchNormLayer(up)
        if dropout_p > 0 and drops_left > 0:
            up = DropoutLayer(up, p=dropout_p)
        drops_left -= 1
        x = NonlinearityLayer(concatenate_layers([up, skips[level]]), leaky_rectify)
    out = Deconvolution(x, 1 if is_b_grayscale else 3)
    return NonlinearityLayer(out, act)


def g_unet(in_shp, is_a_grayscale, is_b_grayscale, nf=64, act=tanh, dropout=False, num_repeats=0,
           bilinear_upsample=False):
    """512-px U-Net of the reference (p2p.py:126-276).  The reference asserts in_shp == 512; any power of
    two >= 4 is accepted here (BASELINE config 5 uses 1024) with the same per-level rule."""
    return _unet(in_shp, is_a_grayscale, is_b_grayscale, nf, act, 0.5 if dropout else 0., num_repeats,
                 bilinear_upsample)


def g_unet_256(in_shp, is_a_grayscale, is_b_grayscale, nf=64, act=tanh, dropout=0.):
    """256-px variant (p2p.py:29-122): same table one level shallower; ``dropout`` is the rate."""
    assert in_shp in [256]
    return _unet(in_shp, is_a_grayscale, is_b_grayscale, nf, act, dropout, 0, False)


def _patch_discriminator(in_shp, is_a_grayscale, is_b_grayscale, nf, act, mul_factor, num_repeats, bn_rule):
    i_a = InputLayer((None, 1 if is_a_grayscale else 3, in_shp, in_shp))
    i_b = InputLayer((None, 1 if is_b_grayscale else 3, in_shp, in_shp))
    x = concatenate_layers([i_a, i_b])
    for idx, m in enumerate(mul_factor):
        for r in range(num_repeats + 1):
            x = Convolution(x, nf * m, s=2 if r == 0 else 1)
            x = NonlinearityLayer(x, leaky_rectify)
            if bn_rule(idx):
                x = BatchNormLayer(x)          # BN comes AFTER the nonlinearity here (p2p.py:286-288)
    x = Convolution(x, 1)
    return {"inputs": [i_a, i_b], "out": NonlinearityLayer(x, act)}


def discriminator(in_shp, is_a_grayscale, is_b_grayscale, nf=32, act=sigmoid, mul_factor=[1, 2, 4, 8],
                  num_repeats=0, bn=False):
    """PatchGAN (p2p.py:278-292): returns {"inputs": [layer_a, layer_b], "out": layer} (pix2pix.py:46-49)."""
    return _patch_discriminator(in_shp, is_a_grayscale, is_b_grayscale, nf, act, mul_factor, num_repeats,
                                lambda idx: bn)


def discriminator2(in_shp, is_a_grayscale, is_b_grayscale, nf=32, act=sigmoid, mul_factor=[1, 2, 4, 8],
                   num_repeats=0):
    """p2p.py:294-308: BN on every block but the first."""
    return _patch_discriminator(in_shp, is_a_grayscale, is_b_grayscale, nf, act, mul_factor, num_repeats,
                                lambda idx: idx != 0)


# debugging architectures (p2p.py:314-325)

def fake_generator(is_a_grayscale, is_b_grayscale, act=tanh, in_shp=512):
    i = InputLayer((None, 1 if is_a_grayscale else 3, in_shp, in_shp))
    c = Convolution(i, f=1 if is_b_grayscale else 3, s=1)
    return NonlinearityLayer(c, act)


def fake_discriminator(is_a_grayscale, is_b_grayscale, in_shp=512):
    i_a = InputLayer((None, 1 if is_a_grayscale else 3, in_shp, in_shp))
    i_b = InputLayer((None, 1 if is_b_grayscale else 3, in_shp, in_shp))
    c = Convolution(concatenate_layers([i_a, i_b]), 1)
    return {"inputs": [i_a, i_b], "out": c}
