"""DCGAN heightmap generator / discriminator with the reference's signatures
(/root/reference/architectures/dcgan.py:14-33 and :35-58)."""
from ..layers import (InputLayer, DenseLayer, BatchNormLayer, ReshapeLayer, Conv2DLayer, NonlinearityLayer,
                      DropoutLayer, Upscale2DLayer, MaxPool2DLayer, Pool2DLayer)
from ..nonlinearities import LeakyRectify, linear, sigmoid
from .layers import BilinearUpsample2DLayer


def _widths(nch, div):
    """nch/elem of the reference (py2 integer division, dcgan.py:19,39)."""
    out = []
    for d in div:
        if nch % d:
            raise ValueError("nch=%d is not divisible by %r" % (nch, d))
        out.append(nch // d)
    return out


def default_generator(latent_dim, is_a_grayscale, nch=512, h=5, initial_size=4, final_size=512,
                      div=[2, 2, 4, 4, 8, 8, 16], num_repeats=0, dropout_p=0., bilinear_upsample=False):
    """z[B, latent_dim] -> Dense -> BN -> [nch, s, s] -> ({conv hxh same, BN, LReLU(0.2)} x (num_repeats+1),
    x2 upsample) per entry of div -> conv hxh -> sigmoid.  ``final_size`` is unused, as in the reference."""
    net = InputLayer((None, latent_dim))
    net = DenseLayer(net, num_units=nch * initial_size * initial_size, nonlinearity=linear)
    net = BatchNormLayer(net)
    net = ReshapeLayer(net, (-1, nch, initial_size, initial_size))
    for width in _widths(nch, div):
        for _ in range(num_repeats + 1):
            net = Conv2DLayer(net, num_filters=width, filter_size=h, pad='same', nonlinearity=linear)
            net = BatchNormLayer(net)
            net = NonlinearityLayer(net, nonlinearity=LeakyRectify(0.2))
            if dropout_p > 0.:
                net = DropoutLayer(net, p=dropout_p)
        # nearest-neighbour by default: only the p2p generator of test1_nobn_bilin_both is bilinear
        net = BilinearUpsample2DLayer(net, factor=2) if bilinear_upsample else Upscale2DLayer(net, scale_factor=2)
    return Conv2DLayer(net, num_filters=1 if is_a_grayscale else 3, filter_size=h, pad='same', nonlinearity=sigmoid)


def default_discriminator(in_shp, is_a_grayscale, nch=512, h=5, div=[8, 4, 4, 2, 2, 1, 1], num_repeats=0, bn=False,
                          pool_mode='max', nonlinearity='sigmoid'):
    """image -> ({conv hxh same, (BN), LReLU(0.2)} x (num_repeats+1), 2x2 pool) per entry of div ->
    conv hxh to 1 channel (lasagne default rectify!) -> global average pool -> [B, 1] -> nonlinearity."""
    net = InputLayer((None, 1 if is_a_grayscale else 3, in_shp, in_shp))
    for width in _widths(nch, div):
        for _ in range(num_repeats + 1):
            net = Conv2DLayer(net, num_filters=width, filter_size=h, pad='same', nonlinearity=linear)
            if bn:
                net = BatchNormLayer(net)
            net = NonlinearityLayer(net, nonlinearity=LeakyRectify(0.2))
        if pool_mode == 'max':
            net = MaxPool2DLayer(net, pool_size=2)
        else:
            net = Pool2DLayer(net, pool_size=2, mode='average_inc_pad')
    # no nonlinearity kwarg in the reference (dcgan.py:50) => lasagne's default rectify is applied
    net = Conv2DLayer(net, num_filters=1, filter_size=h, pad='same')
    # the reference writes nch // 2**len(div) (dcgan.py:51), which equals the remaining map size only when
    # nch == in_shp (512); the remaining map size is what it stands for.
    reduction_factor = in_shp // (2 ** len(div))
    net = Pool2DLayer(net, pool_size=(reduction_factor, reduction_factor), mode='average_inc_pad')
    net = ReshapeLayer(net, (-1, 1))
    return NonlinearityLayer(net, nonlinearity)
