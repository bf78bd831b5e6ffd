"""The lasagne.layers vocabulary the reference's architecture files are written in
(/root/reference/architectures/dcgan.py:15-56, p2p.py:20-27,138-292, layers.py:13-26).

Layers here are pure graph descriptions (shapes, hyper-parameters, parameters with lasagne's order,
tags and initial values).  They carry no arithmetic: gan_heightmaps_amd.engine lowers a graph to
libghm.so kernel launches.
"""
import numpy as np

from . import init as _init
from .nonlinearities import as_nonlinearity, rectify


class Param:
    """One parameter tensor in lasagne layout. ``kind`` selects the HBM layout (conv weights are stored
    packed, see include/ghm.h); ``value`` is the host copy until an engine binds device storage."""

    def __init__(self, name, value, kind, trainable=True, regularizable=True):
        self.name = name
        self.value = np.ascontiguousarray(value, np.float32)
        self.shape = self.value.shape
        self.kind = kind
        self.tags = set()
        if trainable:
            self.tags.add('trainable')
        if regularizable:
            self.tags.add('regularizable')
        self.store = None          # engine.ParamStore once bound
        self.index = None

    def get_value(self):
        if self.store is not None:
            return self.store.download(self)
        return self.value.copy()

    def set_value(self, v):
        v = np.ascontiguousarray(v, np.float32)
        if v.shape != self.shape:
            raise ValueError("shape mismatch for %s: %s vs %s" % (self.name, v.shape, self.shape))
        self.value = v
        if self.store is not None:
            self.store.upload(self)

    def __repr__(self):
        return "<Param %s %s>" % (self.name, self.shape)


class Layer:
    def __init__(self, incoming, name=None):
        if isinstance(incoming, tuple):
            self.input_shape = incoming
            self.input_layer = None
        else:
            self.input_shape = incoming.output_shape
            self.input_layer = incoming
        self.name = name
        self.params = []

    @property
    def output_shape(self):
        return self.get_output_shape_for(self.input_shape)

    def get_output_shape_for(self, input_shape):
        return input_shape

    def add_param(self, spec, shape, name, kind, **tags):
        value = spec(shape) if callable(spec) else np.asarray(spec, np.float32)
        p = Param("%s.%s" % (self.name or type(self).__name__, name), value, kind, **tags)
        self.params.append(p)
        return p

    def get_params(self, **tags):
        out = []
        for p in self.params:
            if all((t in p.tags) == bool(v) for t, v in tags.items()):
                out.append(p)
        return out

    def __repr__(self):
        return "<%s %s>" % (type(self).__name__, self.name or "")


class MergeLayer(Layer):
    def __init__(self, incomings, name=None):
        self.input_shapes = [i if isinstance(i, tuple) else i.output_shape for i in incomings]
        self.input_layers = [None if isinstance(i, tuple) else i for i in incomings]
        self.name = name
        self.params = []

    @property
    def output_shape(self):
        return self.get_output_shape_for(self.input_shapes)


class InputLayer(Layer):
    def __init__(self, shape, input_var=None, name=None):
        self.shape = tuple(shape)
        self.input_layer = None
        self.input_shape = None
        self.name = name
        self.params = []

    @property
    def output_shape(self):
        return self.shape


def _pair(v):
    return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1]))


_DEFAULT = object()     # "argument not given" (lasagne's defaults are initialiser objects; an explicit None means no bias)


def _w_spec(W):
    """lasagne accepts an initialiser, a numpy array or a shared variable for W"""
    return _init.GlorotUniform() if W is None or W is _DEFAULT else W


def _b_spec(b):
    if b is None:
        raise NotImplementedError("b=None (a layer without bias): no reference architecture uses it and the packed "
                                  "parameter layout has no bias-free form")
    return _init.Constant(0.) if b is _DEFAULT else b


class DenseLayer(Layer):
    """y = x W + b, W[in, units]; default nonlinearity rectify (architectures/dcgan.py:16 passes linear)."""

    def __init__(self, incoming, num_units, W=_DEFAULT, b=_DEFAULT, nonlinearity=rectify, name=None):
        Layer.__init__(self, incoming, name)
        self.num_units = int(num_units)
        self.nonlinearity = as_nonlinearity(nonlinearity)
        n_in = int(np.prod(self.input_shape[1:]))
        self.W = self.add_param(_w_spec(W), (n_in, self.num_units), "W", 'dense_w')
        self.b = self.add_param(_b_spec(b), (self.num_units,), "b", 'vec', regularizable=False)

    def get_output_shape_for(self, s):
        return (s[0], self.num_units)


class Conv2DLayer(Layer):
    """lasagne Conv2DLayer: W[num_filters, C, kh, kw], b; stride 1, pad 0, rectify, flip_filters=True
    (true convolution) by default.  pad='same' -> k//2 (odd k), 'valid' -> 0."""

    def __init__(self, incoming, num_filters, filter_size, stride=(1, 1), pad=0, W=_DEFAULT, b=_DEFAULT,
                 nonlinearity=rectify, flip_filters=True, name=None):
        Layer.__init__(self, incoming, name)
        self.num_filters = int(num_filters)          # py3: nch/elem is a float (architectures/dcgan.py:19)
        if self.num_filters != num_filters:
            raise ValueError("num_filters must be integral, got %r" % (num_filters,))
        self.filter_size = _pair(filter_size)
        self.stride = _pair(stride)
        if not flip_filters:
            raise NotImplementedError("flip_filters=False")
        if pad == 'same':
            if self.filter_size[0] % 2 == 0 or self.filter_size[1] % 2 == 0:
                raise NotImplementedError("pad='same' requires odd filter size")
            self.pad = (self.filter_size[0] // 2, self.filter_size[1] // 2)
        elif pad == 'valid':
            self.pad = (0, 0)
        else:
            self.pad = _pair(pad)
        if self.pad[0] != self.pad[1] or self.stride[0] != self.stride[1]:
            raise NotImplementedError("anisotropic stride/pad")
        self.nonlinearity = as_nonlinearity(nonlinearity)
        cin = self.input_shape[1]
        self.W = self.add_param(_w_spec(W), (self.num_filters, cin) + self.filter_size, "W", 'conv_w')
        self.b = self.add_param(_b_spec(b), (self.num_filters,), "b", 'vec', regularizable=False)

    def get_output_shape_for(self, s):
        h = (s[2] + 2 * self.pad[0] - self.filter_size[0]) // self.stride[0] + 1
        w = (s[3] + 2 * self.pad[1] - self.filter_size[1]) // self.stride[1] + 1
        return (s[0], self.num_filters, h, w)


class TransposedConv2DLayer(Layer):
    """lasagne Deconv2DLayer: W[C_in, num_filters, kh, kw], crop=0, flip_filters=False; output size
    (in-1)*stride + k - 2*crop; exact adjoint of Conv2DLayer's true convolution."""

    def __init__(self, incoming, num_filters, filter_size, stride=(1, 1), crop=0, W=_DEFAULT, b=_DEFAULT,
                 nonlinearity=rectify, name=None):
        Layer.__init__(self, incoming, name)
        self.num_filters = int(num_filters)
        self.filter_size = _pair(filter_size)
        self.stride = _pair(stride)
        self.crop = _pair(0 if crop == 'valid' else crop)
        if self.crop != (0, 0) or self.stride[0] != self.stride[1]:
            raise NotImplementedError("crop != 0 / anisotropic stride")
        self.nonlinearity = as_nonlinearity(nonlinearity)
        cin = self.input_shape[1]
        self.W = self.add_param(_w_spec(W), (cin, self.num_filters) + self.filter_size, "W", 'conv_w')
        self.b = self.add_param(_b_spec(b), (self.num_filters,), "b", 'vec', regularizable=False)

    def get_output_shape_for(self, s):
        h = (s[2] - 1) * self.stride[0] + self.filter_size[0]
        w = (s[3] - 1) * self.stride[1] + self.filter_size[1]
        return (s[0], self.num_filters, h, w)


Deconv2DLayer = TransposedConv2DLayer


class BatchNormLayer(Layer):
    """axes = all but 1, epsilon 1e-4, alpha 0.1; params in creation order beta, gamma (trainable),
    mean, inv_std (not trainable)."""

    def __init__(self, incoming, axes='auto', epsilon=1e-4, alpha=0.1, name=None):
        Layer.__init__(self, incoming, name)
        if axes != 'auto':
            raise NotImplementedError("BatchNormLayer axes other than 'auto'")
        self.epsilon, self.alpha = float(epsilon), float(alpha)
        c = (self.input_shape[1],)
        self.beta = self.add_param(_init.Constant(0.), c, "beta", 'vec', regularizable=False)
        self.gamma = self.add_param(_init.Constant(1.), c, "gamma", 'vec')
        self.mean = self.add_param(_init.Constant(0.), c, "mean", 'vec', trainable=False, regularizable=False)
        self.inv_std = self.add_param(_init.Constant(1.), c, "inv_std", 'vec', trainable=False, regularizable=False)


class InstanceNormLayer(Layer):
    """BatchNormLayer's arithmetic with the statistics taken per (sample, channel) over the map (BASELINE north_star names
    InstanceNorm beside BatchNorm; the reference's architecture files only build BatchNormLayer, p2p.py:146-268, so this layer
    has no reference counterpart and follows BatchNormLayer's conventions: epsilon 1e-4 inside the root, biased variance,
    params in creation order beta, gamma).  No running statistics: a deterministic pass normalises with the sample's own."""

    def __init__(self, incoming, epsilon=1e-4, name=None):
        Layer.__init__(self, incoming, name)
        if len(self.input_shape) != 4:
            raise NotImplementedError("InstanceNormLayer on a %d-d input" % len(self.input_shape))
        self.epsilon = float(epsilon)
        c = (self.input_shape[1],)
        self.beta = self.add_param(_init.Constant(0.), c, "beta", 'vec', regularizable=False)
        self.gamma = self.add_param(_init.Constant(1.), c, "gamma", 'vec')


class NonlinearityLayer(Layer):
    def __init__(self, incoming, nonlinearity=rectify, name=None):
        Layer.__init__(self, incoming, name)
        self.nonlinearity = as_nonlinearity(nonlinearity)


class ReshapeLayer(Layer):
    def __init__(self, incoming, shape, name=None):
        Layer.__init__(self, incoming, name)
        self.shape = tuple(shape)

    def get_output_shape_for(self, s):
        known = int(np.prod([d for d in s[1:]]))
        out = list(self.shape)
        if out[0] == -1:
            rest = int(np.prod(out[1:]))
            if known % rest:
                raise ValueError("cannot reshape %s to %s" % (s, self.shape))
            # batch stays symbolic unless the trailing dims change the per-sample size
            out[0] = s[0] if rest == known else (None if s[0] is None else s[0] * known // rest)
            self.batch_factor = known // rest if rest != known else 1
        return tuple(out)


class Upscale2DLayer(Layer):
    """mode='repeat' nearest-neighbour upscaling (architectures/dcgan.py:31)."""

    def __init__(self, incoming, scale_factor, mode='repeat', name=None):
        Layer.__init__(self, incoming, name)
        self.scale_factor = _pair(scale_factor)
        if self.scale_factor != (2, 2) or mode != 'repeat':
            raise NotImplementedError("Upscale2DLayer other than 2x repeat")

    def get_output_shape_for(self, s):
        return (s[0], s[1], s[2] * 2, s[3] * 2)


class Pool2DLayer(Layer):
    def __init__(self, incoming, pool_size, stride=None, pad=(0, 0), ignore_border=True, mode='max', name=None):
        Layer.__init__(self, incoming, name)
        self.pool_size = _pair(pool_size)
        self.stride = self.pool_size if stride is None else _pair(stride)
        self.mode = mode
        if self.stride != self.pool_size or _pair(pad) != (0, 0) or self.pool_size[0] != self.pool_size[1]:
            raise NotImplementedError("pooling with stride != pool_size or padding")
        if mode not in ('max', 'average_inc_pad', 'average_exc_pad'):
            raise ValueError(mode)

    def get_output_shape_for(self, s):
        return (s[0], s[1], s[2] // self.pool_size[0], s[3] // self.pool_size[1])


class MaxPool2DLayer(Pool2DLayer):
    def __init__(self, incoming, pool_size, stride=None, pad=(0, 0), ignore_border=True, name=None):
        Pool2DLayer.__init__(self, incoming, pool_size, stride, pad, ignore_border, 'max', name)


class ConcatLayer(MergeLayer):
    def __init__(self, incomings, axis=1, name=None):
        MergeLayer.__init__(self, incomings, name)
        if axis != 1:
            raise NotImplementedError("ConcatLayer axis != 1")
        self.axis = axis

    def get_output_shape_for(self, shapes):
        s = shapes[0]
        return (s[0], sum(x[1] for x in shapes)) + tuple(s[2:])


class DropoutLayer(Layer):
    def __init__(self, incoming, p=0.5, name=None):
        Layer.__init__(self, incoming, name)
        self.p = float(p)


# ---- graph helpers (lasagne.layers.helper) -------------------------------------------------------

def get_all_layers(layer, treat_as_input=None):
    """Topological order, inputs first; for MergeLayers the incomings are visited in list order, so a
    U-Net's ConcatLayer([dconvN, convK]) yields encoder -> decoder as at g_unet.ipynb:416-480."""
    outs = layer if isinstance(layer, (list, tuple)) else [layer]
    seen, order = set(), []

    def visit(l):
        if id(l) in seen:
            return
        seen.add(id(l))
        for inc in (l.input_layers if isinstance(l, MergeLayer) else [l.input_layer]):
            if inc is not None:
                visit(inc)
        order.append(l)

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 10000))
    try:
        for o in outs:
            visit(o)
    finally:
        sys.setrecursionlimit(old)
    return order


def get_all_params(layer, **tags):
    out, seen = [], set()
    for l in get_all_layers(layer):
        for p in l.get_params(**tags):
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


def count_params(layer, **tags):
    return int(sum(int(np.prod(p.shape)) for p in get_all_params(layer, **tags)))


def get_all_param_values(layer, **tags):
    return [p.get_value() for p in get_all_params(layer, **tags)]


def set_all_param_values(layer, values, **tags):
    params = get_all_params(layer, **tags)
    if len(params) != len(values):
        raise ValueError("mismatch: got %d values to set %d parameters" % (len(values), len(params)))
    for p, v in zip(params, values):
        p.set_value(v)


def get_output_shape(layer):
    return layer.output_shape
