"""A lazy stand-in for the slice of Theano / Lasagne that /root/reference/pix2pix.py:__init__ touches
(pix2pix.py:87-147), so that the reference's own loss / gradient / update WIRING can be executed here.

TEST INFRASTRUCTURE ONLY.  Expressions are closures over a per-call context; `theano.function` evaluates them
eagerly on oracle/ops.py through oracle/tape.py.  Layer graphs are this package's layer objects (which is what
the reference's architecture files build when run against the drop-in vocabulary); `get_output` interprets them
with the oracle's ops, BatchNorm running statistics follow Lasagne's default_update rule,
`rmsprop` / `adam` restate lasagne.updates (Appendix A.8) on top of `grad`.

What is NOT reference code here: every op's arithmetic (oracle/ops.py), the optimiser formulas, the BN update
rule.  What IS reference code when this module is used by make_reference_step.py: which outputs feed which loss,
the targets, the alpha weighting, which parameters each loss updates, the order and contents of `updates`, the
five outputs of train_fn / loss_fn, and the layer graphs themselves.
"""
from collections import OrderedDict

import numpy as np

from oracle import ops as OPS
from oracle import tape as TP
from gan_heightmaps_amd import layers as L


# ---- lazy expressions --------------------------------------------------------------------------------
class Ctx:
    def __init__(self, env, dtype):
        self.env, self.dtype = env, dtype
        self.memo, self.pnodes, self.bn, self.gcache = {}, {}, [], {}

    def param(self, p):
        if id(p) not in self.pnodes:
            self.pnodes[id(p)] = TP.leaf(np.asarray(p.get_value(), self.dtype), p.name)
        return self.pnodes[id(p)]


class Sym:
    def __init__(self, fn, name=None):
        self.fn, self.name = fn, name

    def ev(self, ctx):
        if id(self) not in ctx.memo:
            ctx.memo[id(self)] = self.fn(ctx)
        return ctx.memo[id(self)]

    # arithmetic used by pix2pix.py:102-121
    def __add__(self, o):
        return Sym(lambda c: TP.add(self.ev(c), _node(o, c)))
    __radd__ = __add__

    def __sub__(self, o):
        return Sym(lambda c: _sub(self.ev(c), _node(o, c)))

    def __mul__(self, o):
        if isinstance(o, Sym):
            raise NotImplementedError("product of two expressions")
        return Sym(lambda c: TP.scale(self.ev(c), c.dtype(o)))
    __rmul__ = __mul__

    def mean(self):
        return Sym(lambda c: TP.scalar_loss(self.ev(c), lambda v: (v.mean(), np.full_like(v, 1.0 / v.size))))


def _node(o, c):
    return o.ev(c) if isinstance(o, Sym) else TP.leaf(np.asarray(o, c.dtype))


def _sub(a, b):
    return TP.Node(a.v - b.v, (a, b), lambda g: (g, -g))


def placeholder(name):
    return Sym(lambda c: TP.leaf(np.asarray(c.env[name], c.dtype), name), name)


def abs_(x):
    def f(c):
        a = x.ev(c)
        return TP.Node(np.abs(a.v), (a,), lambda g: (g * np.sign(a.v),))
    return Sym(f)


def squared_error(a, b):
    def f(c):
        x, t = a.ev(c), _node(b, c)
        d = x.v - t.v
        return TP.Node(d * d, (x, t), lambda g: (2 * d * g, _unbroadcast(-2 * d * g, t.v.shape)))
    return Sym(f)


def binary_crossentropy(a, b):
    def f(c):
        x, t = a.ev(c), _node(b, c)
        v = -(t.v * np.log(x.v) + (1 - t.v) * np.log(1 - x.v))
        return TP.Node(v, (x,), lambda g: (g * (-(t.v / x.v) + (1 - t.v) / (1 - x.v)),))
    return Sym(f)


def _unbroadcast(g, shape):
    if g.shape == tuple(shape):
        return g
    return np.asarray(g.sum()).reshape(shape) if int(np.prod(shape)) == 1 else g


# ---- lasagne.layers.get_output over this package's layer objects ---------------------------------------
def _act(x, nl):
    return TP.act(x, nl.kind, nl.alpha if nl.kind == 'lrelu' else None)


def get_output(layer, inputs=None, deterministic=False):
    def f(c):
        vals = {}
        for l in L.get_all_layers(layer):
            cls = type(l).__name__
            if isinstance(l, L.InputLayer):
                feed = inputs[l] if isinstance(inputs, dict) else inputs
                vals[id(l)] = _node(feed, c)
                continue
            if isinstance(l, L.MergeLayer):
                xs = [vals[id(i)] for i in l.input_layers]
                assert cls == 'ConcatLayer' and l.axis == 1
                vals[id(l)] = TP.concat(xs, 1)
                continue
            x = vals[id(l.input_layer)]
            if cls == 'DenseLayer':
                if x.v.ndim > 2:
                    x = TP.reshape(x, (x.v.shape[0], -1))
                y = _act(TP.dense(x, c.param(l.W), c.param(l.b)), l.nonlinearity)
            elif cls == 'Conv2DLayer':
                y = _act(TP.conv2d(x, c.param(l.W), c.param(l.b), l.stride[0], l.pad[0]), l.nonlinearity)
            elif cls in ('TransposedConv2DLayer', 'Deconv2DLayer'):
                y = _act(TP.deconv2d(x, c.param(l.W), c.param(l.b), l.stride[0], l.crop[0] if isinstance(l.crop, tuple) else l.crop),
                         l.nonlinearity)
            elif cls == 'BatchNormLayer':
                if deterministic:
                    y = TP.bn_infer(x, c.param(l.beta), c.param(l.gamma), np.asarray(l.mean.get_value(), c.dtype),
                                    np.asarray(l.inv_std.get_value(), c.dtype))
                else:
                    y, mu, inv = TP.bn_train(x, c.param(l.beta), c.param(l.gamma))
                    c.bn.append((l, mu, inv))
            elif cls == 'InstanceNormLayer':            # (this package's layer; no reference counterpart)
                y = TP.instance_norm(x, c.param(l.beta), c.param(l.gamma), l.epsilon)
            elif cls == 'NonlinearityLayer':
                y = _act(x, l.nonlinearity)
            elif cls == 'ReshapeLayer':
                shp = tuple(l.shape)
                y = TP.reshape(x, (-1,) + shp[1:] if shp[0] == -1 else shp)
            elif cls == 'Upscale2DLayer':
                y = TP.upscale_nearest(x, 2)
            elif cls == 'BilinearUpsample2DLayer':
                y = TP.bilinear_up2(x)
            elif cls == 'MaxPool2DLayer' or (cls == 'Pool2DLayer' and l.mode == 'max'):
                y = TP.maxpool(x, l.pool_size[0])
            elif cls == 'Pool2DLayer':
                y = TP.avgpool(x, l.pool_size[0])
            elif cls == 'DropoutLayer':
                if deterministic or l.p == 0:
                    y = x
                else:       # this build's hash-based mask (oracle/ops.py dropout_mask); env['__rng__'](layer) -> (key, step)
                    key, step = c.env['__rng__'](l)
                    y = TP.dropout(x, l.p, key, step)
            else:
                raise NotImplementedError(cls)
            vals[id(l)] = y
        return vals[id(layer)]
    return Sym(f)


# ---- theano.grad + lasagne.updates ------------------------------------------------------------------------
class State:
    """an optimiser accumulator (a theano shared variable created inside lasagne.updates)"""

    def __init__(self, value):
        self.value = value

    def get_value(self):
        return self.value

    def set_value(self, v):
        self.value = v


def _grads(loss, params, c):
    key = id(loss)
    if key not in c.gcache:
        root = loss.ev(c)
        leaves = [c.param(p) for p in params]
        TP.backward(root)
        c.gcache[key] = {id(p): (n.g.copy() if n.g is not None else np.zeros_like(n.v)) for p, n in zip(params, leaves)}
    return c.gcache[key]


def _lr(learning_rate):
    return float(learning_rate.get_value()) if hasattr(learning_rate, 'get_value') else float(learning_rate)


def rmsprop(loss_or_grads, params, learning_rate=1.0, rho=0.9, epsilon=1e-6):
    updates = OrderedDict()
    for p in params:
        acc = State(np.zeros(p.shape, np.float64))

        def both(c, p=p, acc=acc):
            k = ('rms', id(p), id(loss_or_grads))
            if k not in c.memo:
                g = _grads(loss_or_grads, params, c)[id(p)]
                c.memo[k] = OPS.rmsprop_step(c.param(p).v, g, np.asarray(acc.value, c.dtype), _lr(learning_rate), rho, epsilon)
            return c.memo[k]
        updates[acc] = Sym(lambda c, both=both: TP.leaf(both(c)[1]))
        updates[p] = Sym(lambda c, both=both: TP.leaf(both(c)[0]))
    return updates


def adam(loss_or_grads, params, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
    updates = OrderedDict()
    t_prev = State(0)
    for p in params:
        m, v = State(np.zeros(p.shape, np.float64)), State(np.zeros(p.shape, np.float64))

        def trio(c, p=p, m=m, v=v):
            k = ('adam', id(p), id(loss_or_grads))
            if k not in c.memo:
                g = _grads(loss_or_grads, params, c)[id(p)]
                c.memo[k] = OPS.adam_step(c.param(p).v, g, np.asarray(m.value, c.dtype), np.asarray(v.value, c.dtype),
                                          t_prev.value, _lr(learning_rate), beta1, beta2, epsilon)
            return c.memo[k]
        updates[m] = Sym(lambda c, trio=trio: TP.leaf(trio(c)[1]))
        updates[v] = Sym(lambda c, trio=trio: TP.leaf(trio(c)[2]))
        updates[p] = Sym(lambda c, trio=trio: TP.leaf(trio(c)[0]))
    updates[t_prev] = Sym(lambda c: TP.leaf(np.asarray(t_prev.value + 1)))
    return updates


# ---- theano.function ------------------------------------------------------------------------------------------
DTYPE = [np.float64]


def function(inputs, outputs, updates=None, **unused):
    names = [i.name for i in inputs]
    single = not isinstance(outputs, (list, tuple))
    outs = [outputs] if single else list(outputs)

    def call(*arrays):
        c = Ctx(dict(zip(names, arrays)), DTYPE[0])
        vals = [np.asarray(o.ev(c).v) for o in outs]
        new = [(k, np.asarray(s.ev(c).v)) for k, s in (updates or {}).items()]      # all from the OLD values
        for k, v in new:
            if isinstance(k, State):
                k.set_value(v if v.ndim else v.item())
            else:
                k.set_value(np.asarray(v, np.float32))
        seen = set()
        for l, mu, inv in c.bn:                     # lasagne BatchNormLayer default_updates, alpha = 0.1
            if id(l) in seen:
                raise NotImplementedError("one BatchNormLayer evaluated twice in a function (order-dependent update)")
            seen.add(id(l))
            a = l.alpha
            l.mean.set_value((1 - a) * l.mean.get_value().astype(np.float64) + a * np.asarray(mu, np.float64).ravel())
            l.inv_std.set_value((1 - a) * l.inv_std.get_value().astype(np.float64) + a * np.asarray(inv, np.float64).ravel())
        return vals[0] if single else vals
    return call
