"""Tiny reverse-mode tape over oracle/ops.py (TEST INFRASTRUCTURE ONLY).

Stands in for ``theano.grad`` as used by lasagne.updates inside
/root/reference/pix2pix.py:131-135: the nets in oracle/nets.py are written as plain forward
functions and the gradients come from the per-op vjps in oracle/ops.py.
"""
import numpy as np

from . import ops


class Node:
    __slots__ = ("v", "g", "parents", "vjp", "name")

    def __init__(self, v, parents=(), vjp=None, name=None):
        self.v = v
        self.g = None
        self.parents = parents
        self.vjp = vjp
        self.name = name

    @property
    def shape(self):
        return self.v.shape


def leaf(v, name=None):
    return Node(v, name=name)


def _topo(root, stop):
    order, seen = [], set()
    stack = [(root, False)]
    while stack:
        n, done = stack.pop()
        if done:
            order.append(n)
            continue
        if id(n) in seen:
            continue
        seen.add(id(n))
        stack.append((n, True))
        if id(n) in stop:
            continue
        for p in n.parents:
            if id(p) not in seen:
                stack.append((p, False))
    return order


def backward(root, seed=None, stop_at=()):
    """Accumulate d root / d node into node.g for every node reachable from root.
    Nodes in ``stop_at`` receive their gradient but do not propagate further."""
    stop = {id(n) for n in stop_at}
    order = _topo(root, stop)
    for n in order:
        n.g = None
    root.g = np.ones_like(root.v) if seed is None else seed
    for n in reversed(order):
        if n.vjp is None or n.g is None or id(n) in stop:
            continue
        grads = n.vjp(n.g)
        for p, gp in zip(n.parents, grads):
            if gp is None:
                continue
            p.g = gp if p.g is None else p.g + gp
    return order


# ---- op wrappers ---------------------------------------------------------------------


def conv2d(x, W, b, stride, pad):
    y = ops.conv2d_fwd(x.v, W.v, b.v, stride, pad)
    return Node(y, (x, W, b), lambda g: ops.conv2d_vjp(x.v, W.v, g, stride, pad))


def deconv2d(x, W, b, stride, crop=0):
    y = ops.deconv2d_fwd(x.v, W.v, b.v, stride, crop)
    return Node(y, (x, W, b), lambda g: ops.deconv2d_vjp(x.v, W.v, g, stride, crop))


def dense(x, W, b):
    y = ops.dense_fwd(x.v, W.v, b.v)
    return Node(y, (x, W, b), lambda g: ops.dense_vjp(x.v, W.v, g))


def bn_train(x, beta, gamma):
    """-> (node, mu, inv)"""
    y, mu, inv = ops.bn_train_fwd(x.v, beta.v, gamma.v)
    return Node(y, (x, beta, gamma), lambda g: ops.bn_train_vjp(x.v, gamma.v, mu, inv, g)), mu, inv


def instance_norm(x, beta, gamma, eps=ops.BN_EPS):
    y, mu, inv = ops.in_fwd(x.v, beta.v, gamma.v, eps)
    return Node(y, (x, beta, gamma), lambda g: ops.in_vjp(x.v, gamma.v, mu, inv, g))


def bn_infer(x, beta, gamma, mean, inv_std):
    y = ops.bn_infer_fwd(x.v, beta.v, gamma.v, mean, inv_std)
    return Node(y, (x,), None)


def lrelu(x, a):
    return Node(ops.lrelu_fwd(x.v, a), (x,), lambda g: (ops.lrelu_vjp(x.v, a, g),))


def relu(x):
    return Node(ops.relu_fwd(x.v), (x,), lambda g: (ops.relu_vjp(x.v, g),))


def sigmoid(x):
    y = ops.sigmoid_fwd(x.v)
    return Node(y, (x,), lambda g: (ops.sigmoid_vjp_from_out(y, g),))


def tanh(x):
    y = ops.tanh_fwd(x.v)
    return Node(y, (x,), lambda g: (ops.tanh_vjp_from_out(y, g),))


def act(x, kind, a=None):
    if kind == 'linear':
        return x
    if kind == 'relu':
        return relu(x)
    if kind == 'lrelu':
        return lrelu(x, a)
    if kind == 'sigmoid':
        return sigmoid(x)
    if kind == 'tanh':
        return tanh(x)
    raise ValueError(kind)


def upscale_nearest(x, f=2):
    return Node(ops.upscale_nearest_fwd(x.v, f), (x,), lambda g: (ops.upscale_nearest_vjp(g, f),))


def bilinear_up2(x):
    return Node(ops.bilinear_up2_fwd(x.v), (x,), lambda g: (ops.bilinear_up2_vjp(g),))


def maxpool(x, p=2):
    y = ops.maxpool_fwd(x.v, p)
    return Node(y, (x,), lambda g: (ops.maxpool_vjp(x.v, y, g, p),))


def avgpool(x, p):
    return Node(ops.avgpool_fwd(x.v, p), (x,), lambda g: (ops.avgpool_vjp(x.v.shape, g, p),))


def reshape(x, shape):
    return Node(x.v.reshape(shape), (x,), lambda g: (g.reshape(x.v.shape),))


def concat(xs, axis=1):
    sizes = [x.v.shape[axis] for x in xs]
    cuts = np.cumsum(sizes)[:-1]
    return Node(np.concatenate([x.v for x in xs], axis=axis), tuple(xs),
                lambda g: tuple(np.split(g, cuts, axis=axis)))


def scalar_loss(x, fn):
    """fn(x.v) -> (loss_scalar, dloss/dx)."""
    loss, dl = fn(x.v)
    return Node(np.asarray(loss), (x,), lambda g: (g * dl,))


def add(a, b):
    return Node(a.v + b.v, (a, b), lambda g: (g, g))


def scale(a, s):
    return Node(a.v * s, (a,), lambda g: (g * s,))


def dropout(x, p, key, step):
    m = ops.dropout_mask(x.v.shape, p, key, step).astype(x.v.dtype) / x.v.dtype.type(1.0 - np.float32(p))
    return Node(x.v * m, (x,), lambda g: (g * m,))

