"""Names of lasagne.nonlinearities the reference passes around (experiments.py:5, architectures/*.py).
Each is a small descriptor the engine lowers to a libghm activation code; calling one on a numpy array
evaluates it on the host (used only for tiny host-side checks)."""
import numpy as np


class Nonlinearity:
    def __init__(self, kind, alpha=0.0):
        self.kind = kind
        self.alpha = float(alpha)

    def __call__(self, x):
        x = np.asarray(x)
        if self.kind == 'linear':
            return x
        if self.kind == 'relu':
            return np.maximum(x, 0)
        if self.kind == 'lrelu':
            return np.where(x > 0, x, self.alpha * x)
        if self.kind == 'sigmoid':
            return 1.0 / (1.0 + np.exp(-x))
        if self.kind == 'tanh':
            return np.tanh(x)
        raise ValueError(self.kind)

    def __repr__(self):
        return "<nonlinearity %s%s>" % (self.kind, "(%g)" % self.alpha if self.kind == 'lrelu' else "")

    def __eq__(self, other):
        return isinstance(other, Nonlinearity) and (self.kind, self.alpha) == (other.kind, other.alpha)

    def __hash__(self):
        return hash((self.kind, self.alpha))


class LeakyRectify(Nonlinearity):
    """lasagne.nonlinearities.LeakyRectify(leakiness) (architectures/dcgan.py:24,45 use 0.2)."""

    def __init__(self, leakiness=0.01):
        Nonlinearity.__init__(self, 'lrelu', leakiness)


linear = identity = Nonlinearity('linear')
rectify = Nonlinearity('relu')
sigmoid = Nonlinearity('sigmoid')
tanh = Nonlinearity('tanh')
leaky_rectify = LeakyRectify(0.01)          # the predefined instance used throughout architectures/p2p.py
very_leaky_rectify = LeakyRectify(1. / 3)


def as_nonlinearity(n):
    if n is None:
        return linear
    if isinstance(n, Nonlinearity):
        return n
    if isinstance(n, str):                    # reference default nonlinearity='sigmoid' is a string (dcgan.py:35)
        return {'linear': linear, 'sigmoid': sigmoid, 'tanh': tanh, 'rectify': rectify, 'relu': rectify}[n]
    raise TypeError("unsupported nonlinearity %r" % (n,))
