"""lasagne.updates / theano.shared subset used by experiments.py:116-117 and pix2pix.py:30.
``rmsprop`` / ``adam`` are passed to Pix2Pix as ``opt``; calling one with hyper-parameters returns the
specification the engine turns into a flat multi-tensor optimiser kernel (ghm_rmsprop / ghm_adam)."""
import numpy as np


class SharedScalar:
    """theano.shared(floatX(v)) for the learning rate: get_value()/set_value() (pix2pix.py:259).
    The engine mirrors the value into a device scalar read by the optimiser kernels, so a captured
    HIP graph sees set_value() on the next step."""

    def __init__(self, value):
        self._v = np.float32(value)
        self._listeners = []

    def get_value(self):
        return self._v

    def set_value(self, v):
        self._v = np.float32(v)
        for fn in self._listeners:
            fn(self._v)

    def __float__(self):
        return float(self._v)


def shared(value, name=None):
    return SharedScalar(value)


class OptimizerSpec:
    def __init__(self, kind, learning_rate, **hp):
        self.kind = kind
        self.learning_rate = learning_rate
        self.hp = hp


def rmsprop(learning_rate=1.0, rho=0.9, epsilon=1e-6):
    """lasagne.updates.rmsprop defaults."""
    return OptimizerSpec('rmsprop', learning_rate, rho=rho, epsilon=epsilon)


def adam(learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """lasagne.updates.adam defaults."""
    return OptimizerSpec('adam', learning_rate, beta1=beta1, beta2=beta2, epsilon=epsilon)
