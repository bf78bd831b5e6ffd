"""lasagne.init subset: GlorotUniform / Constant, plus the RNG lasagne draws from (lasagne.random).
The reference never seeds it; set_rng(np.random.RandomState(seed)) makes a run
reproducible and bit-identical to oracle.step.init_state(cfg, seed)."""
import numpy as np

_rng = np.random


def get_rng():
    return _rng


def set_rng(rng):
    global _rng
    _rng = rng


def floatX(arr):
    """lasagne.utils.floatX with floatX=float32 (experiment.5.sh:5)."""
    return np.asarray(arr, dtype=np.float32)


class GlorotUniform:
    def __init__(self, gain=1.0):
        self.gain = gain

    def __call__(self, shape):
        n1, n2 = shape[0], shape[1]
        rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        std = self.gain * np.sqrt(2.0 / ((n1 + n2) * rf))
        a = np.sqrt(3.0) * std
        return floatX(get_rng().uniform(-a, a, size=shape))


class Constant:
    def __init__(self, val=0.0):
        self.val = val

    def __call__(self, shape):
        return np.full(shape, self.val, dtype=np.float32)
