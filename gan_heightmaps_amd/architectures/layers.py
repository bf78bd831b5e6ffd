"""Custom layer of the reference (/root/reference/architectures/layers.py:13-26)."""
from ..layers import Layer


class BilinearUpsample2DLayer(Layer):
    """x2 upsampling with the semantics of theano.tensor.nnet.abstract_conv.bilinear_upsampling(ratio=factor):
    per axis out[2m] = x[m], out[2m+1] = (x[m] + x[min(m+1, n-1)]) / 2.
    Lowered to ghm_upsample_bilinear2_{fwd,bwd}."""

    def __init__(self, incoming, factor, **kwargs):
        Layer.__init__(self, incoming, **kwargs)
        if int(factor) != 2:
            raise NotImplementedError("BilinearUpsample2DLayer: only factor=2 has a kernel")
        self.factor = int(factor)

    def get_output_shape_for(self, input_shape):
        return tuple(input_shape[0:2]) + (input_shape[2] * self.factor, input_shape[3] * self.factor)
