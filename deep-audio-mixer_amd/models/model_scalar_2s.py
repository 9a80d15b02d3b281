"""MixingModelScalar2s -- drop-in for the reference's models/model_scalar_2s.py (:64-132): as the 1s model but the
first block is dilated (dilation=2, :68) and the reference input is 1025x173 (flattened_dim 30807, :77)."""
from ..layers import ConvBlock2d
from ._scalar import ScalarMixingNet
from .model_scalar_1s import amplitude_to_dB, dB_to_amplitude

__all__ = ['ConvBlock2d', 'MixingModelScalar2s', 'dB_to_amplitude', 'amplitude_to_dB']


class MixingModelScalar2s(ScalarMixingNet):
    first_dilation = 2

    def __init__(self, *, n_stems=4, input_shape=(1025, 173)):
        super().__init__(n_stems, input_shape)
