"""MixingModelScalar1s -- drop-in for the reference's models/model_scalar_1s.py (:207-275; ConvBlock2d :151-190).
``MixingModelScalar1s()`` reproduces the reference (4 stems, 1025x87 input, flattened_dim 10290, 908,400
parameters, same state_dict keys); ``n_stems`` / ``input_shape`` are keyword-only extensions."""
import torch

from ..layers import ConvBlock2d
from ._scalar import ScalarMixingNet

__all__ = ['ConvBlock2d', 'MixingModelScalar1s', 'dB_to_amplitude', 'amplitude_to_dB']


def dB_to_amplitude(x: torch.Tensor):
    """models/model_scalar_1s.py:193-197: 10 ** (0.5 * x) (sic)."""
    return torch.pow(10.0, 0.5 * x)


def amplitude_to_dB(x: torch.Tensor):
    """models/model_scalar_1s.py:200-204."""
    return 20 * torch.log10(x)


class MixingModelScalar1s(ScalarMixingNet):
    first_dilation = 1

    def __init__(self, *, n_stems=4, input_shape=(1025, 87)):
        super().__init__(n_stems, input_shape)
