"""Shared body of MixingModelScalar1s / MixingModelScalar2s."""
from ..layers import ConvBlock2d, MixingNet
from ..ops import conv_out_size

BLOCKS = ((16, 3, 0.2), (32, 5, 0.2), (48, 5, 0.2), (64, 7, 0.2), (128, 9, 0.3))   # (channels, kernel, dropout)


class ScalarMixingNet(MixingNet):
    first_dilation = 1

    def __init__(self, n_stems, input_shape):
        super().__init__()
        cin, (f, t) = n_stems, input_shape
        for i, (w, k, p) in enumerate(BLOCKS, start=1):
            s, d = (2, self.first_dilation) if i == 1 else (1, 1)
            setattr(self, 'conv_b%d' % i, ConvBlock2d(cin, w, k, dropout_p=p, stride=s, dilation=d, in_nchw=(i == 1)))
            f, t = conv_out_size(f, k, s, 0, d), conv_out_size(t, k, s, 0, d)
            cin = w
        if f <= 0 or t <= 0:
            raise ValueError('input_shape %r is too small for the five valid convolutions' % (input_shape,))
        self._init_heads(128, n_stems, f * t)

    def conv_pairs(self):
        return [(getattr(self, 'conv_b%d' % i).spec, getattr(self, 'conv_b%d' % i).conv.weight, i > 1) for i in range(1, 6)]

    DDP_BOUNDARY = 3      # conv_b4 (7x7), conv_b5 (9x9) and the heads hold 88 % of the parameters

    def trunk(self, x, tap=None):
        out = x
        for i in range(1, 6):
            out = getattr(self, 'conv_b%d' % i)(out)
            if tap is not None and i == self.DDP_BOUNDARY:
                tap.append(out)
        return out

    def ddp_late_parameters(self):
        late = []
        for i in range(self.DDP_BOUNDARY + 1, 6):
            late += list(getattr(self, 'conv_b%d' % i).parameters())
        return late + list(self._heads.parameters())
