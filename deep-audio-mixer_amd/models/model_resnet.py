"""ResNet18 gain predictor -- drop-in for the reference's models/model_resnet.py (ResNet :59-126,
ResNet18 :129-130) on MI355X.  Same constructor call (``ResNet18()``), same forward contract
``forward(x[B,S,F,T]) -> (masked[B,F,T], (g_1..g_S) each [B,1])``, same state_dict keys, so reference
checkpoints load.  Two keyword-only extensions: ``n_stems`` (reference: 4, hard-wired at
:64,75-85) and ``input_shape=(F, T)`` from which the head width is derived (reference: flattened_dim = 231
at :73, i.e. a 1025x216 input).
"""
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401  (kept for notebook parity: the reference module exposes F)

from ..layers import BasicBlock, ConvBnReluFn, ConvSpec, FoldedConvBn, MixingNet, inference_mode
from ..ops import conv_out_size

__all__ = ['BasicBlock', 'ResNet', 'ResNet18']


def _trunk_hw(f, t, strides):
    for s in strides:
        if s != 1:
            f, t = conv_out_size(f, 3, s, 1), conv_out_size(t, 3, s, 1)
    return f, t


class ResNet(MixingNet):
    def __init__(self, block, num_blocks, *, n_stems=4, input_shape=(1025, 216)):
        super().__init__()
        self.in_planes = 16
        self.conv1 = nn.Conv2d(n_stems, 16, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(16)
        self._stem = ConvSpec(n_stems, 16, 3, 1, 1, in_nchw=True)
        widths, strides = (16, 32, 64, 96, 128, 256), (1, 2, 2, 2, 2, 2)
        for i, (w, s, n) in enumerate(zip(widths, strides, num_blocks), start=1):
            setattr(self, 'layer%d' % i, self._make_layer(block, w, n, stride=s))
        fh, ft = _trunk_hw(input_shape[0], input_shape[1], strides)
        flattened_dim = fh * ft      # 231 for the reference's 1025 x 216 input
        self._init_heads(256 * block.expansion, n_stems, flattened_dim)

    def _make_layer(self, block, planes, num_blocks, stride):
        layers = []
        for s in [stride] + [1] * (num_blocks - 1):
            layers.append(block(self.in_planes, planes, s))
            self.in_planes = planes * block.expansion
        return nn.Sequential(*layers)

    def conv_pairs(self):
        pairs = [(self._stem, self.conv1.weight, False)]          # the input needs no gradient
        for i in range(1, 7):
            for blk in getattr(self, 'layer%d' % i):
                pairs += [(blk.spec1, blk.conv1.weight, True), (blk.spec2, blk.conv2.weight, True)]
                if blk.spec_sc is not None:
                    pairs.append((blk.spec_sc, blk.shortcut[0].weight, True))
        return pairs

    DDP_BOUNDARY = 4      # layer5, layer6 and the heads hold 85 % of the parameters and come first in backward

    def trunk(self, x, tap=None):
        stem = getattr(self, '_stem_fold', None)
        if stem is None:
            stem = self._stem_fold = FoldedConvBn(self._stem, self.conv1, self.bn1)
        if inference_mode(self) and stem.foldable():
            out = stem.fwd(x)
        else:
            out = ConvBnReluFn.apply(x, self.conv1.weight, None, self.bn1.weight, self.bn1.bias, self._stem, self.bn1,
                                     self.training)
        for i in range(1, 7):
            out = getattr(self, 'layer%d' % i)(out)
            if tap is not None and i == self.DDP_BOUNDARY:
                tap.append(out)
        return out

    def ddp_late_parameters(self):
        late = []
        for i in range(self.DDP_BOUNDARY + 1, 7):
            late += list(getattr(self, 'layer%d' % i).parameters())
        return late + list(self._heads.parameters())


def ResNet18(**kwargs):
    return ResNet(BasicBlock, [2, 2, 2, 2, 2, 2], **kwargs)
