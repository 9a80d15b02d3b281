"""Oracle: the three gain-predicting CNNs on PyTorch-CPU (test infrastructure).

Restates the reference model definitions with the stem count and the head
width made constructor arguments; with ``n_stems=4`` and the
reference's input shape the parameter names, shapes and arithmetic are those of

  models/model_resnet.py:6-28     BasicBlock
  models/model_resnet.py:59-126   ResNet / ResNet18 (:129-130)
  models/model_scalar_1s.py:151-190, 207-275   ConvBlock2d / MixingModelScalar1s
  models/model_scalar_2s.py:9-47, 64-132       ConvBlock2d / MixingModelScalar2s

Everything here runs through torch.nn.functional on the CPU.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def conv_out(n, k, s=1, d=1, p=0):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


class _Heads(nn.Module):
    """Shared tail: per stem conv1x1(C->1)+ReLU -> flatten -> Linear(flat->1);
    masked = sum_s g_s * x[:, s]  (models/model_resnet.py:108-126)."""

    def _make_heads(self, channels, n_stems, flattened_dim):
        self.n_stems = n_stems
        for s in range(1, n_stems + 1):
            setattr(self, 'conv_head%d' % s, nn.Conv2d(channels, 1, kernel_size=(1, 1)))
            setattr(self, 'fc_head%d' % s, nn.Linear(flattened_dim, 1))

    def _run_heads(self, x, trunk):
        b = x.size(0)
        gains = []
        for s in range(1, self.n_stems + 1):
            h = F.relu(getattr(self, 'conv_head%d' % s)(trunk))
            gains.append(getattr(self, 'fc_head%d' % s)(h.view(b, -1)))
        masked = torch.zeros_like(x[:, 0])
        for s, g in enumerate(gains):
            masked = masked + g.unsqueeze(2) * x[:, s]
        return masked, tuple(gains)


class RefBasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.shortcut = nn.Sequential()
        if stride != 1 or cin != cout:
            self.shortcut = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                          nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + self.shortcut(x))


RESNET_WIDTHS = (16, 32, 64, 96, 128, 256)      # models/model_resnet.py:66-71
RESNET_STRIDES = (1, 2, 2, 2, 2, 2)


def resnet_trunk_hw(f, t):
    for s in RESNET_STRIDES:
        if s == 2:
            f, t = conv_out(f, 3, 2, 1, 1), conv_out(t, 3, 2, 1, 1)
    return f, t


class RefResNet18(_Heads):
    """models/model_resnet.py:59-130; reference defaults n_stems=4, input 1025x216 -> flat 231."""

    def __init__(self, n_stems=4, input_shape=(1025, 216)):
        super().__init__()
        self.conv1 = nn.Conv2d(n_stems, 16, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(16)
        cin = 16
        for i, (w, s) in enumerate(zip(RESNET_WIDTHS, RESNET_STRIDES), start=1):
            setattr(self, 'layer%d' % i, nn.Sequential(RefBasicBlock(cin, w, s), RefBasicBlock(w, w, 1)))
            cin = w
        fh, ft = resnet_trunk_hw(*input_shape)
        self._make_heads(256, n_stems, fh * ft)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        for i in range(1, 7):
            y = getattr(self, 'layer%d' % i)(y)
        return self._run_heads(x, y)


class RefConvBlock2d(nn.Module):
    """models/model_scalar_1s.py:151-190: valid conv(+bias) -> BN(eps 1e-3, momentum .9)
    -> ReLU -> Dropout only while training."""

    def __init__(self, cin, cout, k, stride=1, dilation=1, dropout_p=-1.0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, 0, dilation)
        self.batch_norm = nn.BatchNorm2d(cout, momentum=0.90, eps=0.001)
        self.dropout_p = dropout_p

    def forward(self, x):
        y = F.relu(self.batch_norm(self.conv(x)))
        if self.training and self.dropout_p != -1:
            y = F.dropout(y, self.dropout_p, True)
        return y


SCALAR_BLOCKS = ((16, 3, 0.2), (32, 5, 0.2), (48, 5, 0.2), (64, 7, 0.2), (128, 9, 0.3))


def scalar_trunk_hw(f, t, first_dilation):
    for i, (_, k, _) in enumerate(SCALAR_BLOCKS):
        s, d = (2, first_dilation) if i == 0 else (1, 1)
        f, t = conv_out(f, k, s, d), conv_out(t, k, s, d)
    return f, t


class _RefScalar(_Heads):
    first_dilation = 1

    def __init__(self, n_stems, input_shape):
        super().__init__()
        cin = n_stems
        for i, (w, k, p) in enumerate(SCALAR_BLOCKS, start=1):
            s, d = (2, self.first_dilation) if i == 1 else (1, 1)
            setattr(self, 'conv_b%d' % i, RefConvBlock2d(cin, w, k, s, d, p))
            cin = w
        fh, ft = scalar_trunk_hw(*input_shape, self.first_dilation)
        self._make_heads(128, n_stems, fh * ft)

    def forward(self, x):
        y = x
        for i in range(1, 6):
            y = getattr(self, 'conv_b%d' % i)(y)
        return self._run_heads(x, y)


class RefMixingModelScalar1s(_RefScalar):
    """models/model_scalar_1s.py:207-275; reference: 4 stems, 1025x87 -> flat 10290."""
    first_dilation = 1

    def __init__(self, n_stems=4, input_shape=(1025, 87)):
        super().__init__(n_stems, input_shape)


class RefMixingModelScalar2s(_RefScalar):
    """models/model_scalar_2s.py:64-132; reference: 4 stems, 1025x173 -> flat 30807."""
    first_dilation = 2

    def __init__(self, n_stems=4, input_shape=(1025, 173)):
        super().__init__(n_stems, input_shape)


def closed_form_fill(model: nn.Module, seed: int = 0):
    """Deterministic parameter fill shared by the golden generator and the tests, so that no multi-MB
    weight fixture is needed.  Values are an integer hash of (tensor index, flat
    index) mapped to U(-0.5, 0.5) -- white-noise-like, portable (numpy integer arithmetic only).  Conv /
    linear weights are scaled like Kaiming-uniform, BN gammas are 1 +- 0.1, biases / betas +- 0.05.
    (A smooth closed form such as sin(a*i) gives near-degenerate filters: with it the reference's OWN float32
    and float64 training-mode outputs differ by 2e-2; with this fill they agree to 2e-5.)"""
    import numpy as np
    u32 = np.uint64(0xffffffff)
    with torch.no_grad():
        for k, (name, p) in enumerate(model.named_parameters()):
            i = np.arange(p.numel(), dtype=np.uint64)
            h = (i * np.uint64(2654435761) + np.uint64(k * 40503 + seed * 7919 + 12345)) & u32
            h = ((h ^ (h >> np.uint64(15))) * np.uint64(2246822519)) & u32
            h = ((h ^ (h >> np.uint64(13))) * np.uint64(3266489917)) & u32
            h = h ^ (h >> np.uint64(16))
            u = h.astype(np.float64) / 4294967296.0 - 0.5
            if p.dim() >= 2:
                v = u * 2.0 * (3.0 / p[0].numel()) ** 0.5
            elif name.endswith('weight'):          # BN gamma
                v = 1.0 + 0.2 * u
            else:                                   # biases / BN beta
                v = 0.1 * u
            p.copy_(torch.from_numpy(v).reshape(p.shape).to(p.dtype))
    return model


def train_step_ref(model, optimizer, x, gt):
    """One body of model_trainer.py:25-44 (zero_grad, fwd, MSE, backward, step)."""
    optimizer.zero_grad()
    masked, gains = model(x)
    loss = F.mse_loss(masked, gt)
    loss.backward()
    optimizer.step()
    return loss.detach(), gains


# ---- ReLU decisions given from outside (per-block gradient parity, tests/test_blocks_gpu.py) --------------------------------
# A float32 implementation and this float64 restatement disagree on relu'(v) wherever |v| is below the float32 rounding
# error of v (a handful of the ~17 M activations of a full-size layer); ONE such element moves a weight gradient by ~1e-3 of
# its norm.  For a deterministic gradient comparison the restatement therefore takes the 0/1 decisions as an argument -- the
# test passes the device path's own decisions and asserts separately that they differ from sign(v) only where |v| is at
# rounding level (so a wrong mask still fails).  With mask=None these functions are the plain reference arithmetic.
def masked_relu(v, mask=None):
    """relu(v) with the decisions `mask` (bool, same shape): v * mask.  Returns (result, v detached)."""
    if mask is None:
        mask = v.detach() > 0
    return v * mask.to(v.dtype), v.detach()


def stem_forward_masked(conv, bn, x, mask=None):
    """models/model_resnet.py:97 -- relu(bn1(conv1(x))) -> (a, pre-activation)."""
    return masked_relu(bn(conv(x)), mask)


def block_forward_masked(block: RefBasicBlock, x, m1=None, m2=None):
    """models/model_resnet.py:23-28 -> (out, pre-activation of the inner ReLU, pre-activation of the outer ReLU)."""
    a1, v1 = masked_relu(block.bn1(block.conv1(x)), m1)
    out, v2 = masked_relu(block.bn2(block.conv2(a1)) + block.shortcut(x), m2)
    return out, v1, v2
