import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_audio_mixer_amd
from deep_audio_mixer_amd import ops
torch.manual_seed(0)
dev = torch.device('cuda', 0)
for (B, H, W, c) in ((2, 257, 64, 16), (2, 129, 32, 32), (2, 41, 130, 16), (1, 9, 127, 16), (3, 37, 65, 32)):
    x = torch.randn((B, H, W, c), device=dev)
    wt = torch.randn((c, c, 3, 3), device=dev) * 0.1
    wp = ops.pack_weights(wt)
    bias = torch.randn(c, device=dev)
    res = torch.randn((B, H, W, c), device=dev)
    msk = torch.randn((B, H, W, c), device=dev)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), padding=1).permute(0, 2, 3, 1)
    def err(y, want):
        d = (y.double().cpu() - want).abs()
        bad = (d > 1e-4 * want.abs().max()).nonzero()
        return '%.2e nbad %d first %s' % (d.max().item() / want.abs().max().item(), len(bad), bad[:2].tolist())
    print((B, H, W, c))
    print('  raw       ', err(ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1), ref))
    print('  bias      ', err(ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1, bias=bias), ref + bias.double().cpu()))
    print('  bias+relu ', err(ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1, bias=bias, relu_out=True), torch.relu(ref + bias.double().cpu())))
    print('  relu only ', err(ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1, relu_out=True), torch.relu(ref)))
    print('  b+res+relu', err(ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1, bias=bias, res=res, relu_out=True), torch.relu(ref + bias.double().cpu() + res.double().cpu())))
    print('  res       ', err(ops.conv2d_fwd(x, wp, c, 3, 3, 1, 1, 1, res=res), ref + res.double().cpu()))
    wpt = ops.pack_weights(wt, transpose=True)
    refd = torch.nn.functional.conv_transpose2d(x.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), padding=1).permute(0, 2, 3, 1)
    print('  dgrad     ', err(ops.conv2d_dgrad(x, wpt, c, H, W, 3, 3, 1, 1, 1), refd))
    print('  dgrad+rm  ', err(ops.conv2d_dgrad(x, wpt, c, H, W, 3, 3, 1, 1, 1, res=res, res_mask=msk), refd + (res * (msk > 0)).double().cpu()))
