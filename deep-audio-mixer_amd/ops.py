"""Thin Python wrappers over the C ABI (include/dam_hip.h): shape checks, output allocation through
torch's caching allocator, launch on the current torch stream.  No arithmetic happens here and
there is no CPU fallback."""
import ctypes

import os

import torch

from . import _lib


def _f32c(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError('%s must be a contiguous float32 tensor' % name)
    return t


def conv_out_size(n, k, stride=1, pad=0, dil=1):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


# ----------------------------------------------------------------------------- convolution
def pack_weights(w, transpose=False, out=None):
    """[O,I,KH,KW] -> packed image for the forward (transpose=False) or dgrad operator."""
    _lib.require_cuda(w)
    w = _f32c(w.detach(), 'weight')
    o, i, kh, kw = w.shape
    n_out, k_in = (i, o) if transpose else (o, i)
    n = _lib.lib().dam_conv_packed_weight_count(n_out, k_in, kh, kw)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.check(_lib.lib().dam_conv_pack_weights_f32(_lib.ptr(w), o, i, kh, kw, 1 if transpose else 0,
                                                    _lib.ptr(out), _lib.stream()), 'dam_conv_pack_weights_f32')
    return out


def pack_weights_multi(desc, n_tensors, max_total):
    """desc: CUDA int64 tensor [n,8] = {src ptr, dst ptr, O, I, KH, KW, transpose, count}: all packs in one launch."""
    _lib.check(_lib.lib().dam_conv_pack_weights_multi_f32(_lib.ptr(desc), n_tensors, max_total, _lib.stream()),
               'dam_conv_pack_weights_multi_f32')


_counters = {}
# "The last workgroup finalizes" (csrc/dam_bn_fin.h) is implemented, tested and OFF: measured on the ResNet18 step it is
# slower than the separate finalize launches it replaces (5.83 -> 6.09 ms per step; the strip convolution 66 -> 72 us per
# launch): the last workgroup's returning atomic + its fetch of the records that the sc1 stores pushed out of L2 cost more
# than a 1.5-2 us kernel boundary.  Kernel boundaries are the cheap synchronisation on this part.
INKERNEL_FINALIZE = False


# Finalize inside the elementwise consumer (dam_bn_finalize_apply_f32; the backward entry points do the same internally):
# DAM_BN_FUSED_FIN=0 keeps the separate finalize launches (A/B switch; the library reads the same variable)
FUSED_FINALIZE = os.environ.get('DAM_BN_FUSED_FIN', '1') != '0'


# strided 3x3 data gradients of the thin stages as one launch (dam_dgrad_s2_3x3_f32); DAM_NO_DGRAD_S2=1: the parity-class launches (A/B)
DGRAD_S2 = not os.environ.get('DAM_NO_DGRAD_S2')
dgrad_s2_launches = 0        # launches dam_dgrad_s2_3x3_f32 accepted (tests check that a shape took the one-launch form)
# shortcut data gradient riding in conv1's single-tap class launch (dam_conv1x1_pair_f32); DAM_NO_PAIR_1X1=1: two launches (A/B)
PAIR_1X1 = not os.environ.get('DAM_NO_PAIR_1X1')


# a down-sampling block's conv1 + shortcut convolution + both statistics passes as one launch (dam_conv_s2_pair_fwd_f32);
# DAM_NO_CONV_S2_PAIR=1: the separate launches (A/B)
CONV_S2_PAIR = not os.environ.get('DAM_NO_CONV_S2_PAIR')


# BatchNorm-backward sums from the data-gradient epilogue (include/dam_hip.h: dam_bn_bwd_sums); DAM_NO_DGRAD_SUMS=1 keeps the
# separate pass over dy and x (A/B switch)
DGRAD_BN_SUMS = not os.environ.get('DAM_NO_DGRAD_SUMS')


def arrival_counter(device):
    """The zero-initialised device word of the "last workgroup finalizes" hand-off (include/dam_hip.h: dam_bn_fin): one per
    device -- every launch that uses it returns it to zero and all of them are ordered on the current stream.  None while
    the hand-off is switched off (two-launch form)."""
    if not INKERNEL_FINALIZE:
        return None
    key = (device.type, device.index)
    if key not in _counters:
        _counters[key] = torch.zeros(4, dtype=torch.int32, device=device)
    return _counters[key]


def _bn_fin_struct(bn, out4, device):
    """bn = (gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps); out4: [4, C] result rows."""
    gamma, beta, rm, rv, nbt, mom, eps = bn
    return _lib.BnFin(_lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(rm), _lib.ptr(rv), _lib.ptr(nbt), float(mom), float(eps),
                      _lib.ptr(out4[0]), _lib.ptr(out4[1]), _lib.ptr(out4[2]), _lib.ptr(out4[3]),
                      _lib.ptr(arrival_counter(device)))


def _tapgrid(x, B, H, W, C, in_nchw, wp, k_chunks, n_out, bias, in_scale, in_shift, relu_in, y, OHt, OWt, Ho, Wo,
             out_stride, oo_h, oo_w, in_stride, nA, nB, off_h, step_h, off_w, step_w, wt_base, wt_sa, wt_sb,
             res=None, res_mask=None, bn_partial=None, bn_fin=None, relu_out=False, batch=None, bn_bwd=None):
    parts = ctypes.c_int(0)
    # split-K scratch: the host code only splits when no tile shape gives 400 workgroups, i.e. for outputs below
    # 400 * 64 px * 64 ch = 1.64 M floats, and then at most 8 ways (dam_conv.hip) -- never more than 13.1 M floats
    ws = _workspace(y.device, min(8 * y.numel(), 8 * 400 * 64 * 64))
    st = _lib.lib().dam_conv2d_tapgrid_f32(
        _lib.ptr(x), B, H, W, C, 1 if in_nchw else 0, _lib.ptr(wp), k_chunks, n_out, _lib.ptr(bias),
        _lib.ptr(in_scale), _lib.ptr(in_shift), 1 if relu_in else 0, 1 if relu_out else 0, _lib.ptr(y), OHt, OWt, Ho, Wo,
        out_stride,
        oo_h, oo_w, in_stride, nA, nB, off_h, step_h, off_w, step_w, wt_base, wt_sa, wt_sb, _lib.ptr(res),
        _lib.ptr(res_mask), _lib.ptr(bn_partial), ctypes.byref(parts) if bn_partial is not None else None,
        ctypes.byref(bn_fin) if bn_fin is not None else None, ctypes.byref(bn_bwd) if bn_bwd is not None else None,
        _lib.ptr(ws), ws.numel(),
        ctypes.addressof(batch) if batch is not None else None, _lib.stream())
    _lib.check(st, 'dam_conv2d_tapgrid_f32')
    return parts.value


def conv2d_fwd(x, wp, n_out, kh, kw, stride=1, pad=0, dil=1, bias=None, in_scale=None, in_shift=None,
               relu_in=False, in_nchw=False, bn_partial=None, bn=None, res=None, relu_out=False, finalize=True):
    """x: NHWC [B,H,W,C] (C % 16 == 0), or NCHW [B,C,H,W] with C <= 16 if in_nchw.  Returns NHWC
    [B,Ho,Wo,n_out] with n_out rounded up to a multiple of 16 (extra channels are zero).
    res (same shape as the result) is added, relu_out applies max(., 0) last: with an eval-mode BatchNorm folded into
    the weights and `bias`, one launch is relu(bn(conv(x)) [+ shortcut])."""
    _lib.require_cuda(x, wp)
    _f32c(x, 'x'), _f32c(bias, 'bias'), _f32c(in_scale, 'in_scale'), _f32c(in_shift, 'in_shift'), _f32c(res, 'res')
    if in_nchw:
        B, C, H, W = x.shape
        k_chunks = 1
    else:
        B, H, W, C = x.shape
        k_chunks = C // 16
    n16 = (n_out + 15) // 16 * 16
    Ho, Wo = conv_out_size(H, kh, stride, pad, dil), conv_out_size(W, kw, stride, pad, dil)
    if Ho <= 0 or Wo <= 0:
        raise ValueError('convolution output would be empty')
    y = torch.empty((B, Ho, Wo, n16), dtype=torch.float32, device=x.device)
    out4 = fin = None
    if bn is not None and bn_partial is not None and INKERNEL_FINALIZE:
        # in-kernel finalize: the launch's last workgroup merges the partial records into (mean, invstd, scale, shift)
        out4 = torch.empty((4, n16), dtype=torch.float32, device=x.device)
        fin = _bn_fin_struct(bn, out4, x.device)
    parts = _tapgrid(x, B, H, W, C, in_nchw, wp, k_chunks, n16, bias, in_scale, in_shift, relu_in, y, Ho, Wo, Ho, Wo,
                     1, 0, 0, stride, kh, kw, -pad, dil, -pad, dil, 0, kw, 1, res=res, bn_partial=bn_partial, bn_fin=fin,
                     relu_out=relu_out)
    if bn is not None and bn_partial is not None:
        if parts > 0 and fin is None and finalize:      # two-launch form: merge the records with the finalize kernel
            out4 = bn_finalize(bn_partial, parts, *bn)
        return y, parts, out4    # parts == 0: no statistics from this launch (out4 is then unset); finalize=False: the caller
        #                          hands (bn_partial, parts) to bn_finalize_apply
    if bn_partial is not None:
        return y, parts          # parts == 0: the launch could not produce the statistics
    return y


def _dgrad_axis(parity, pad, dil, k, stride):
    """Taps of one output-parity class of a strided data gradient along one axis:
    returns (count, k0, kstep, in_off, in_step) or None if the class receives nothing."""
    valid = [t for t in range(k) if (parity + pad - t * dil) % stride == 0]
    if not valid:
        return None
    k0 = valid[0]
    kstep = valid[1] - valid[0] if len(valid) > 1 else 1
    return len(valid), k0, kstep, (parity + pad - k0 * dil) // stride, -(kstep * dil) // stride


def conv2d_dgrad(dy, wpt, n_in, H, W, kh, kw, stride=1, pad=0, dil=1, res=None, res_mask=None, accumulate_into=None,
                 bn_bwd=None, res_mask_bits=None, _diag_bias=None, pair_1x1=None, _s2_sums=None):
    """dy: NHWC [B,Ho,Wo,Cout]; wpt packed with transpose=True.  Returns dx NHWC [B,H,W,n_in16]
    (+ res * (res_mask > 0) if given).  With accumulate_into=dx0 the result is added to dx0 in place.
    bn_bwd=(x, save_mean, save_invstd, mask_scale, mask_shift[, mask_bits]) (stride 1; 3x3 / stride 2 / pad 1 with the mask_bits
    form): dx is the gradient reaching relu(bn(x)) (or, with
    mask_bits and scale = shift = None, relu(bn(x) + shortcut) whose sign bytes they are); returns
    (dx, partials) where partials = (records, count) for bn_backward(partials=) when the launch could also take that
    BatchNorm's two backward sums from its epilogue, else None (stride 1; with a residual only when its mask is also given
    as the sign bytes of bn_apply, res_mask_bits -- the float res_mask serves the launches that cannot use them)."""
    _lib.require_cuda(dy, wpt)
    if bn_bwd is not None and stride == 2:
        # the strided form: the upstream BatchNorm is a residual block's bn2 (mask = the sign bytes of the block output); only the
        # one-launch kernels of the thin stages take its sums, (dx, None) otherwise
        if accumulate_into is not None or res is not None or len(bn_bwd) < 6 or bn_bwd[5] is None or bn_bwd[3] is not None:
            raise ValueError('bn_bwd at stride 2: mask as sign bytes, no residual / accumulation')
        xb, mean, invstd, up_bits = bn_bwd[0], bn_bwd[1], bn_bwd[2], bn_bwd[5]
        _lib.require_cuda(xb)
        n16 = (n_in + 15) // 16 * 16
        if tuple(xb.shape) != (dy.shape[0], H, W, n16):
            raise ValueError('bn_bwd: x has the shape of the data gradient')
        rec = torch.empty(_lib.lib().dam_bn_workspace_floats(n16), dtype=torch.float32, device=dy.device)
        epi = _lib.BnBwdSums(_lib.ptr(xb), _lib.ptr(mean), _lib.ptr(invstd), None, None, None, _lib.ptr(up_bits))
        got = conv2d_dgrad(dy, wpt, n_in, H, W, kh, kw, stride, pad, dil, pair_1x1=pair_1x1, _s2_sums=(epi, rec))
        return got if isinstance(got, tuple) else (got, None)
    if bn_bwd is not None:
        if stride != 1 or accumulate_into is not None:
            raise ValueError('bn_bwd: stride-1 data gradients only')
        if res is not None and (res_mask is None or res_mask_bits is None):
            raise ValueError('bn_bwd with a residual: res_mask and res_mask_bits')
        _f32c(dy, 'dy'), _f32c(res, 'res'), _f32c(res_mask, 'res_mask')
        xb, mean, invstd, msc, msh = bn_bwd[:5]
        up_bits = bn_bwd[5] if len(bn_bwd) > 5 else None     # the BatchNorm's own mask as sign bytes (msc = msh = None then)
        if (up_bits is None) == (msc is None) or (up_bits is not None and res is None):
            raise ValueError('bn_bwd: the mask as (scale, shift) or -- with a residual -- as sign bytes')
        _lib.require_cuda(xb)
        B, Ho, Wo, Co = dy.shape
        n16 = (n_in + 15) // 16 * 16
        if tuple(xb.shape) != (B, H, W, n16):
            raise ValueError('bn_bwd: x has the shape of the data gradient')
        dx = torch.empty((B, H, W, n16), dtype=torch.float32, device=dy.device)
        rec = torch.empty(_lib.lib().dam_bn_workspace_floats(n16), dtype=torch.float32, device=dy.device)
        epi = _lib.BnBwdSums(_lib.ptr(xb), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(msc), _lib.ptr(msh), _lib.ptr(res_mask_bits),
                             _lib.ptr(up_bits))
        # (_diag_bias: timing builds only -- tools/dxhat_ladder.py hands the kernel a second input stream through the bias pointer)
        parts = _tapgrid(dy, B, Ho, Wo, Co, False, wpt, Co // 16, n16, _diag_bias, None, None, False, dx, H, W, H, W, 1, 0, 0, 1,
                         kh, kw, pad, -dil, pad, -dil, 0, kw, 1, res, res_mask, bn_partial=rec, bn_bwd=epi)
        return dx, ((rec, parts) if parts > 0 else None)
    _f32c(dy, 'dy'), _f32c(res, 'res'), _f32c(res_mask, 'res_mask')
    B, Ho, Wo, Co = dy.shape
    n16 = (n_in + 15) // 16 * 16
    if accumulate_into is not None:
        dx, res, res_mask = accumulate_into, accumulate_into, None
    else:
        dx = torch.empty((B, H, W, n16), dtype=torch.float32, device=dy.device)
    if stride == 1:
        _tapgrid(dy, B, Ho, Wo, Co, False, wpt, Co // 16, n16, None, None, None, False, dx, H, W, H, W, 1, 0, 0, 1,
                 kh, kw, pad, -dil, pad, -dil, 0, kw, 1, res, res_mask)
        return dx
    # pair_1x1=(dy2, wpt2): a SECOND operator's data gradient into the same dx -- a 1x1 / stride-`stride` / pad-0 convolution of the
    # same input (a down-sampling block's shortcut beside its conv1): its one tap lands on the pixels of this operator's single-tap
    # parity class (0, 0) and rides in that class's launch (dam_conv1x1_pair_f32) instead of a launch of its own that re-reads dx
    if pair_1x1 is not None:
        dy2, wpt2 = pair_1x1
        _lib.require_cuda(dy2, wpt2)
        _f32c(dy2, 'pair dy')
        if accumulate_into is not None or res is not None or tuple(dy2.shape) != tuple(dy.shape):
            raise ValueError('pair_1x1: a second gradient of the same shape, no residual / accumulation')
        a0 = _dgrad_axis(0, pad, dil, kh, stride), _dgrad_axis(0, pad, dil, kw, stride)
        if a0[0] is None or a0[1] is None or a0[0][0] != 1 or a0[1][0] != 1 or a0[0][3] != 0 or a0[1][3] != 0:
            raise ValueError('pair_1x1: class (0, 0) of this operator is not a single tap at offset 0')
    # 3x3 / stride 2 / pad 1 (the first convolution of a down-sampling block): all four parity classes (+ the pair term) from
    # one read of dy in one launch (dam_dgrad_s2_3x3_f32: weights resident in LDS for the thin layers, streamed from L2 for the wide)
    if (DGRAD_S2 and stride == 2 and kh == 3 and kw == 3 and pad == 1 and dil == 1 and accumulate_into is None and res is None
            and Ho == (H + 1) // 2 and Wo == (W + 1) // 2):
        parts = ctypes.c_int(0)
        st = _lib.lib().dam_dgrad_s2_3x3_f32(_lib.ptr(dy), _lib.ptr(wpt), _lib.ptr(pair_1x1[0]) if pair_1x1 else None,
                                             _lib.ptr(pair_1x1[1]) if pair_1x1 else None, B, Ho, Wo, Co, n16, _lib.ptr(dx), H, W,
                                             ctypes.byref(_s2_sums[0]) if _s2_sums else None,
                                             _lib.ptr(_s2_sums[1]) if _s2_sums else None, ctypes.byref(parts) if _s2_sums else None,
                                             _lib.stream())
        if st == 0:
            global dgrad_s2_launches
            dgrad_s2_launches += 1
            if _s2_sums:
                return dx, ((_s2_sums[1], parts.value) if parts.value > 0 else None)
            return dx
        if st != -2:                                   # DAM_ERR_UNSUPPORTED: not a layer that kernel takes -> the class launches
            _lib.check(st, 'dam_dgrad_s2_3x3_f32')
    classes = [(_dgrad_axis(ph, pad, dil, kh, stride), _dgrad_axis(pw, pad, dil, kw, stride))
               for ph in range(stride) for pw in range(stride)]
    if any(a is None or b is None for a, b in classes) and accumulate_into is None:
        if res is not None:
            raise ValueError('a fused residual needs every output parity class to receive taps')
        dx.zero_()          # classes without taps get no gradient (e.g. the 1x1 stride-2 shortcut)
    # the classes write disjoint pixels of dx from the same dy and weights: those that take the tile kernel are recorded
    # and run as ONE launch (class in blockIdx.z) at the flush
    batch = _conv_batch(dy.device)
    try:
        for ph in range(stride):
            for pw in range(stride):
                ah, aw = _dgrad_axis(ph, pad, dil, kh, stride), _dgrad_axis(pw, pad, dil, kw, stride)
                nh, nw = (H - ph + stride - 1) // stride, (W - pw + stride - 1) // stride
                if nh <= 0 or nw <= 0 or ah is None or aw is None:
                    continue
                if pair_1x1 is not None and ph == 0 and pw == 0:
                    if nh != Ho or nw != Wo:
                        raise ValueError('pair_1x1: class (0, 0) does not cover the gradient\'s pixels')
                    _lib.check(_lib.lib().dam_conv1x1_pair_f32(_lib.ptr(dy), _lib.ptr(wpt), ah[1] * kw + aw[1], _lib.ptr(pair_1x1[0]),
                                                               _lib.ptr(pair_1x1[1]), 0, B, Ho, Wo, Co, n16, _lib.ptr(dx), H, W, stride,
                                                               0, 0, _lib.stream()), 'dam_conv1x1_pair_f32')
                    continue
                _tapgrid(dy, B, Ho, Wo, Co, False, wpt, Co // 16, n16, None, None, None, False, dx, H, W, nh, nw,
                         stride, ph, pw, 1, ah[0], aw[0], ah[3], ah[4], aw[3], aw[4], ah[1] * kw + aw[1],
                         ah[2] * kw, aw[2], res, res_mask, batch=batch)
        _lib.check(_lib.lib().dam_conv_batch_flush(ctypes.addressof(batch), _lib.stream()), 'dam_conv_batch_flush')
    except Exception:
        _lib.lib().dam_conv_batch_init(ctypes.addressof(batch))      # drop what a failed call left recorded
        raise
    return dx


_conv_batches = {}


def _conv_batch(device):
    """The device's launch batch for sibling tile-kernel launches (include/dam_hip.h: dam_conv_batch_*)."""
    key = (device.type, device.index)
    b = _conv_batches.get(key)
    if b is None:
        L = _lib.lib()
        b = _conv_batches[key] = ctypes.create_string_buffer(int(L.dam_conv_batch_bytes()))
        _lib.check(L.dam_conv_batch_init(ctypes.addressof(b)), 'dam_conv_batch_init')
    return b


_workspaces = {}
_retired = []          # superseded buffers are kept: a captured hipGraph may have their address baked in


def _grow(table, key, device, floats):
    ws = table.get(key)
    if ws is None or ws.numel() < floats:
        if ws is not None:
            _retired.append(ws)
        ws = torch.empty(int(floats), dtype=torch.float32, device=device)
        table[key] = ws
    return ws


def _workspace(device, floats):
    """One grow-only scratch buffer per device and stream role (split-K slabs, weight-gradient slabs, reduction scratch of
    the heads / mask-sum); all users of one buffer are ordered on one stream -- launches made inside side_stream_run()
    get their own.  A buffer that has to grow is never freed: hipGraphs captured earlier keep replaying into the old one
    (the launches recorded there were sized for it)."""
    key = (device.type, device.index, 'side') if _side_active else (device.type, device.index)
    return _grow(_workspaces, key, device, floats)


# -- side stream for the leaves of the backward pass ------------------------------------------------------------------------
# A weight gradient is a leaf: nothing in the backward chain (BatchNorm backward -> data gradient -> next layer) waits for it,
# so it can run on a second stream beside the chain (under hipGraph capture: a parallel branch of the graph).  MEASURED, C3
# ResNet18 step: 5.57 ms on one stream, 5.96 ms with the weight gradients on the side stream -- the kernels do overlap
# (profiles/r02_side_stream_trace_summary.txt) but the big ones are one-workgroup-per-CU persistent kernels with 60-150 KB
# of LDS: two of them cannot share a CU, so each runs 1.3-1.7 x longer (conv_strip 66 -> 85-92 us, wgrad_rows 54 -> 79-96 us)
# and the sum grows.  Hence OFF by default; DAM_SIDE_STREAM=1 turns it on for the A/B.  Only gradients written in place into a
# caller-owned buffer (Adam's flat gradient bucket) take it: the consumer joins with side_stream_join() before it reads them.
SIDE_STREAM = bool(os.environ.get('DAM_SIDE_STREAM'))
_side_active = False
_side_streams = {}
_side_dirty = set()


def side_stream_run(fn, reads, device):
    """Runs fn() (kernel launches only) on the device's side stream, ordered after everything issued so far on the current
    stream.  `reads`: the tensors those launches read -- their memory is not handed out again before the side stream is done."""
    global _side_active
    if not SIDE_STREAM or device.type != 'cuda' or _side_active:
        return fn()
    key = (device.type, device.index)
    side = _side_streams.get(key)
    if side is None:
        side = _side_streams[key] = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    _side_active = True
    try:
        with torch.cuda.stream(side):
            out = fn()
    finally:
        _side_active = False
    for t in reads:
        if t is not None:
            t.record_stream(side)
    _side_dirty.add(key)
    return out


def side_stream_join(device=None):
    """The current stream waits for everything side_stream_run() has issued (call before reading its results, and before
    the end of a hipGraph capture that contains such launches)."""
    for key in list(_side_dirty):
        if device is not None and key != (device.type, device.index):
            continue
        torch.cuda.current_stream(torch.device(*key)).wait_stream(_side_streams[key])
        _side_dirty.discard(key)


# Parameters and BatchNorm running statistics are updated IN PLACE by kernels of this library (the fused Adam launch, the
# statistics finalize), possibly inside a replayed hipGraph: torch's tensor version counters do not see that.  Whatever
# caches something derived from them (layers.FoldedConvBn) keys on this counter too; it is bumped by every training forward
# (the statistics finalize updates the running buffers), every optimizer launch and every graph replay of a step.
PARAM_EPOCH = 0


def params_changed():
    global PARAM_EPOCH
    PARAM_EPOCH += 1


# The equal weight gradients of a deep stage as one launch (include/dam_hip.h: dam_wgrad_queue_set_batching); DAM_WGRAD_BATCH=0: A/B
WGRAD_BATCH = os.environ.get('DAM_WGRAD_BATCH', '1') != '0'
_wgrad_queues = {}      # device -> [ctypes buffer of the library's queue, calls recorded since the last flush]


def _wgrad_queue(device):
    key = (device.type, device.index)
    q = _wgrad_queues.get(key)
    if q is None:
        L = _lib.lib()
        buf = ctypes.create_string_buffer(int(L.dam_wgrad_queue_bytes()))
        _lib.check(L.dam_wgrad_queue_init(ctypes.addressof(buf)), 'dam_wgrad_queue_init')
        if WGRAD_BATCH:
            _lib.check(L.dam_wgrad_queue_set_batching(ctypes.addressof(buf), 1), 'dam_wgrad_queue_set_batching')
        # [the library's queue, calls recorded since the last flush, tensors those calls read (kept alive until the flush: with
        #  batching a recorded call's slab kernel may not have been launched yet)]
        q = _wgrad_queues[key] = [buf, 0, []]
    return q


def wgrad_abandon(device=None):
    """Drops reductions that were recorded but never flushed -- a backward pass that raised half-way leaves them behind,
    and the next step's jobs would target the same gradients from the same flush launch.  Called at the start of every
    step (optim.Adam.zero_grad, engine.TrainStep); a no-op in the normal case."""
    for key, q in _wgrad_queues.items():
        if q[1] and (device is None or key == (device.type, device.index)):
            _lib.check(_lib.lib().dam_wgrad_queue_init(ctypes.addressof(q[0])), 'dam_wgrad_queue_init')
            del q[2][:]
            q[1] = 0


def wgrad_flush(device=None):
    """Runs the slab reductions recorded by conv2d_wgrad(..., defer=True) in one launch on the current stream; the
    gradients are in their `out` buffers when that launch is done.  No-op when nothing is recorded."""
    for key, q in _wgrad_queues.items():
        if q[1] and (device is None or key == (device.type, device.index)):
            _lib.check(_lib.lib().dam_wgrad_queue_flush(ctypes.addressof(q[0]), _lib.stream()), 'dam_wgrad_queue_flush')
            q[1] = 0
            del q[2][:]


def conv2d_wgrad(x, dy, n_out, kh, kw, stride=1, pad=0, dil=1, in_scale=None, in_shift=None, relu_in=False,
                 in_nchw=False, out=None, c_real=None, defer=False):
    """dW in torch layout [n_out, c_real or C, kh, kw] from x (NHWC, or NCHW first layer) and dy NHWC [B,Ho,Wo,n16].
    out: where to write it (e.g. the parameter's slice of a flat gradient buffer).
    defer (needs out): only the slab kernel runs now; the reduction into `out` is recorded and happens in wgrad_flush(),
    one launch for every weight gradient of the backward pass (each deferred call keeps its own slab buffer until then)."""
    _lib.require_cuda(x, dy)
    _f32c(x, 'x'), _f32c(dy, 'dy')
    if in_nchw:
        B, C, H, W = x.shape
    else:
        B, H, W, C = x.shape
    _, Ho, Wo, n_chan = dy.shape
    L = _lib.lib()
    floats = L.dam_conv2d_wgrad_workspace_floats(n_out, C, kh, kw)
    cr = C if c_real is None else int(c_real)
    if out is None:
        if defer:
            raise ValueError('a deferred weight gradient needs the buffer it will be written to')
        out = torch.empty((n_out, cr, kh, kw), dtype=torch.float32, device=x.device)
    elif out.numel() != n_out * cr * kh * kw or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError('bad out tensor for the weight gradient')
    if defer:
        q = _wgrad_queue(x.device)
        ws = _grow(_workspaces, (x.device.type, x.device.index, 'wgrad slabs', q[1]), x.device, floats)
        q[1] += 1
        q[2].append((x, dy, in_scale, in_shift))
        queue = ctypes.addressof(q[0])
    else:
        ws, queue = _workspace(x.device, floats), None
    _lib.check(L.dam_conv2d_wgrad_f32(_lib.ptr(x), B, H, W, C, 1 if in_nchw else 0, _lib.ptr(in_scale),
                                      _lib.ptr(in_shift), 1 if relu_in else 0, _lib.ptr(dy), Ho, Wo, n_chan, n_out,
                                      kh, kw, stride, pad, dil, _lib.ptr(out), cr, _lib.ptr(ws), ws.numel(), queue,
                                      _lib.stream()), 'dam_conv2d_wgrad_f32')
    return out


def nchw_to_nhwc16(x):
    """[B,C,H,W] float32 (C <= 16) -> NHWC [B,H,W,16], channels C..15 zero."""
    _lib.require_cuda(x)
    _f32c(x, 'x')
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, 16), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dam_nchw_to_nhwc16_f32(_lib.ptr(x), B, C, H * W, _lib.ptr(y), _lib.stream()), 'dam_nchw_to_nhwc16_f32')
    return y


# ----------------------------------------------------------------------------- batch norm
def _bn_ws(device, C):
    return _workspace(device, _lib.lib().dam_bn_workspace_floats(C))


_bn_partials = {}


def bn_stats(x, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps):
    """Training-mode statistics of NHWC x.  Returns (save_mean, save_invstd, scale, shift), updates the running buffers."""
    _lib.require_cuda(x)
    C = x.shape[-1]
    P = x.numel() // C
    out = torch.empty((4, C), dtype=torch.float32, device=x.device)
    ws = _bn_ws(x.device, C)
    _lib.check(_lib.lib().dam_bn_stats_f32(_lib.ptr(x), P, C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                                           _lib.ptr(running_var), _lib.ptr(num_batches_tracked), float(momentum), float(eps),
                                           _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]), _lib.ptr(out[3]),
                                           _lib.ptr(ws), _lib.ptr(arrival_counter(x.device)), _lib.stream()), 'dam_bn_stats_f32')
    return out[0], out[1], out[2], out[3]


def bn_stats_partial(x):
    """The partial statistics records of NHWC x alone: (records, count) for bn_finalize_apply / bn_finalize."""
    _lib.require_cuda(x)
    C = x.shape[-1]
    ws = bn_partial_buffer(x.device, C)
    parts = ctypes.c_int(0)
    _lib.check(_lib.lib().dam_bn_stats_partial_f32(_lib.ptr(x), x.numel() // C, C, _lib.ptr(ws), ctypes.byref(parts), _lib.stream()),
               'dam_bn_stats_partial_f32')
    return ws, parts.value


def bn_finalize_apply(partial, parts, bn, x, relu=True, res=None, res_scale=None, res_shift=None, sign_bits=False):
    """bn_finalize + bn_apply in ONE launch (dam_bn_finalize_apply_f32): bn = (gamma, beta, running_mean, running_var,
    num_batches_tracked, momentum, eps).  Returns ((save_mean, save_invstd, scale, shift), y or (y, bits))."""
    _lib.require_cuda(partial, x)
    C = x.shape[-1]
    out4 = torch.empty((4, C), dtype=torch.float32, device=x.device)
    fin = _bn_fin_struct(bn, out4, x.device)
    y = torch.empty_like(x)
    bits = torch.empty(x.shape[:-1] + (C // 4,), dtype=torch.uint8, device=x.device) if sign_bits else None
    _lib.check(_lib.lib().dam_bn_finalize_apply_f32(_lib.ptr(partial), int(parts), C, ctypes.byref(fin), _lib.ptr(x), x.numel() // C,
                                                    _lib.ptr(res), _lib.ptr(res_scale), _lib.ptr(res_shift), 1 if relu else 0,
                                                    _lib.ptr(y), _lib.ptr(bits), _lib.stream()), 'dam_bn_finalize_apply_f32')
    return (out4[0], out4[1], out4[2], out4[3]), ((y, bits) if sign_bits else y)


def bn_partial_buffer(device, C):
    """The BatchNorm partial records a convolution launch can emit: a buffer of their own per device and width (they
    live from the convolution launch to the finalize launch and must not alias any kernel's scratch)."""
    return _grow(_bn_partials, (device.type, device.index, C), device, _lib.lib().dam_bn_workspace_floats(C))


def bn_stats_pair(xa, bn_a, xb, bn_b):
    """bn_stats for two tensors of one shape in one partial + one finalize launch.  bn_a, bn_b = (gamma, beta, running_mean,
    running_var, num_batches_tracked, momentum, eps).  Returns two (save_mean, save_invstd, scale, shift) tuples."""
    _lib.require_cuda(xa, xb)
    if xa.shape != xb.shape:
        raise ValueError('the two tensors of a pair have the same shape')
    C = xa.shape[-1]
    L = _lib.lib()
    outs = [torch.empty((4, C), dtype=torch.float32, device=xa.device) for _ in range(2)]
    fa, fb = _bn_fin_struct(bn_a, outs[0], xa.device), _bn_fin_struct(bn_b, outs[1], xa.device)
    ws = _workspace(xa.device, 2 * L.dam_bn_workspace_floats(C))
    _lib.check(L.dam_bn_stats_pair_f32(_lib.ptr(xa), _lib.ptr(xb), xa.numel() // C, C, ctypes.byref(fa), ctypes.byref(fb),
                                       _lib.ptr(ws), _lib.stream()), 'dam_bn_stats_pair_f32')
    return tuple(outs[0]), tuple(outs[1])


def conv_s2_pair_fwd(x, wp, wp_sc, n_out, stats=True):
    """conv1 (3x3 / stride 2 / pad 1) and the 1x1 / stride-2 shortcut convolution of one NHWC input in one launch:
    (c1, cs, (records1, records_sc, parts) or None), or None when the layer is not one dam_conv_s2_pair_fwd_f32 takes."""
    _lib.require_cuda(x, wp, wp_sc)
    _f32c(x, 'x')
    B, H, W, C = x.shape
    n16 = (n_out + 15) // 16 * 16
    Hd, Wd = (H + 1) // 2, (W + 1) // 2
    c1 = torch.empty((B, Hd, Wd, n16), dtype=torch.float32, device=x.device)
    cs = torch.empty_like(c1)
    p1 = p2 = None
    parts = ctypes.c_int(0)
    if stats:
        n = _lib.lib().dam_bn_workspace_floats(n16)
        p1 = torch.empty(n, dtype=torch.float32, device=x.device)
        p2 = torch.empty(n, dtype=torch.float32, device=x.device)
    st = _lib.lib().dam_conv_s2_pair_fwd_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(wp_sc), B, H, W, C, n16, _lib.ptr(c1), _lib.ptr(cs),
                                             _lib.ptr(p1), _lib.ptr(p2), ctypes.byref(parts), _lib.stream())
    if st == -2:                                       # DAM_ERR_UNSUPPORTED
        return None
    _lib.check(st, 'dam_conv_s2_pair_fwd_f32')
    return c1, cs, ((p1, p2, parts.value) if stats else None)


def bn_finalize_pair(partial_a, partial_b, parts, bn_a, bn_b):
    """bn_finalize for two BatchNorms whose records have one count (conv_s2_pair_fwd): one launch.  bn_a, bn_b as bn_stats_pair.
    Returns two (save_mean, save_invstd, scale, shift) tuples."""
    C = bn_a[0].numel()
    outs = [torch.empty((4, C), dtype=torch.float32, device=partial_a.device) for _ in range(2)]
    fa, fb = _bn_fin_struct(bn_a, outs[0], partial_a.device), _bn_fin_struct(bn_b, outs[1], partial_a.device)
    _lib.check(_lib.lib().dam_bn_finalize_pair_f32(_lib.ptr(partial_a), _lib.ptr(partial_b), parts, C, ctypes.byref(fa),
                                                   ctypes.byref(fb), _lib.stream()), 'dam_bn_finalize_pair_f32')
    return tuple(outs[0]), tuple(outs[1])


def bn_finalize(partial, parts, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps):
    """Merges `parts` partial records (from conv2d_fwd(..., bn_partial=...)) -> (save_mean, save_invstd, scale, shift)."""
    C = gamma.numel()
    out = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    _lib.check(_lib.lib().dam_bn_finalize_f32(_lib.ptr(partial), parts, C, _lib.ptr(gamma), _lib.ptr(beta),
                                              _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(num_batches_tracked),
                                              float(momentum), float(eps), _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]),
                                              _lib.ptr(out[3]), _lib.stream()), 'dam_bn_finalize_f32')
    return out[0], out[1], out[2], out[3]


def bn_eval_affine(gamma, beta, running_mean, running_var, eps):
    _lib.require_cuda(gamma)
    C = gamma.numel()
    out = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    _lib.check(_lib.lib().dam_bn_eval_affine_f32(C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                                                 _lib.ptr(running_var), float(eps), _lib.ptr(out[0]), _lib.ptr(out[1]),
                                                 _lib.ptr(out[2]), _lib.ptr(out[3]), _lib.stream()), 'dam_bn_eval_affine_f32')
    return out[0], out[1], out[2], out[3]


def bn_apply(x, scale, shift, relu=True, res=None, res_scale=None, res_shift=None, sign_bits=False):
    """y = relu?(x*scale + shift [+ res [*res_scale + res_shift]]).  sign_bits=True: returns (y, bits) with one uint8 per
    channel quad, bit i = (y[..., 4q + i] > 0): what bn_backward(mask_bits=) reads instead of y."""
    _lib.require_cuda(x)
    C = x.shape[-1]
    y = torch.empty_like(x)
    bits = torch.empty(x.shape[:-1] + (C // 4,), dtype=torch.uint8, device=x.device) if sign_bits else None
    _lib.check(_lib.lib().dam_bn_apply_f32(_lib.ptr(x), x.numel() // C, C, _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(res),
                                           _lib.ptr(res_scale), _lib.ptr(res_shift), 1 if relu else 0, _lib.ptr(y),
                                           _lib.ptr(bits), _lib.stream()), 'dam_bn_apply_f32')
    return (y, bits) if sign_bits else y


def bn_backward_pair(dy, y_mask, a, b, training=True, mask_bits=None):
    """bn_backward for two BatchNorms fed by the same dy through the same ReLU mask (a residual block's bn2 and its shortcut
    BatchNorm): three launches instead of six, dy / y_mask read once per pass.  a, b = (x, gamma, save_mean, save_invstd,
    dgamma or None, dbeta or None).  Returns ((dx, dgamma, dbeta), (dx, dgamma, dbeta)), bitwise equal to two bn_backward calls."""
    _lib.require_cuda(dy, a[0], b[0])
    if (y_mask is None) == (mask_bits is None):
        raise ValueError('exactly one of y_mask / mask_bits')
    C = a[0].shape[-1]
    if b[0].shape != a[0].shape or dy.shape != a[0].shape:
        raise ValueError('the two BatchNorms of a pair see the same shape')
    outs = []
    for x, gamma, mean, invstd, dgamma, dbeta in (a, b):
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if dgamma is None else dgamma
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if dbeta is None else dbeta
        outs.append((dx, dgamma, dbeta))
    L = _lib.lib()
    ws = _workspace(dy.device, L.dam_bn_pair_workspace_floats(C))
    args = []
    for (x, gamma, mean, invstd, _, _), (dx, dgamma, dbeta) in zip((a, b), outs):
        args += [_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta)]
    _lib.check(L.dam_bn_backward_pair_f32(_lib.ptr(dy), _lib.ptr(y_mask), _lib.ptr(mask_bits), dy.numel() // C, C,
                                          1 if training else 0, *args,
                                          _lib.ptr(ws), _lib.stream()), 'dam_bn_backward_pair_f32')
    return outs[0], outs[1]


def bn_backward(dy, y_mask, x, gamma, save_mean, save_invstd, training=True, mask_affine=None, dgamma=None, dbeta=None,
                mask_bits=None, partials=None):
    """Returns (dx, dgamma, dbeta) for y = [relu](bn(x) + ...).  The ReLU mask comes from y_mask (the saved output), or --
    for a plain relu(bn(x)) -- from mask_affine=(scale, shift), the forward's fused affine (the saved output is not read),
    or there is none (both None)."""
    _lib.require_cuda(dy, x)
    C = x.shape[-1]
    dx = torch.empty_like(x)
    # two separate allocations: autograd takes ownership of a whole tensor it is handed as a .grad, but clones a view
    # (30 extra device copies per ResNet18 step when these were rows of one [2, C] tensor)
    if dgamma is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    # partials=(records, count) from conv2d_dgrad(bn_bwd=): the two sums are already there, only finalize + apply run
    ws, given = (_bn_ws(x.device, C), 0) if partials is None else partials
    _lib.check(_lib.lib().dam_bn_backward_f32(_lib.ptr(dy), _lib.ptr(y_mask), _lib.ptr(x), x.numel() // C, C, _lib.ptr(gamma),
                                              _lib.ptr(save_mean), _lib.ptr(save_invstd), 1 if training else 0,
                                              _lib.ptr(mask_affine[0]) if mask_affine else None,
                                              _lib.ptr(mask_affine[1]) if mask_affine else None, _lib.ptr(mask_bits), _lib.ptr(dx),
                                              _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws), given,
                                              _lib.ptr(arrival_counter(x.device)), _lib.stream()),
               'dam_bn_backward_f32')
    return dx, dgamma, dbeta


def channel_sum(x, n_real, out=None):
    _lib.require_cuda(x)
    C = x.shape[-1]
    if out is None:
        out = torch.empty(n_real, dtype=torch.float32, device=x.device)
    ws = _bn_ws(x.device, C)
    _lib.check(_lib.lib().dam_channel_sum_f32(_lib.ptr(x), x.numel() // C, C, n_real, _lib.ptr(out), _lib.ptr(ws),
                                              _lib.stream()), 'dam_channel_sum_f32')
    return out


# ----------------------------------------------------------------------------- heads / mask-sum / loss
def heads_fwd(trunk, conv_w, conv_b, fc_w, fc_b):
    """trunk NHWC [B,h,w,C]; conv_w [S,C]; conv_b [S]; fc_w [S,P]; fc_b [S] -> (h [B,S,P], gains [B,S])."""
    _lib.require_cuda(trunk)
    B, C = trunk.shape[0], trunk.shape[-1]
    P = trunk.numel() // (B * C)
    S = conv_w.shape[0]
    if fc_w.shape != (S, P):
        raise ValueError('fc_head expects %d inputs but the trunk gives %d (flattened_dim mismatch)' % (fc_w.shape[1], P))
    h = torch.empty((B, S, P), dtype=torch.float32, device=trunk.device)
    g = torch.empty((B, S), dtype=torch.float32, device=trunk.device)
    _lib.check(_lib.lib().dam_heads_fwd_f32(_lib.ptr(trunk), B, P, C, S, _lib.ptr(conv_w), _lib.ptr(conv_b), _lib.ptr(fc_w),
                                            _lib.ptr(fc_b), _lib.ptr(h), _lib.ptr(g), _lib.stream()), 'dam_heads_fwd_f32')
    return h, g


def heads_bwd(dgains, h, trunk, conv_w, fc_w, outs=None):
    """outs: optional (dconv_w [S,C], dconv_b [S], dfc_w [S,P], dfc_b [S]) destinations (gradient slots)."""
    B, S, P = h.shape
    C = trunk.shape[-1]
    dev = trunk.device
    dtrunk = torch.empty_like(trunk)
    o = outs or (None, None, None, None)
    dcw = o[0] if o[0] is not None else torch.empty((S, C), dtype=torch.float32, device=dev)
    dcb = o[1] if o[1] is not None else torch.empty(S, dtype=torch.float32, device=dev)
    dfw = o[2] if o[2] is not None else torch.empty((S, P), dtype=torch.float32, device=dev)
    dfb = o[3] if o[3] is not None else torch.empty(S, dtype=torch.float32, device=dev)
    L = _lib.lib()
    ws = _workspace(dev, L.dam_heads_bwd_workspace_floats(B, P, C, S))
    _lib.check(L.dam_heads_bwd_f32(_lib.ptr(dgains), _lib.ptr(h), _lib.ptr(trunk), B, P, C, S, _lib.ptr(conv_w), _lib.ptr(fc_w),
                                   _lib.ptr(dtrunk), _lib.ptr(dcw), _lib.ptr(dcb), _lib.ptr(dfw), _lib.ptr(dfb), _lib.ptr(ws),
                                   _lib.stream()), 'dam_heads_bwd_f32')
    return dtrunk, dcw, dcb, dfw, dfb


def masksum_fwd(x, gains):
    """x [B,S,F,T], gains [B,S] -> masked [B,F,T]."""
    _lib.require_cuda(x, gains)
    B, S, F, T = x.shape
    masked = torch.empty((B, F, T), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dam_masksum_fwd_f32(_lib.ptr(x), _lib.ptr(gains), B, S, F * T, _lib.ptr(masked), _lib.stream()),
               'dam_masksum_fwd_f32')
    return masked


def masksum_bwd(dmasked, x):
    B, S, F, T = x.shape
    dg = torch.empty((B, S), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws = _workspace(x.device, L.dam_masksum_workspace_floats(B, S))
    _lib.check(L.dam_masksum_bwd_f32(_lib.ptr(dmasked), _lib.ptr(x), B, S, F * T, _lib.ptr(dg), _lib.ptr(ws), _lib.stream()),
               'dam_masksum_bwd_f32')
    return dg


def masksum_mse(x, gains, gt, want_masked=True):
    """Fused masked-sum + MSE: returns (masked or None, loss [1], dloss/dgains [B,S])."""
    _lib.require_cuda(x, gains, gt)
    B, S, F, T = x.shape
    dev = x.device
    masked = torch.empty((B, F, T), dtype=torch.float32, device=dev) if want_masked else None
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dg = torch.empty((B, S), dtype=torch.float32, device=dev)
    L = _lib.lib()
    ws = _workspace(dev, L.dam_masksum_workspace_floats(B, S))
    _lib.check(L.dam_masksum_mse_f32(_lib.ptr(x), _lib.ptr(gains), _lib.ptr(gt), B, S, F * T, _lib.ptr(masked), _lib.ptr(loss),
                                     _lib.ptr(dg), _lib.ptr(ws), _lib.stream()), 'dam_masksum_mse_f32')
    return masked, loss, dg


# ----------------------------------------------------------------------------- optimizer
def adam_l2_step(params, grads, exp_avg, exp_avg_sq, step, derived, lr, beta1, beta2, eps, weight_decay, grad_scale=1.0,
                 hyper=None):
    """hyper: optional CUDA float32[8] {lr, beta1, beta2, eps, weight_decay, grad_scale, 1-beta1, 1-beta2} read by the kernel at run time
    (a captured graph then follows hyper-parameter edits); the scalars are used when it is None."""
    _lib.require_cuda(params, grads, hyper)
    _lib.check(_lib.lib().dam_adam_l2_step_f32(_lib.ptr(params), _lib.ptr(grads), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq),
                                               params.numel(), _lib.ptr(step), _lib.ptr(derived), float(lr), float(beta1),
                                               float(beta2), float(eps), float(weight_decay), float(grad_scale),
                                               _lib.ptr(hyper), _lib.stream()), 'dam_adam_l2_step_f32')


# ----------------------------------------------------------------------------- inference tail
def _audio_kind(t, what):
    if t.dtype not in (torch.float32, torch.float64):
        raise TypeError('%s must be float32 or float64' % what)
    return 1 if t.dtype == torch.float64 else 0


def gains_smooth(raw_db, window, polyorder=2, out=None, want_f32=False):
    """raw_db: CUDA float32 [n_chunks, S] (model outputs) -> out [2, S, n_chunks] float64 = (10 ** (0.5 * g),
    scipy.signal.savgol_filter(that, window, polyorder) in its default mode 'interp'), computed on the device.
    Returns (amp, smooth[, smooth as float32])."""
    _lib.require_cuda(raw_db, out)
    raw_db = _f32c(raw_db, 'raw_db')
    n, S = raw_db.shape
    if window % 2 == 0 or window <= polyorder or window > n:
        # the conditions scipy.signal.savgol_filter rejects (inference_utils.py:140 would raise there too)
        raise ValueError('savgol_filter needs an odd window with polyorder < window <= len(x): window=%d polyorder=%d '
                         'len=%d' % (window, polyorder, n))
    if out is None:
        out = torch.empty((2, S, n), dtype=torch.float64, device=raw_db.device)
    elif tuple(out.shape) != (2, S, n) or out.dtype != torch.float64 or not out.is_contiguous():
        raise ValueError('bad out tensor')
    s32 = torch.empty((S, n), dtype=torch.float32, device=raw_db.device) if want_f32 else None
    _lib.check(_lib.lib().dam_gains_smooth(_lib.ptr(raw_db), n, S, window, polyorder, _lib.ptr(out[0]), _lib.ptr(out[1]),
                                           _lib.ptr(s32), _lib.stream()), 'dam_gains_smooth')
    return (out[0], out[1], s32) if want_f32 else (out[0], out[1])


def gain_ramp_apply(audio, gains, out=None, out_dtype=None):
    """audio [G, rows_per_gain, n] or [rows, n] (CUDA float32/float64), gains [G, n_gains] or [n_gains] float64 ->
    audio * piecewise-constant gain (one gain sequence per leading index); the result is float64 unless out_dtype says
    float32 (numpy's float32-track * float64-mask product is float64, inference_utils.py:143)."""
    _lib.require_cuda(audio, gains, out)
    ak = _audio_kind(audio, 'audio')
    if gains.dtype != torch.float64:
        raise TypeError('gains must be float64')
    audio, gains = audio.contiguous(), gains.contiguous()
    if audio.dim() == 2:
        rpg, rows, n = audio.shape[0], audio.shape[0], audio.shape[1]
        n_gains = gains.numel()
    else:
        G, rpg, n = audio.shape
        rows, n_gains = G * rpg, gains.shape[-1]
        if gains.numel() != G * n_gains:
            raise ValueError('one gain sequence per leading index expected')
    if out is None:
        out = torch.empty(audio.shape, dtype=out_dtype or torch.float64, device=audio.device)
    elif out.shape != audio.shape or not out.is_contiguous():
        raise ValueError('bad out tensor')
    _lib.check(_lib.lib().dam_gain_ramp_apply(_lib.ptr(audio), ak, _lib.ptr(gains), rows, rpg, n, n_gains, _lib.ptr(out),
                                              _audio_kind(out, 'out'), _lib.stream()), 'dam_gain_ramp_apply')
    return out


def mixdown_peak_normalize(audio, gains, normalize=True, out=None, out_dtype=None, workspace=None):
    """audio [S, rows, n] (CUDA float32/float64), gains [S, n_gains] float64 -> gain-ramped stem sum [rows, n], optionally
    divided row-wise by its max-abs."""
    _lib.require_cuda(audio, gains, out)
    ak = _audio_kind(audio, 'audio')
    if gains.dtype != torch.float64:
        raise TypeError('gains must be float64')
    audio, gains = audio.contiguous(), gains.contiguous()
    S, rows, n = audio.shape
    if out is None:
        out = torch.empty((rows, n), dtype=out_dtype or torch.float64, device=audio.device)
    L = _lib.lib()
    if workspace is None:
        workspace = torch.empty(L.dam_mixdown_workspace_elems(rows), dtype=out.dtype, device=audio.device)
    _lib.check(L.dam_mixdown_peak_normalize(_lib.ptr(audio), ak, _lib.ptr(gains), S, rows, n, gains.shape[1],
                                            1 if normalize else 0, _lib.ptr(out), _audio_kind(out, 'out'),
                                            _lib.ptr(workspace), _lib.stream()), 'dam_mixdown_peak_normalize')
    return out


# ----------------------------------------------------------------------------- dropout
_dropout_counters = {}


def dropout_tick(device, n):
    """Snapshots and advances the per-device dropout call counter; returns the snapshot tensor (int64[1])."""
    key = (device.type, device.index)
    if key not in _dropout_counters:
        _dropout_counters[key] = torch.zeros(1, dtype=torch.int64, device=device)
    snap = torch.empty(1, dtype=torch.int64, device=device)
    _lib.check(_lib.lib().dam_dropout_tick(_lib.ptr(_dropout_counters[key]), n, _lib.ptr(snap), _lib.stream()), 'dam_dropout_tick')
    return snap


def dropout_apply(x, p, seed, snapshot):
    _lib.require_cuda(x, snapshot)
    x = x.contiguous()
    y = torch.empty_like(x)
    _lib.check(_lib.lib().dam_dropout_apply_f32(_lib.ptr(x), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(snapshot),
                                                _lib.ptr(y), _lib.stream()), 'dam_dropout_apply_f32')
    return y
