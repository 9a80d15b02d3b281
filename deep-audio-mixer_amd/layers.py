"""Building blocks shared by the three models: autograd Functions whose forward AND backward are
sequences of libdam_hip.so kernels (no torch arithmetic on the path), plus the nn.Modules that own
the parameters under the reference's state_dict names.

Granularity: one Function per conv+BN+ReLU block, per residual BasicBlock and for the gain heads, so
that the backward pass can use the fused forms (BatchNorm backward -> dgrad with the residual
gradient added in its epilogue, shortcut dgrad accumulated in place, one pass for all S heads).
Activations between Functions are NHWC float32.
"""
import torch
import torch.nn as nn

from . import ops


def _slot(p):
    """The parameter's slice of a flat gradient buffer, if an optimizer bound one (optim.Adam.bind_grad_slots): the
    backward kernels then write the gradient THERE and the autograd Function returns None for that input -- no .grad
    tensor, no AccumulateGrad copy, no gather launch.  Overwrite semantics (one backward per zero_grad), which is what
    the step engine does; without a bound slot the gradient flows through autograd as usual."""
    return None if p is None else getattr(p, '_dam_grad', None)


class ConvSpec:
    """Static description of one convolution (not a tensor)."""

    def __init__(self, cin, cout, k, stride=1, pad=0, dil=1, in_nchw=False):
        self.cin, self.cout, self.k, self.stride, self.pad, self.dil, self.in_nchw = cin, cout, k, stride, pad, dil, in_nchw
        self.wp = self.wpt = None      # packed images kept fresh by a WeightPacker (one launch per forward), if any
        # 3x3 / stride 1 / pad 1 over <= 16 NCHW planes (the ResNet stem): the input is re-laid once as NHWC with 16
        # zero-padded channels and the layer runs the strip / row-streaming kernels of layer1 (the packed weight image of an
        # NCHW layer already is one 16-channel chunk with zero rows, so it is shared)
        self.nhwc16 = None
        if in_nchw and k == 3 and stride == 1 and pad == 1 and dil == 1 and cin <= 16:
            self.nhwc16 = _Nhwc16Spec(self)

    def packed(self, w, transpose=False):
        cached = self.wpt if transpose else self.wp
        return cached if cached is not None else ops.pack_weights(w, transpose=transpose)

    def fwd(self, x, w, bias=None):
        return ops.conv2d_fwd(x, self.packed(w), self.cout, self.k, self.k, self.stride, self.pad, self.dil,
                              bias=bias, in_nchw=self.in_nchw)

    def fwd_conv(self, x, w, bn, training, bias=None, in_affine=None, fuse_stats=True, finalize=True):
        """The convolution in front of a BatchNorm: (c, stats) with stats = (save_mean, save_invstd, scale, shift) when the
        conv launch produced the training-mode statistics itself (strip kernel epilogue), else None.  in_affine=(scale,
        shift): the input is relu(x*scale + shift), applied while the kernel loads x (the producer's BatchNorm + ReLU
        never materialised).  finalize=False: stats is the un-merged (records, count) pair instead -- for a consumer that
        merges them itself (ops.bn_finalize_apply)."""
        wp = self.packed(w)
        aff = {} if in_affine is None else dict(in_scale=in_affine[0], in_shift=in_affine[1], relu_in=True)
        if training and not self.in_nchw and fuse_stats:
            n16 = (self.cout + 15) // 16 * 16
            buf = ops.bn_partial_buffer(x.device, n16)
            c, parts, out4 = ops.conv2d_fwd(x, wp, self.cout, self.k, self.k, self.stride, self.pad, self.dil, bias=bias,
                                            bn_partial=buf, bn=_bn_args(bn, training), finalize=finalize, **aff)
            if not finalize:
                return c, ((buf, parts) if parts > 0 else None)
            return c, ((out4[0], out4[1], out4[2], out4[3]) if parts > 0 else None)
        return ops.conv2d_fwd(x, wp, self.cout, self.k, self.k, self.stride, self.pad, self.dil, bias=bias,
                              in_nchw=self.in_nchw, **aff), None

    def fwd_bn(self, x, w, bn, training, bias=None, in_affine=None):
        """Convolution followed by the statistics of its BatchNorm: (c, save_mean, save_invstd, scale, shift); a separate
        statistics pass runs over c when the conv launch did not emit them."""
        c, stats = self.fwd_conv(x, w, bn, training, bias=bias, in_affine=in_affine)
        return (c,) + tuple(stats if stats is not None else _bn_fwd_stats(c, bn, training))

    def fwd_bn_apply(self, x, w, bn, training, in_affine=None, res=None, res_affine=None, sign_bits=False, bias=None):
        """fwd_bn followed by out = relu(bn(c) + res [* res_scale + res_shift]): (c, save_mean, save_invstd, scale, shift, out);
        with sign_bits the last item is (out, bits): see ops.bn_apply.  In training mode the statistics records (the
        convolution epilogue's, or a partial pass over c) are merged INSIDE the apply launch (ops.bn_finalize_apply): no
        finalize launch between the two."""
        rs, rh = res_affine if res_affine is not None else (None, None)
        if training and ops.FUSED_FINALIZE:
            c, rec = self.fwd_conv(x, w, bn, training, bias=bias, in_affine=in_affine, finalize=False)
            rec = rec if rec is not None else ops.bn_stats_partial(c)
            (m, i, sc, sh), out = ops.bn_finalize_apply(rec[0], rec[1], _bn_args(bn, training), c, relu=True, res=res,
                                                        res_scale=rs, res_shift=rh, sign_bits=sign_bits)
            return c, m, i, sc, sh, out
        c, m, i, sc, sh = self.fwd_bn(x, w, bn, training, bias=bias, in_affine=in_affine)
        return c, m, i, sc, sh, ops.bn_apply(c, sc, sh, relu=True, res=res, res_scale=rs, res_shift=rh, sign_bits=sign_bits)

    def wgrad(self, x, dy, in_affine=None, out=None):
        aff = {} if in_affine is None else dict(in_scale=in_affine[0], in_shift=in_affine[1], relu_in=True)
        # out given = written in place into the optimizer's gradient bucket: a leaf of the backward pass, nothing reads it
        # before the optimizer, so its slab reduction joins the one launch of ops.wgrad_flush()
        call = lambda: ops.conv2d_wgrad(x, dy, self.cout, self.k, self.k, self.stride, self.pad, self.dil,
                                        in_nchw=self.in_nchw, out=out, defer=out is not None, **aff)
        if out is None:
            return call()
        return ops.side_stream_run(call, (x, dy) + (tuple(in_affine) if in_affine is not None else ()), x.device)

    def dgrad(self, dy, w, hw, **kw):
        return ops.conv2d_dgrad(dy, self.packed(w, transpose=True), self.cin, hw[0], hw[1], self.k, self.k,
                                self.stride, self.pad, self.dil, **kw)


class _Nhwc16Spec(ConvSpec):
    """The same convolution seen over the re-laid NHWC-16 input (see ConvSpec.__init__)."""

    def __init__(self, parent):
        self.cin, self.cout, self.k, self.stride, self.pad, self.dil, self.in_nchw = 16, parent.cout, 3, 1, 1, 1, False
        self.parent, self.nhwc16, self.wpt = parent, None, None

    @property
    def wp(self):
        return self.parent.wp

    def packed(self, w, transpose=False):
        return self.parent.packed(w, transpose)

    def wgrad(self, x, dy, in_affine=None, out=None):
        # only the real input planes are written ([cout, cin, 3, 3] = conv1.weight's shape): the zero-padded ones drop out
        call = lambda: ops.conv2d_wgrad(x, dy, self.cout, 3, 3, 1, 1, 1, out=out, c_real=self.parent.cin, defer=out is not None)
        return call() if out is None else ops.side_stream_run(call, (x, dy), x.device)


class WeightPacker:
    """Keeps the packed (forward) and transposed-packed (dgrad) images of every convolution weight of a model up to date
    with ONE launch per forward pass instead of two small launches per convolution."""

    def __init__(self, pairs):
        self.pairs = list(pairs)           # (ConvSpec, weight Parameter, needs_dgrad)
        self._key = None

    def _build(self):
        rows, dev = [], self.pairs[0][1].device
        L = ops._lib.lib()
        for spec, w, need_t in self.pairs:
            o, i, kh, kw = w.shape
            for transpose in ((False, True) if need_t else (False,)):
                n_out, k_in = (i, o) if transpose else (o, i)
                n = L.dam_conv_packed_weight_count(n_out, k_in, kh, kw)
                buf = torch.empty(n, dtype=torch.float32, device=dev)
                if transpose:
                    spec.wpt = buf
                else:
                    spec.wp = buf
                rows.append([w.data_ptr(), buf.data_ptr(), o, i, kh, kw, int(transpose), n])
        self.desc = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.n, self.max_total = len(rows), max(r[7] for r in rows)

    def pack_all(self):
        key = tuple(w.data_ptr() for _, w, _ in self.pairs)
        if key != self._key:               # parameters moved (.to(device), optimizer flattening): rebuild the table
            self._build()
            self._key = key
        ops.pack_weights_multi(self.desc, self.n, self.max_total)


def _bn_args(bn, training):
    """(gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps) of a batch-statistics pass."""
    mom = bn.momentum if bn.momentum is not None else 0.1
    track = training and bn.track_running_stats
    return (bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None,
            bn.num_batches_tracked if track else None, mom, bn.eps)


def _bn_fwd_stats(c, bn, training):
    """(save_mean, save_invstd, scale, shift) for conv output c under nn.BatchNorm2d semantics."""
    if training or bn.running_mean is None:
        return ops.bn_stats(c, *_bn_args(bn, training))
    return ops.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


def _hw(x, in_nchw):
    return (x.shape[2], x.shape[3]) if in_nchw else (x.shape[1], x.shape[2])


class FoldedConvBn:
    """Inference form of conv -> BatchNorm2d(eval): the running statistics are constants, so the BatchNorm's scale goes
    into the weights and its shift (plus the scaled convolution bias) into the bias of ONE convolution launch whose epilogue
    also adds the shortcut and applies the ReLU (`ops.conv2d_fwd(res=, relu_out=)`) -- no BatchNorm kernel, no weight
    re-packing per forward.  Built lazily, rebuilt when a tensor it was folded from has changed: torch's version counters
    for torch-side writes, ops.PARAM_EPOCH for the in-place updates of this library's own kernels (Adam, running statistics)."""

    def __init__(self, spec, conv, bn):
        self.spec, self.conv, self.bn = spec, conv, bn
        self._key = self.wp = self.bias = None

    def foldable(self):
        return self.bn.running_mean is not None and self.bn.running_var is not None

    def _tensors(self):
        bn, conv = self.bn, self.conv
        return [t for t in (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var) if t is not None]

    def get(self):
        key = (ops.PARAM_EPOCH,) + tuple((t.data_ptr(), t._version) for t in self._tensors())
        if key != self._key:
            bn, conv = self.bn, self.conv
            with torch.no_grad():
                scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
                shift = bn.bias - bn.running_mean * scale
                if conv.bias is not None:
                    shift = shift + conv.bias * scale
                n16 = (self.spec.cout + 15) // 16 * 16
                bias = torch.zeros(n16, dtype=torch.float32, device=scale.device)
                bias[:self.spec.cout] = shift
                self.wp = ops.pack_weights(conv.weight * scale.view(-1, 1, 1, 1))
                self.bias = bias
            self._key = key
        return self.wp, self.bias

    def fwd(self, x, res=None, relu=True):
        wp, bias = self.get()
        sp = self.spec
        if sp.nhwc16 is not None:            # the stem: NCHW planes re-laid once, then the 16-channel kernels
            sp, x = sp.nhwc16, ops.nchw_to_nhwc16(x)
        return ops.conv2d_fwd(x, wp, sp.cout, sp.k, sp.k, sp.stride, sp.pad, sp.dil, bias=bias, in_nchw=sp.in_nchw,
                              res=res, relu_out=relu)


def inference_mode(module):
    """True when a conv-BatchNorm block may take its folded one-launch form: eval mode and no autograd graph wanted."""
    return not module.training and not torch.is_grad_enabled()


UNIT_SEED = '_dam_unit_seed'        # attribute of a gradient seed tensor that is known to hold 1.0 (engine.TrainStep)


class _UpstreamBn:
    """What the consumer of a relu(bn(c)) activation needs to take that BatchNorm's two backward sums in its own data-gradient
    epilogue (ops.conv2d_dgrad(bn_bwd=)), and the slot where it leaves them for the producer's backward.  Travels as an attribute
    of the activation tensor: ConvBnReluFn.forward attaches it, an identity BasicBlock that receives the tensor picks it up."""
    __slots__ = ('c', 'mean', 'invstd', 'scale', 'shift', 'bits', 'partials', 'dx_ptr', 'dx_version')

    def __init__(self, c, mean, invstd, scale, shift, bits=None):
        # ReLU mask of the activation: fma(c, scale, shift) > 0 (plain relu(bn(c))) or the sign bytes of relu(bn(c) + shortcut)
        self.c, self.mean, self.invstd, self.scale, self.shift, self.bits = c, mean, invstd, scale, shift, bits
        self.partials, self.dx_ptr, self.dx_version = None, None, None

    def request(self):
        return (self.c, self.mean, self.invstd, self.scale, self.shift) + ((self.bits,) if self.bits is not None else ())

    def offer(self, sums, dx):
        """Consumer side: `sums` were taken over exactly the values `dx` holds now."""
        self.partials, self.dx_ptr, self.dx_version = sums, dx.data_ptr(), dx._version

    def take(self, grad):
        """The sums the consumer left, if they belong to this gradient tensor AS IT IS NOW: same storage and no in-place
        write since (autograd accumulates a second consumer's gradient, or a hook's edit, in place into the first-arrived
        buffer -- same address, bumped version counter: the sums are stale then and the separate pass runs).  The slot is
        cleared either way."""
        ok = self.partials is not None and self.dx_ptr == grad.data_ptr() and self.dx_version == grad._version
        sums = self.partials if ok else None
        self.partials = None
        if sums is not None:
            _UpstreamBn.hits += 1
        return sums

    hits = 0       # how often a producer's backward used sums left by its consumer (diagnostic; the tests read it)


class ConvBnReluFn(torch.autograd.Function):
    """a = relu(bn(conv(x) [+ bias]))  -- ResNet stem (models/model_resnet.py:97) and
    ConvBlock2d (models/model_scalar_1s.py:179-190 without the dropout)."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, spec, bn, training):
        if spec.nhwc16 is not None and not ctx.needs_input_grad[0]:
            spec, x = spec.nhwc16, ops.nchw_to_nhwc16(x)
        c, mean, invstd, scale, shift, a = spec.fwd_bn_apply(x, w.detach(), bn, training,
                                                             bias=None if bias is None else bias.detach())
        ctx.save_for_backward(x, w, c, gamma, mean, invstd, scale, shift)
        ctx.up = a._dam_upstream = _UpstreamBn(c, mean, invstd, scale, shift) if ops.DGRAD_BN_SUMS else None
        ctx.spec, ctx.training, ctx.has_bias = spec, training, bias is not None
        ctx.slots = (_slot(w), _slot(bias), _slot(gamma), _slot(beta))
        return a

    @staticmethod
    def backward(ctx, da):
        x, w, c, gamma, mean, invstd, scale, shift = ctx.saved_tensors
        spec = ctx.spec
        # relu mask recomputed from c and the forward's affine: the saved activation is not read (nor kept by this node)
        sw, sb, sg, sbt = ctx.slots
        da = da.contiguous()
        # the consumer (an identity block's conv1 data gradient) may have left this BatchNorm's two sums beside the gradient
        sums = ctx.up.take(da) if ctx.up is not None else None
        dc, dgamma, dbeta = ops.bn_backward(da, None, c, gamma, mean, invstd, ctx.training,
                                            mask_affine=(scale, shift), dgamma=sg, dbeta=sbt, partials=sums)
        dw = spec.wgrad(x, dc, out=None if sw is None else sw.view(w.shape))
        dbias = ops.channel_sum(dc, spec.cout, out=sb) if ctx.has_bias else None
        dx = spec.dgrad(dc, w, _hw(x, spec.in_nchw)) if ctx.needs_input_grad[0] else None
        return (dx, None if sw is not None else dw, None if sb is not None else dbias, None if sg is not None else dgamma,
                None if sbt is not None else dbeta, None, None, None)


class DropoutFn(torch.autograd.Function):
    """nn.Dropout(p) of ConvBlock2d as a HIP kernel; the mask is regenerated in backward from the saved call offset."""

    @staticmethod
    def forward(ctx, x, p):
        snap = ops.dropout_tick(x.device, x.numel())
        ctx.p, ctx.seed = p, torch.initial_seed()
        ctx.save_for_backward(snap)
        return ops.dropout_apply(x, p, ctx.seed, snap)

    @staticmethod
    def backward(ctx, dy):
        (snap,) = ctx.saved_tensors
        return ops.dropout_apply(dy, ctx.p, ctx.seed, snap), None


class BasicBlockFn(torch.autograd.Function):
    """models/model_resnet.py:23-28: relu(bn2(conv2(relu(bn1(conv1(x))))) + shortcut(x))."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, wsc, gsc, bsc, blk, training):
        # a1 = relu(bn1(c1)) is never written: conv2 (and later its weight gradient and bn1's backward) read c1 and apply the
        # fused affine themselves
        if wsc is not None:
            bn_sc = blk.shortcut[1]
            # (no statistics from these two launches: the pair pass below is one launch sequence for both tensors, a
            # fused epilogue on conv1 alone would leave the shortcut's statistics a pass of their own)
            one = None
            if (ops.CONV_S2_PAIR and training and blk.spec1.k == 3 and blk.spec1.stride == 2 and blk.spec1.pad == 1
                    and blk.spec1.dil == 1 and blk.spec_sc.k == 1 and blk.spec_sc.stride == 2 and blk.spec_sc.pad == 0
                    and not blk.spec1.in_nchw):
                # both convolutions and both statistics passes from one read of x (thin stages; None: not such a layer)
                one = ops.conv_s2_pair_fwd(x, blk.spec1.packed(w1.detach()), blk.spec_sc.packed(wsc.detach()), blk.spec1.cout)
            if one is not None:
                c1, cs, (p1, p2, parts) = one
                st1, sts = ops.bn_finalize_pair(p1, p2, parts, _bn_args(blk.bn1, True), _bn_args(bn_sc, True))
            else:
                c1, st1 = blk.spec1.fwd_conv(x, w1.detach(), blk.bn1, training, fuse_stats=False)
                cs, sts = blk.spec_sc.fwd_conv(x, wsc.detach(), bn_sc, training, fuse_stats=False)
            if st1 is None and sts is None and training and c1.shape == cs.shape:
                # two independent BatchNorms over tensors of one shape, ready together: one statistics pass for both
                st1, sts = ops.bn_stats_pair(c1, _bn_args(blk.bn1, True), cs, _bn_args(bn_sc, True))
            m1, i1, sc1, sh1 = st1 if st1 is not None else _bn_fwd_stats(c1, blk.bn1, training)
            ms, is_, scs, shs = sts if sts is not None else _bn_fwd_stats(cs, bn_sc, training)
            # the backward BatchNorm passes want only the SIGN of the block output (the ReLU mask): bn_apply leaves it as one
            # byte per channel quad, 1/16 of the bytes those passes would read from `out`
            c2, m2, i2, sc2, sh2, (out, bits) = blk.spec2.fwd_bn_apply(c1, w2.detach(), blk.bn2, training, in_affine=(sc1, sh1),
                                                                       res=cs, res_affine=(scs, shs), sign_bits=True)
            ctx.save_for_backward(x, w1, g1, w2, g2, c1, c2, out, m1, i1, m2, i2, sc1, sh1, bits, wsc, gsc, cs, ms, is_)
        else:
            c1, m1, i1, sc1, sh1 = blk.spec1.fwd_bn(x, w1.detach(), blk.bn1, training)
            c2, m2, i2, sc2, sh2, (out, bits) = blk.spec2.fwd_bn_apply(c1, w2.detach(), blk.bn2, training, in_affine=(sc1, sh1),
                                                                       res=x, sign_bits=True)
            ctx.save_for_backward(x, w1, g1, w2, g2, c1, c2, out, m1, i1, m2, i2, sc1, sh1, bits)
        # this block's bn2 can be the upstream BatchNorm of the next identity block in turn (mask = the sign bytes of `out`)
        ctx.up_self = None
        if wsc is None and ops.DGRAD_BN_SUMS:
            ctx.up_self = out._dam_upstream = _UpstreamBn(c2, m2, i2, None, None, bits)
        ctx.blk, ctx.training, ctx.has_sc = blk, training, wsc is not None
        # an identity block's conv1 data gradient + shortcut IS the gradient of its input: if that input is a relu(bn(c)) whose
        # producer left its record (_UpstreamBn: the stem in front of the first block), backward takes that BatchNorm's sums too
        # (a down-sampling block's one-launch data gradient does the same for the residual block in front of it: mask as sign bytes)
        ctx.upstream = getattr(x, '_dam_upstream', None)
        if wsc is not None and (ctx.upstream is None or ctx.upstream.bits is None or ctx.upstream.scale is not None):
            ctx.upstream = None
        ctx.slots = tuple(_slot(p) for p in (w1, g1, b1, w2, g2, b2, wsc, gsc, bsc))
        return out

    @staticmethod
    def backward(ctx, dout):
        blk, tr = ctx.blk, ctx.training
        if ctx.has_sc:
            x, w1, g1, w2, g2, c1, c2, out, m1, i1, m2, i2, sc1, sh1, bits, wsc, gsc, cs, ms, is_ = ctx.saved_tensors
        else:
            x, w1, g1, w2, g2, c1, c2, out, m1, i1, m2, i2, sc1, sh1, bits = ctx.saved_tensors
        dout = dout.contiguous()
        hw = (x.shape[1], x.shape[2])
        s_w1, s_g1, s_b1, s_w2, s_g2, s_b2, s_ws, s_gs, s_bs = ctx.slots
        keep = lambda grad, slot: None if slot is not None else grad      # slotted gradients are already in place
        wview = lambda slot, w: None if slot is None else slot.view(w.shape)
        if ctx.has_sc:      # bn2 and the shortcut's BatchNorm see the same dout through the same mask: one pass for both
            (dc2, dg2, db2), (dcs, dgs, dbs) = ops.bn_backward_pair(dout, None, (c2, g2, m2, i2, s_g2, s_b2),
                                                                    (cs, gsc, ms, is_, s_gs, s_bs), tr, mask_bits=bits)
        else:
            sums2 = ctx.up_self.take(dout) if ctx.up_self is not None else None      # left by the next block's data gradient
            dc2, dg2, db2 = ops.bn_backward(dout, None, c2, g2, m2, i2, tr, dgamma=s_g2, dbeta=s_b2, mask_bits=bits, partials=sums2)
        dw2 = blk.spec2.wgrad(c1, dc2, in_affine=(sc1, sh1), out=wview(s_w2, w2))
        # conv2's data gradient IS the gradient reaching relu(bn1(c1)): where the kernel can, bn1's two backward sums come out
        # of its epilogue (sums = (records, count)) and bn_backward only finalizes and applies
        sums = None
        if ops.DGRAD_BN_SUMS:
            da1, sums = blk.spec2.dgrad(dc2, w2, (c1.shape[1], c1.shape[2]), bn_bwd=(c1, m1, i1, sc1, sh1))
        else:
            da1 = blk.spec2.dgrad(dc2, w2, (c1.shape[1], c1.shape[2]))
        dc1, dg1, db1 = ops.bn_backward(da1, None, c1, g1, m1, i1, tr, mask_affine=(sc1, sh1),   # mask = (bn1(c1) > 0)
                                        dgamma=s_g1, dbeta=s_b1, partials=sums)
        dw1 = blk.spec1.wgrad(x, dc1, out=wview(s_w1, w1))
        first = (keep(dw1, s_w1), keep(dg1, s_g1), keep(db1, s_b1), keep(dw2, s_w2), keep(dg2, s_g2), keep(db2, s_b2))
        if ctx.has_sc:
            dws = blk.spec_sc.wgrad(x, dcs, out=wview(s_ws, wsc))
            if (ops.PAIR_1X1 and blk.spec1.k == 3 and blk.spec1.stride == 2 and blk.spec1.pad == 1 and blk.spec1.dil == 1
                    and blk.spec_sc.k == 1 and blk.spec_sc.stride == 2 and blk.spec_sc.pad == 0):      # (the forward's gate)
                # the shortcut's whole data gradient and the centre tap of conv1's land on the same (even, even) pixels: one launch
                up = ctx.upstream
                pair = (dcs, blk.spec_sc.packed(wsc, transpose=True))
                if up is not None and ops.DGRAD_BN_SUMS and ops.DGRAD_S2 and up.c.shape == x.shape:
                    dx, sums = blk.spec1.dgrad(dc1, w1, hw, pair_1x1=pair, bn_bwd=up.request())
                    if sums is not None:
                        up.offer(sums, dx)
                else:
                    dx = blk.spec1.dgrad(dc1, w1, hw, pair_1x1=pair)
            else:
                dx = blk.spec1.dgrad(dc1, w1, hw)
                blk.spec_sc.dgrad(dcs, wsc, hw, accumulate_into=dx)
            return (dx,) + first + (keep(dws, s_ws), keep(dgs, s_gs), keep(dbs, s_bs), None, None)
        up = ctx.upstream
        if up is not None and ops.DGRAD_BN_SUMS and up.c.shape == x.shape:
            dx, sums = blk.spec1.dgrad(dc1, w1, hw, res=dout, res_mask=out, res_mask_bits=bits, bn_bwd=up.request())
            up.offer(sums, dx)
        else:
            dx = blk.spec1.dgrad(dc1, w1, hw, res=dout, res_mask=out)      # + dout * (out > 0): identity shortcut
        return (dx,) + first + (None, None, None, None, None)


class HeadsFn(torch.autograd.Function):
    """All S heads + the gain-weighted sum (models/model_resnet.py:108-126)."""

    @staticmethod
    def forward(ctx, trunk, x, cw, cb, fw, fb):
        h, g = ops.heads_fwd(trunk, cw.detach(), cb.detach(), fw.detach(), fb.detach())
        masked = ops.masksum_fwd(x, g)
        ctx.save_for_backward(trunk, x, h, cw, fw)
        ctx.set_materialize_grads(False)
        ctx.slots = (_slot(cw), _slot(cb), _slot(fw), _slot(fb))
        return masked, g

    @staticmethod
    def backward(ctx, dmasked, dg_out):
        trunk, x, h, cw, fw = ctx.saved_tensors
        dg = None
        if dmasked is not None:
            dg = ops.masksum_bwd(dmasked.contiguous(), x)
        if dg_out is not None:
            dg = dg_out.contiguous() if dg is None else dg + dg_out
        if dg is None:
            return None, None, None, None, None, None
        dtrunk, dcw, dcb, dfw, dfb = ops.heads_bwd(dg, h, trunk, cw, fw, outs=ctx.slots)
        sl = ctx.slots
        return (dtrunk, None, None if sl[0] is not None else dcw, None if sl[1] is not None else dcb,
                None if sl[2] is not None else dfw, None if sl[3] is not None else dfb)


class HeadsMseFn(torch.autograd.Function):
    """Heads + masked sum + MSELoss(masked, gt) in one Function: the loss and d loss/d gains come out of a
    single pass over x (dam_masksum_mse_f32).  Returns (loss, masked, gains); only loss carries gradient."""

    @staticmethod
    def forward(ctx, trunk, x, gt, cw, cb, fw, fb):
        h, g = ops.heads_fwd(trunk, cw.detach(), cb.detach(), fw.detach(), fb.detach())
        masked, loss, dg = ops.masksum_mse(x, g, gt.contiguous())
        ctx.save_for_backward(trunk, h, cw, fw, dg)
        ctx.mark_non_differentiable(masked, g)
        ctx.set_materialize_grads(False)      # (two zero-fill launches per step for gradients nobody reads, one of them 4 MB)
        ctx.slots = (_slot(cw), _slot(cb), _slot(fw), _slot(fb))
        return loss.reshape(()), masked, g

    @staticmethod
    def backward(ctx, dloss, _dm, _dg):
        trunk, h, cw, fw, dg = ctx.saved_tensors
        if dloss is None:
            dloss = torch.zeros((), dtype=torch.float32, device=dg.device)
        # engine.TrainStep seeds the backward pass with its own constant 1 (UNIT_SEED marks it): no multiply launch then
        seed = dg if getattr(dloss, UNIT_SEED, False) else dg * dloss
        dtrunk, dcw, dcb, dfw, dfb = ops.heads_bwd(seed, h, trunk, cw, fw, outs=ctx.slots)
        sl = ctx.slots
        return (dtrunk, None, None, None if sl[0] is not None else dcw, None if sl[1] is not None else dcb,
                None if sl[2] is not None else dfw, None if sl[3] is not None else dfb)


class BasicBlock(nn.Module):
    """Parameter container with the reference's names (models/model_resnet.py:6-21); forward = BasicBlockFn."""
    expansion = 1

    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_channels != self.expansion * out_channels:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_channels, self.expansion * out_channels, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(self.expansion * out_channels))
        self.spec1 = ConvSpec(in_channels, out_channels, 3, stride, 1)
        self.spec2 = ConvSpec(out_channels, out_channels, 3, 1, 1)
        self.spec_sc = ConvSpec(in_channels, out_channels, 1, stride, 0) if len(self.shortcut) else None

    def _folded(self):
        f = getattr(self, '_fold', None)
        if f is None:
            f = self._fold = (FoldedConvBn(self.spec1, self.conv1, self.bn1), FoldedConvBn(self.spec2, self.conv2, self.bn2),
                              FoldedConvBn(self.spec_sc, self.shortcut[0], self.shortcut[1]) if self.spec_sc is not None else None)
        return f

    def forward(self, x):
        if inference_mode(self):
            f1, f2, fs = self._folded()
            if f1.foldable() and f2.foldable() and (fs is None or fs.foldable()):
                # relu(bn1(conv1 x)); shortcut; relu(bn2(conv2 .) + shortcut): three launches (two without a shortcut conv)
                a1 = f1.fwd(x)
                return f2.fwd(a1, res=x if fs is None else fs.fwd(x, relu=False))
        if self.spec_sc is not None:
            sc, sbn = self.shortcut[0], self.shortcut[1]
            return BasicBlockFn.apply(x, self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight,
                                      self.bn2.weight, self.bn2.bias, sc.weight, sbn.weight, sbn.bias, self, self.training)
        return BasicBlockFn.apply(x, self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight,
                                  self.bn2.weight, self.bn2.bias, None, None, None, self, self.training)


class ConvBlock2d(nn.Module):
    """models/model_scalar_1s.py:151-190: Conv2D(valid, bias) -> BatchNorm(momentum .9, eps 1e-3) -> ReLU -> Dropout
    (dropout only while self.training, as in the reference)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, dropout_p=-1.0, in_nchw=False):
        super().__init__()
        self.add_padding = None
        self.conv = nn.Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                              stride=stride, padding=0, dilation=dilation)
        self.batch_norm = nn.BatchNorm2d(num_features=out_channels, momentum=0.90, eps=0.001)
        self.activation = nn.ReLU()
        self.dropout = nn.Dropout(dropout_p) if dropout_p != -1 else None
        self.spec = ConvSpec(in_channels, out_channels, kernel_size, stride, 0, dilation, in_nchw=in_nchw)

    def forward(self, x):
        if inference_mode(self):
            f = getattr(self, '_fold', None)
            if f is None:
                f = self._fold = FoldedConvBn(self.spec, self.conv, self.batch_norm)
            if f.foldable():
                return f.fwd(x)                  # dropout is the identity in eval mode
        out = ConvBnReluFn.apply(x, self.conv.weight, self.conv.bias, self.batch_norm.weight, self.batch_norm.bias,
                                 self.spec, self.batch_norm, self.training)
        if self.training and self.dropout:
            out = DropoutFn.apply(out, self.dropout.p) if self.dropout.p > 0 else out
        return out


class GainHeads(nn.Module):
    """The S (conv1x1 -> ReLU -> Linear) heads, stored stacked ([S,C], [S], [S,P], [S]) so that one kernel serves all
    stems; state_dict()/load_state_dict() speak the reference's per-stem keys conv_head{i}.weight [1,C,1,1],
    conv_head{i}.bias [1], fc_head{i}.weight [1,P], fc_head{i}.bias [1] (models/model_resnet.py:75-85)."""

    def __init__(self, channels, n_stems, flattened_dim):
        super().__init__()
        self.n_stems, self.channels, self.flattened_dim = n_stems, channels, flattened_dim
        cw, cb, fw, fb = [], [], [], []
        for _ in range(n_stems):   # same construction order (and RNG draws) as the reference
            conv, fc = nn.Conv2d(channels, 1, kernel_size=(1, 1)), nn.Linear(flattened_dim, 1)
            cw.append(conv.weight.detach().view(1, channels)), cb.append(conv.bias.detach())
            fw.append(fc.weight.detach()), fb.append(fc.bias.detach())
        self.conv_w, self.conv_b = nn.Parameter(torch.cat(cw)), nn.Parameter(torch.cat(cb))
        self.fc_w, self.fc_b = nn.Parameter(torch.cat(fw)), nn.Parameter(torch.cat(fb))

    # --- reference-named views of the stacked parameters
    def reference_items(self):
        for i in range(self.n_stems):
            yield 'conv_head%d.weight' % (i + 1), self.conv_w[i].view(1, self.channels, 1, 1)
            yield 'conv_head%d.bias' % (i + 1), self.conv_b[i:i + 1]
            yield 'fc_head%d.weight' % (i + 1), self.fc_w[i:i + 1]
            yield 'fc_head%d.bias' % (i + 1), self.fc_b[i:i + 1]

    def forward(self, trunk, x):
        return HeadsFn.apply(trunk, x, self.conv_w, self.conv_b, self.fc_w, self.fc_b)

    def forward_mse(self, trunk, x, gt):
        return HeadsMseFn.apply(trunk, x, gt, self.conv_w, self.conv_b, self.fc_w, self.fc_b)


class MixingNet(nn.Module):
    """Common shell of the three models: a trunk producing NHWC features, GainHeads registered as `_heads`
    with the reference's flat key names in state_dict, forward(x) -> (masked, (g_1..g_S))."""

    def _init_heads(self, channels, n_stems, flattened_dim):
        self._heads = GainHeads(channels, n_stems, flattened_dim)
        self.n_stems = n_stems
        self._register_state_dict_hook(MixingNet._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    @staticmethod
    def _sd_hook(module, state_dict, prefix, local_metadata):
        for k in ('conv_w', 'conv_b', 'fc_w', 'fc_b'):
            state_dict.pop(prefix + '_heads.' + k, None)
        for name, view in module._heads.reference_items():
            state_dict[prefix + name] = view.detach()
        return state_dict

    def _load_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        h = self._heads
        parts = {}
        for i in range(h.n_stems):
            for kind in ('conv_head%d.weight', 'conv_head%d.bias', 'fc_head%d.weight', 'fc_head%d.bias'):
                key = prefix + kind % (i + 1)
                if key in state_dict:
                    parts.setdefault(kind, []).append(state_dict.pop(key))
        if not parts:
            return
        if any(len(v) != h.n_stems for v in parts.values()) or len(parts) != 4:
            error_msgs.append('incomplete per-stem head parameters in state_dict')
            return
        dev = h.conv_w.device
        state_dict[prefix + '_heads.conv_w'] = torch.cat([t.reshape(1, h.channels) for t in parts['conv_head%d.weight']]).to(dev)
        state_dict[prefix + '_heads.conv_b'] = torch.cat([t.reshape(1) for t in parts['conv_head%d.bias']]).to(dev)
        state_dict[prefix + '_heads.fc_w'] = torch.cat([t.reshape(1, -1) for t in parts['fc_head%d.weight']]).to(dev)
        state_dict[prefix + '_heads.fc_b'] = torch.cat([t.reshape(1) for t in parts['fc_head%d.bias']]).to(dev)

    def trunk(self, x, tap=None):
        """x [B,S,F,T] -> NHWC trunk features.  tap (optional list): the model appends the activation at its
        data-parallel bucket boundary (see ddp_late_parameters) to it."""
        raise NotImplementedError

    def ddp_late_parameters(self):
        """Parameters of the layers BEHIND the bucket boundary (the deep, parameter-heavy end of the trunk and the
        heads), whose gradients backward produces first: the step engine all-reduces them while the rest of backward
        runs.  They are the tail of model.parameters()."""
        raise NotImplementedError

    @staticmethod
    def _check_input(x):
        if x.dim() != 4:
            raise ValueError('expected x of shape [B, S, F, T]')
        if not x.is_cuda:
            raise RuntimeError('deep_audio_mixer_amd models run on the GPU only (got a %s tensor); there is no CPU '
                               'fallback' % x.device)
        if x.dtype != torch.float32:
            # the reference raises too ("expected scalar type Double but found Float")
            raise RuntimeError('expected scalar type Float but found %s' % str(x.dtype).replace('torch.', '').capitalize())
        return x.contiguous()

    def conv_pairs(self):
        """(ConvSpec, weight, needs_dgrad) of every convolution of the trunk; overridden by the models."""
        return []

    def _pack_weights(self):
        if self.training:
            ops.params_changed()        # this forward's statistics passes move the BatchNorm running buffers
        if getattr(self, '_packer', None) is None:
            pairs = self.conv_pairs()
            self._packer = WeightPacker(pairs) if pairs else False
        if self._packer:
            self._packer.pack_all()

    def forward(self, x):
        x = self._check_input(x)
        self._pack_weights()
        masked, g = self._heads(self.trunk(x), x)
        return masked, tuple(g[:, s:s + 1] for s in range(self.n_stems))

    def predict_gains(self, x):
        """The gain heads alone, [B, S] raw outputs, without autograd and without the masked sum (full-song inference
        only wants the gains: inference_utils.py:123 discards the first output of model(x))."""
        x = self._check_input(x)
        self._pack_weights()
        with torch.no_grad():
            h = self._heads
            return ops.heads_fwd(self.trunk(x), h.conv_w, h.conv_b, h.fc_w, h.fc_b)[1]

    def forward_mse(self, x, gt, tap=None):
        """Fused fast path for criterion == nn.MSELoss(): returns (loss, masked, gains tuple); masked and the
        gains are detached outputs, loss carries the gradient.  tap: see trunk()."""
        x = self._check_input(x)
        self._pack_weights()
        loss, masked, g = self._heads.forward_mse(self.trunk(x) if tap is None else self.trunk(x, tap), x, gt)
        return loss, masked, tuple(g[:, s:s + 1] for s in range(self.n_stems))
