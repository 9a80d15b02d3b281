"""One full training step of the hot path as a replayable unit:

    interleaved PCM (stems + mix, resident in HBM)
      -> dam_stft_logmag_f32 (features x [B,S,F,T], target gt [B,F,T])
      -> model.forward_mse (conv/BN/ReLU trunk, heads, fused masked-sum + MSE)
      -> backward (BN backward, dgrad, wgrad, heads)            [model_trainer.py:30-37 of the reference]
      -> flat gradient buckets -> (RCCL all-reduce over the data-parallel group) -> Adam(+L2)

The kernel sequence is static, so after a few eager warm-up steps it is captured into hipGraphs
(torch.cuda.CUDAGraph on the ROCm build) and replayed: one host call per graph instead of ~400 launches.

One rank: a single graph (front-end + forward + backward + bucket + Adam).

N ranks (data parallel): backward is cut at the model's bucket boundary into two graphs and the all-reduces
stay between them, on RCCL's own stream, so that the big bucket travels while the rest of backward runs:

    graph A1: front-end, forward, backward of heads + deep layers  -> bucket 1 (ResNet18: layer5/6 + heads, 85 % of
              the 12.6 MB, ready after the first tenth of backward)
    all-reduce(bucket 1)  [async, overlaps A2]
    graph A2: backward of the shallow layers                        -> bucket 0
    all-reduce(bucket 0)  [async]; wait both
    graph B : Adam over the whole flat buffer (1/world folded in)

The cut uses autograd itself: stage 1 is torch.autograd.grad(loss, late parameters + [boundary activation]), stage 2
torch.autograd.backward(boundary activation, its gradient, inputs=early parameters).
"""
import os

import torch

from . import distributed, features, layers, ops


class TrainStep:
    MAX_ROTATION = 64

    def __init__(self, model, optimizer, n_stems, n_samples=None, channels=2, batch=8, n_fft=2048, hop=1024,
                 use_graph=True, device=None, overlap=True, feature_shape=None, pcm_dtype=torch.float32, track_gains=False,
                 normalize=False, copy_mark=False):
        """feature_shape=(F, T): the step starts from FEATURES instead of PCM (what a DataLoader over the reference's
        Dataset yields, model_trainer.py:31-33): no front-end launch in the step, `load_features(x, gt)` fills the static
        inputs.  pcm_dtype: float32, or int16 / int32 for integer PCM read by the front-end as decoded from the file
        (DAM_PCM_S16 / DAM_PCM_S32, include/dam_hip.h).  track_gains: the front-end multiplies every track by an entry of a
        static [B, S+1] table (data/dataset.py:164-168, 198-199: the augmentation draws; `bind_clips(clips, gain)` fills it,
        ones otherwise).  normalize: the per-frame max-abs normalisation of data/dataset.py:159-160 (off at the reference HEAD).
        copy_mark: the step records `self.copy_mark` (staging.StepMark) where backward crosses the model's bucket boundary
        (ResNet: below layer5) -- the uploader of the NEXT batch waits for it (staging.BatchStager(gate=)): a host-to-device
        copy beside the forward pass slows its latency-bound launches, beside the shallow layers' backward it does not."""
        self.model, self.opt = model, optimizer
        self.device = device or next(model.parameters()).device
        self.n_fft, self.hop, self.batch, self.n_stems = n_fft, hop, batch, n_stems
        dev = self.device
        self.from_features = feature_shape is not None
        if self.from_features:
            f, t = feature_shape
            self.pcm = self.pcm_word = self.pcm_table = self._bound = None
        else:
            f, t = n_fft // 2 + 1, features.num_frames(n_samples, hop)
            # stems and mix of a batch live in ONE buffer ([B, S+1, n, ch], mix last): the front-end is one launch
            self.pcm = torch.zeros((batch, n_stems + 1, n_samples, channels), dtype=pcm_dtype, device=dev)
            # the front-end reads the batch THROUGH this device table (DAM_PCM_ROTATE: {address of the optimizer's step count, n,
            # offset, addr[0..n)} -> addr[(count + offset) % n]): one entry pointing at self.pcm unless bind_clips() re-pointed
            # it at another resident batch, or bind_rotation() filled it with n batches the replays then walk by themselves --
            # a graph replay reads its batch in place, no copy, and with a rotation nothing is launched between the replays
            self.pcm_table = torch.zeros(3 + self.MAX_ROTATION, dtype=torch.int64, device=dev)
            self.pcm_table[:4] = torch.tensor([optimizer._step.data_ptr(), 1, 0, self.pcm.data_ptr()], dtype=torch.int64)
            self.pcm_word = self.pcm_table[3:4]        # (the single-batch form re-points entry 0)
            self._rotation = 1
            self._bound = self.pcm
        self.gain = torch.ones((batch, n_stems + 1), dtype=torch.float32, device=dev) if track_gains and not self.from_features else None
        self.normalize = bool(normalize)
        self.x = torch.empty((batch, n_stems, f, t), dtype=torch.float32, device=dev)
        self.gt = torch.empty((batch, f, t), dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self._unit_seed = torch.ones((), dtype=torch.float32, device=dev)
        setattr(self._unit_seed, layers.UNIT_SEED, True)
        self.use_graph = use_graph
        self._graphs = None
        self._steps_run = 0
        # outputs of the last step's forward (masked [B,F,T], tuple of S gains [B,1]): tensors of the captured graph's pool,
        # kept alive here, so after a replay they hold that step's values (read them before the next step)
        self.masked = self.gains = None
        self.measure_exposed, self._exposed = False, []
        # N-rank diagnostics (bench.py's overlap_ab): overlap_reduce=False starts BOTH all-reduces only after graph A2 has been
        # enqueued (the same three graphs, the exchange not beside backward); measure_a2 brackets graph A2 with HIP events
        self.overlap_reduce, self.measure_a2, self._a2 = True, False, []
        # how the staged step exchanges its buckets (set between steps, same three graphs in every mode):
        #   'overlapped' : both all-reduces asynchronous on RCCL's stream, the big one beside graph A2 (overlap_reduce=False: both
        #                  started only after A2, "serialized");
        #   'inline'     : synchronous collectives -- since torch 2.8 ProcessGroupNCCL launches those on the CURRENT stream: no
        #                  stream waits for an event of the training stream.  Measured with one rank (tools/nccl_sync_probe.py,
        #                  profiles/r05_nccl_sync_probe.txt): the event traffic of an asynchronous all-reduce alone costs the
        #                  captured step 0.10-0.13 ms, a synchronous one 0.02 ms; what inline gives up is the overlap itself
        #                  (12.6 MB over xGMI).  bench.py times both on the node it runs on and keeps the faster.
        self.reduce_mode = 'overlapped'
        self.frames_per_step = batch * n_stems * t          # BASELINE metric unit: stem-spectrogram frames
        self.staged = optimizer.world_size > 1 and overlap
        if self.staged:
            late = model.ddp_late_parameters()
            n_late = len(late)
            if [id(p) for p in optimizer._params[-n_late:]] != [id(p) for p in late]:
                raise ValueError('ddp_late_parameters() must be the tail of the optimizer\'s parameter list')
            optimizer.set_bucket_boundaries([late[0]])
        self._stage = None
        self.copy_mark = None
        if copy_mark:
            from . import staging
            self.copy_mark = staging.StepMark()
        # gradients go straight from the backward kernels into the flat buckets (no .grad tensors, no gather launch).  The
        # binding changes the MODEL (its layers then return no .grad): close() -- or leaving the `with` block -- undoes it
        optimizer.bind_grad_slots()
        self._slots_checked = False

    def close(self):
        """Gives the model back to ordinary autograd: parameters get .grad tensors again (gradient accumulation, clipping,
        another optimizer, torch.save(model)).  The captured graphs stay replayable -- they write the flat bucket directly."""
        self.opt.unbind_grad_slots()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # bit pattern the gradient bucket is poisoned with before the first eager step: a quiet NaN with a payload no arithmetic
    # produces (hardware-generated NaNs are 0x7FC00000; an input's payload would have to be this very one), so a slice that
    # still holds it was not written -- and a legitimately non-finite first gradient is not mistaken for a missing one
    _SENTINEL = 0x7FDA4D17

    def _check_slots_written(self):
        """Overwrite semantics: a bound parameter that backward does not reach would keep the PREVIOUS step's gradient
        (already summed by the all-reduce).  Checked once, on the first eager step, BETWEEN backward and everything that
        consumes the bucket (all-reduce, Adam): the bucket is poisoned before backward and no element of any parameter's
        slice may still hold the poison.  Raising here leaves parameters and moments untouched."""
        left = self.opt._grad.view(torch.int32) == self._SENTINEL
        if bool(left.any()):
            bad = [i for i in range(len(self.opt._params))
                   if bool(left[self.opt._offsets[i]:self.opt._offsets[i + 1]].any())]
            self.opt._grad.zero_()
            self.close()
            raise RuntimeError('TrainStep: backward wrote no gradient for parameter(s) %s of the optimizer -- bound gradient '
                               'slots need every parameter to receive a gradient each step' % bad[:8])
        self._slots_checked = True

    @property
    def stems(self):
        return self.pcm[:, :self.n_stems]

    @property
    def mix(self):
        return self.pcm[:, self.n_stems]

    # -- pieces ---------------------------------------------------------------------------------
    def _front_end(self):
        if self.from_features:
            return
        features.stft_logmag_clips(self.pcm, self.n_fft, self.hop, gain=self.gain, normalize=self.normalize, out_stems=self.x,
                                   out_mix=self.gt, pcm_table=self.pcm_table)

    def _fwd_bwd(self):
        self._front_end()
        self.opt.zero_grad(set_to_none=True)
        if self.copy_mark is None:
            loss, self.masked, self.gains = self.model.forward_mse(self.x, self.gt)
        else:
            tap = []
            loss, self.masked, self.gains = self.model.forward_mse(self.x, self.gt, tap=tap)
            if tap and tap[0].requires_grad and os.environ.get('DAM_COPY_MARK_AT', 'boundary') == 'boundary':
                tap[0].register_hook(self._mark_hook)       # runs when backward reaches the boundary activation
            else:
                self._record_mark()                         # a model without a boundary: behind the forward pass
        loss.backward(self._unit_seed if loss.dim() == 0 else None)      # (no ones_like fill, no seed multiply: layers.UNIT_SEED)
        ops.side_stream_join(self.device)
        ops.wgrad_flush(self.device)                # every weight gradient's slab reduction, one launch
        # the loss tensor itself is the step's output: inside a captured graph its address is fixed (the graph's pool keeps it
        # while this reference lives), so no copy launch into a separate buffer
        self.loss = loss.detach()
        self.opt.gather_grads()

    def _stage1(self):
        """Front-end, forward, backward down to the bucket boundary; fills bucket 1 (the deep layers + heads)."""
        self._front_end()
        self.opt.zero_grad(set_to_none=True)
        tap = []
        loss, self.masked, self.gains = self.model.forward_mse(self.x, self.gt, tap=tap)
        grads, dmid = distributed.backward_late(loss, self.opt.bucket_params(1), tap[0])
        ops.side_stream_join(self.device)
        ops.wgrad_flush(self.device)                # bucket 1 is complete
        self.loss.copy_(loss.detach())
        self.opt.gather_grads(1, grads=grads)
        self._stage = (tap[0], dmid)

    def _record_mark(self):
        """The mark is a timing aid, never a reason to fail a step or a capture: if the runtime refuses the record (the
        event-record node of a capture is added through the capture-info API, include/dam_hip.h), the step goes on without
        it -- waiters then find the last successful record (or none) and simply do not wait."""
        try:
            self.copy_mark.record()
        except RuntimeError as e:
            if not getattr(self, '_mark_warned', False):
                import warnings
                warnings.warn('TrainStep: the step mark could not be recorded (%s); uploads are not timed by it' % e, RuntimeWarning)
                self._mark_warned = True

    def _mark_hook(self, grad):
        self._record_mark()
        return None

    def _stage2(self):
        """Backward from the boundary activation to the input; fills bucket 0."""
        mid, dmid = self._stage
        self._stage = None
        if self.copy_mark is not None:
            self._record_mark()
        distributed.backward_early(mid, dmid, self.opt.bucket_params(0))
        ops.side_stream_join(self.device)
        ops.wgrad_flush(self.device)
        self.opt.gather_grads(0)

    def _update(self):
        self.opt.launch_update()

    def _eager(self):
        first = not self._slots_checked and not torch.cuda.is_current_stream_capturing()
        if first:
            self.opt._grad.view(torch.int32).fill_(self._SENTINEL)
        self._eager_body(self._check_slots_written if first else None)

    def _eager_body(self, check=None):
        """check (first eager step only): called when backward has filled the whole bucket and nothing has consumed it yet
        -- the staged form then starts both all-reduces after it instead of overlapping the first with stage 2."""
        if self.staged:
            self._stage1()
            w1 = self.opt.all_reduce_grads(1, async_op=True) if check is None else None
            self._stage2()
            if check is not None:
                check()
                w1 = self.opt.all_reduce_grads(1, async_op=True)
            w0 = self.opt.all_reduce_grads(0, async_op=True)
            w1.wait()
            w0.wait()
        else:
            self._fwd_bwd()
            if check is not None:
                check()
            self.opt.all_reduce_grads()
        self._update()

    # -- public ---------------------------------------------------------------------------------
    def _point_at(self, t):
        if self._rotation != 1:
            self.pcm_table[1:3].zero_()
            self.pcm_table[1:2].fill_(1)
            self._rotation, self._bound = 1, None
        if self._bound is not t or self._bound_ptr != t.data_ptr():
            self.pcm_word.fill_(t.data_ptr())       # one tiny launch on the current stream, ordered with the steps around it
            self._bound, self._bound_ptr = t, t.data_ptr()

    def bind_rotation(self, batches, first=0):
        """The next steps read batches[first], batches[first + 1], ... (cyclically) IN PLACE, one per step, with nothing launched
        between the replays: the front-end indexes a device table of the n addresses with the optimizer's device-side step
        count (DAM_PCM_ROTATE).  batches: 1..MAX_ROTATION contiguous tensors like ``self.pcm`` on this device (resident
        batches, or the staging buffers of an uploader); the caller keeps them valid.  Synchronises once (reads the step
        count); ``bind_clips`` / ``load_*`` return to the single-batch form."""
        n = len(batches)
        if not 1 <= n <= self.MAX_ROTATION:
            raise ValueError('bind_rotation: 1..%d batches' % self.MAX_ROTATION)
        for t in batches:
            if t.device != self.pcm.device or t.dtype != self.pcm.dtype or tuple(t.shape) != tuple(self.pcm.shape) \
                    or not t.is_contiguous():
                raise ValueError('bind_rotation: contiguous %s %s tensors on %s' % (self.pcm.dtype, tuple(self.pcm.shape), self.pcm.device))
        if self.gain is not None:
            self.gain.fill_(1.0)
        now = int(self.opt._step.item())
        words = [self.opt._step.data_ptr(), n, (first - now) % n] + [t.data_ptr() for t in batches]
        self.pcm_table[:len(words)].copy_(torch.tensor(words, dtype=torch.int64))
        self._rotation, self._bound, self._bound_ptr = n, list(batches), None

    _bound_ptr = None

    def load_batch(self, stems, mix):
        """Copies one batch of PCM (stems [B,S,n,ch] and mix [B,n,ch], already on the device) into the static input."""
        self.pcm[:, :self.n_stems].copy_(stems, non_blocking=True)
        self.pcm[:, self.n_stems].copy_(mix, non_blocking=True)
        self._point_at(self.pcm)

    def load_features(self, x, gt):
        """feature_shape mode: copies one batch of features (x [B,S,F,T], gt [B,F,T]; device or page-locked host memory)
        into the static inputs of the step."""
        self.x.copy_(x, non_blocking=True)
        self.gt.copy_(gt, non_blocking=True)

    def load_clips(self, clips):
        """Copies one batch of whole clips [B, S+1, n, ch] (mix last; device or page-locked host memory) into the
        static input: one contiguous copy."""
        self.pcm.copy_(clips, non_blocking=True)
        self._point_at(self.pcm)

    def bind_clips(self, clips, gain=None):
        """Zero-copy form of load_clips for a batch that is ALREADY on this device: the next steps read `clips`
        ([B, S+1, n, ch] float32, contiguous, mix last) in place -- only the front-end's 8-byte address word changes.
        The caller keeps `clips` alive and unmodified until those steps have run (a reference is held here until the
        next load/bind).  gain (track_gains steps): the batch's [B, S+1] per-track gains, copied into the static table
        (None: ones)."""
        if clips.device != self.pcm.device or clips.dtype != self.pcm.dtype or tuple(clips.shape) != tuple(self.pcm.shape) \
                or not clips.is_contiguous():
            raise ValueError('bind_clips: a contiguous %s %s tensor on %s' % (self.pcm.dtype, tuple(self.pcm.shape), self.pcm.device))
        if gain is not None and self.gain is None:
            raise ValueError('bind_clips(gain=): construct the step with track_gains=True')
        if self.gain is not None:
            if gain is None:
                self.gain.fill_(1.0)
            else:
                self.gain.copy_(gain.reshape(self.gain.shape), non_blocking=True)
        self._point_at(clips)

    def capture(self, warmup=3):
        """Eager warm-up on a side stream (sizes every workspace), then capture."""
        self.opt.sync_hyper()
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        if not self.use_graph:
            return self
        # Other threads keep talking to the GPU while this one captures (MultitrackAudioDataset.iter_batches' feeder: event
        # waits, H2D copies, allocator misses, its own front-end launch on the copy stream).  In torch's default "global"
        # capture mode any such call from ANY thread invalidates the capture; "thread_local" confines the check to this
        # thread, whose calls are all capturable.  staging.capture_guard additionally keeps the feeder's GPU section and a
        # capture from overlapping at all (the feeder takes it per batch).
        from . import staging
        mode = dict(capture_error_mode='thread_local')
        ga = torch.cuda.CUDAGraph()
        with staging.capture_guard:
            if self.staged:
                ga2, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, **mode):
                    self._stage1()
                with torch.cuda.graph(ga2, pool=ga.pool(), **mode):
                    self._stage2()
                with torch.cuda.graph(gb, pool=ga.pool(), **mode):
                    self._update()
                self._graphs = (ga, ga2, gb)
            elif self.opt.world_size > 1:
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, **mode):
                    self._fwd_bwd()
                with torch.cuda.graph(gb, pool=ga.pool(), **mode):
                    self._update()
                self._graphs = (ga, gb)
            else:
                with torch.cuda.graph(ga, **mode):
                    self._fwd_bwd()
                    self._update()
                self._graphs = (ga,)
        return self

    def exposed_wait_ms(self):
        """Mean device time per step the all-reduces were NOT hidden behind backward (measure_exposed; N-rank staged steps);
        None when nothing was measured.  Synchronises."""
        if not self._exposed:
            return None
        torch.cuda.synchronize(self.device)
        ms = sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)
        self._exposed = []
        return ms

    def a2_ms(self):
        """Mean device time of graph A2 (backward of the shallow layers) per step while measure_a2 was on; None if nothing was
        measured.  With the big bucket's all-reduce beside it (overlap_reduce) against alone: what the overlap costs backward.
        Synchronises."""
        if not self._a2:
            return None
        torch.cuda.synchronize(self.device)
        ms = sum(a.elapsed_time(b) for a, b in self._a2) / len(self._a2)
        self._a2 = []
        return ms

    def __call__(self):
        """Runs one step on the data currently in the static buffers; returns the (device) loss tensor."""
        self.opt.sync_hyper()
        g = self._graphs
        if g is None:
            self._eager()
        elif len(g) == 1:
            g[0].replay()
        elif len(g) == 2:
            g[0].replay()
            self.opt.all_reduce_grads()
            g[1].replay()
        elif self.reduce_mode == 'inline':
            g[0].replay()
            self.opt.all_reduce_grads(1)            # on the training stream: nothing waits across streams, nothing overlaps
            if self.measure_a2:
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record()
            g[1].replay()
            if self.measure_a2:
                a1.record()
                self._a2.append((a0, a1))
            if self.measure_exposed:                # inline: the whole exchange of bucket 0 is exposed (bucket 1's is not timed)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.opt.all_reduce_grads(0)
            if self.measure_exposed:
                e1.record()
                self._exposed.append((e0, e1))
            g[2].replay()
        else:
            g[0].replay()
            w1 = self.opt.all_reduce_grads(1, async_op=True) if self.overlap_reduce else None     # RCCL's stream: runs beside graph A2
            if self.measure_a2:
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record()
            g[1].replay()
            if self.measure_a2:
                a1.record()
                self._a2.append((a0, a1))
            if w1 is None:
                w1 = self.opt.all_reduce_grads(1, async_op=True)
            w0 = self.opt.all_reduce_grads(0, async_op=True)
            if self.measure_exposed:      # device time from the end of backward to both buckets being there
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            w1.wait()
            w0.wait()
            if self.measure_exposed:
                e1.record()
                self._exposed.append((e0, e1))
            g[2].replay()
        self._steps_run += 1
        ops.params_changed()        # parameters / running statistics moved (a graph replay runs none of the Python above)
        return self.loss
