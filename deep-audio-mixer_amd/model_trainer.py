"""ModelTrainer for the HIP models -- same public surface and observable behaviour as the reference's
model_trainer.py:5-67: ``ModelTrainer(model, criterion, optimizer, device, model_name='scalar2d')`` and
``fit(train_loader, val_loader, start_epoch, num_epochs) -> (train_loss, val_loss)``, the same progress lines on stdout
and the same ``./weights/mixmodel_{name}_1s_{epoch:04d}_{loss:.4f}.pt`` checkpoints (the directory must
exist, "_1s_" is literal, the header prints ``num_epochs - 1``).

What is different underneath:
  * with ``criterion = nn.MSELoss()`` (what every notebook passes) loss and gradient come from the model's fused
    ``forward_mse`` (one pass over the features); any other criterion is applied to ``masked`` as in the reference;
  * the training loop body (model_trainer.py:30-37: zero_grad, forward, loss, backward, step) becomes ONE hipGraph replay
    once it has proven static (and the per-batch ``loss.item()`` is read one batch late, see ``_run``): the model is this
    package's, the criterion ``nn.MSELoss()``, the optimizer this package's ``optim.Adam`` OR the plain
    ``torch.optim.Adam(model.parameters(), weight_decay=1e-5)`` of training.ipynb cell 11 -- that one is ADOPTED (see
    ``_adopt_torch_adam``: same hyper-parameter dict, its moments become views of the fused optimizer's buffers, so the
    caller's object, its ``state_dict()`` and an LR scheduler attached to it keep working) --
    and the batch shape has repeated -- the first ``EAGER_BATCHES`` batches run the eager autograd sequence, the next one
    captures ``engine.TrainStep`` (its warm-up step is rolled back, so no extra optimizer step is
    taken) and from then on a batch is a copy into the static inputs + a replay -- or, for the ``PcmBatch`` items of
    ``MultitrackAudioDataset.batch_loader(pcm=True)``, no copy at all: the captured step contains the front-end and reads the
    uploaded PCM in place; batches of another shape (a ragged last
    batch) run eagerly.  The reference's per-batch ``loss.item()`` and prints stay.  ``graph=False`` keeps everything eager;
    when the captured path was wanted but cannot be taken (another criterion, another optimizer), ONE warning says why;
  * ``DataLoader(dataset, batch_size, num_workers=6, pin_memory=True)`` over this package's MultitrackAudioDataset (training.ipynb
    cell 6 as written) yields HostPcmBatch objects -- decoded clips in page-locked host memory, no GPU work in the workers;
    they are uploaded on a copy stream into alternating device slots (``_upload``) and bound to the same PCM-fed captured step;
  * loaders without ``__len__`` (generators such as ``MultitrackAudioDataset.iter_batches``) are accepted: the epoch mean
    is taken over the batches seen;
  * like the reference, the trainer never switches the model between train and eval mode: validation runs
    under ``torch.no_grad()`` with whatever mode the caller left the model in;
  * under ``torch.distributed`` only rank 0 prints and saves.
"""
import os
import time
import warnings

import torch

_COPY_MARK = os.environ.get('DAM_TRAINER_COPY_MARK', '1') != '0'     # A/B switch: uploads timed by the step mark
_LOG_EVERY = 10                                   # model_trainer.py:39
_CKPT_PATTERN = 'mixmodel_{}_1s_{:04d}_{:.4f}.pt'  # model_trainer.py:64


def _rank0():
    dist = torch.distributed
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def _adopt_torch_adam(optimizer):
    """training.ipynb cell 11 hands ModelTrainer a ``torch.optim.Adam(model.parameters(), weight_decay=1e-5)``.  Returns
    (this package's fused Adam over the same parameters, None) when that optimizer is one the fused launch reproduces --
    exactly torch.optim.Adam, one parameter group, L2 weight decay, no amsgrad / maximize / tensor lr, float32 CUDA
    parameters, one common step count -- else (None, reason).  The adopted pair stays ONE optimizer to the caller:
      * ``param_groups[0]`` is the SAME dict object in both, so ``optimizer.param_groups[0]['lr'] = ...`` or an LR scheduler
        built on the caller's object reaches the fused launch (optim.Adam.sync_hyper reads the dict every step);
      * the caller's per-parameter ``exp_avg`` / ``exp_avg_sq`` become views of the fused optimizer's flat buffers (after
        its existing state has been loaded into them), so ``optimizer.state_dict()`` is always current; the per-parameter
        step counts are written back at the end of every fit() and by close()."""
    from .optim import Adam
    if type(optimizer) is not torch.optim.Adam:
        return None, 'the optimizer is %s, not torch.optim.Adam or this package\'s optim.Adam' % type(optimizer).__name__
    if len(optimizer.param_groups) != 1:
        return None, 'the optimizer has %d parameter groups' % len(optimizer.param_groups)
    g = optimizer.param_groups[0]
    for flag in ('amsgrad', 'maximize', 'differentiable', 'decoupled_weight_decay'):
        if g.get(flag):
            return None, 'torch.optim.Adam(%s=True) is not what the fused launch computes' % flag
    if isinstance(g['lr'], torch.Tensor):
        return None, 'a tensor learning rate'
    params = [p for p in g['params'] if p.requires_grad]
    if not params or any((not p.is_cuda) or p.dtype != torch.float32 for p in params):
        return None, 'parameters that are not float32 CUDA tensors'
    try:
        saved = optimizer.state_dict() if optimizer.state else None
        new = Adam(g['params'], lr=g['lr'], betas=g['betas'], eps=g['eps'], weight_decay=g['weight_decay'])
        if saved is not None:
            new.load_state_dict(saved)
    except (ValueError, RuntimeError) as e:
        return None, str(e)
    new.param_groups[0] = g
    new._hyper_host = None
    new.sync_hyper()
    return new, None


class ModelTrainer:
    EAGER_BATCHES = 2          # same-shape batches run eagerly before the step is captured

    def __init__(self, model, criterion, optimizer, device, model_name='scalar2d', *, graph=True):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.device = device
        self.model_name = model_name
        self.weights_dir = './weights'
        fusable = type(criterion) is torch.nn.MSELoss and criterion.reduction == 'mean'
        self._fused = fusable and hasattr(model, 'forward_mse')
        from .optim import Adam
        self._adopted_from = None
        why = None
        if not self._fused:
            why = 'the criterion is not nn.MSELoss() on a model of this package'
        elif bool(graph) and not isinstance(optimizer, Adam):
            fused_opt, why = _adopt_torch_adam(optimizer)
            if fused_opt is not None:
                self._adopted_from, self.optimizer = optimizer, fused_opt
                self._push_adopted_state()
        self._graphable = bool(graph) and self._fused and isinstance(self.optimizer, Adam)
        if bool(graph) and not self._graphable:
            warnings.warn('ModelTrainer: the training loop runs launch by launch (eager), not as a captured hipGraph: %s' % why,
                          RuntimeWarning, stacklevel=2)
        self._step = self._shape = None
        self._seen = 0
        # With the fused optimizer the backward kernels write into its flat gradient bucket from the FIRST batch on (what
        # engine.TrainStep binds for the captured step): the eager batches -- the first EAGER_BATCHES, a ragged last one, every
        # batch with graph=False -- then run the same launches as the captured ones (queued weight gradients, the equal ones of
        # a deep stage batched: another pixel split, i.e. another summation order, than the unqueued form), so the loss list
        # of a run does not depend on which batches happened to be captured.  close() gives the model back to plain autograd.
        if self._fused and isinstance(self.optimizer, Adam):
            self.optimizer.bind_grad_slots()
        self.graph_steps = self.eager_steps = 0       # diagnostic: how the training batches were run
        # diagnostic: where the HOST spent a training epoch's wall time (seconds, summed over fit()): waiting for the loader,
        # enqueuing the batch's work, waiting for the previous batch's loss; 'epoch_start' = the part of 'loader' before an epoch's
        # FIRST batch arrived (a DataLoader with workers forks them there, every epoch)
        self.host_times = {'loader': 0.0, 'enqueue': 0.0, 'loss_wait': 0.0, 'epoch_start': 0.0}

    # ---- an adopted torch.optim.Adam stays usable by its owner
    def _push_adopted_state(self):
        """The caller's torch.optim.Adam sees the fused optimizer's moments (views) and step count."""
        orig, opt = self._adopted_from, self.optimizer
        if orig is None:
            return
        step = float(opt._step.item())
        for i, p in enumerate(opt._params):
            lo, hi = opt._offsets[i], opt._offsets[i + 1]
            st = orig.state[p]
            st['exp_avg'] = opt._exp_avg[lo:hi].view(p.shape)
            st['exp_avg_sq'] = opt._exp_avg_sq[lo:hi].view(p.shape)
            old = st.get('step')
            st['step'] = torch.tensor(step, dtype=torch.float32, device=old.device if isinstance(old, torch.Tensor) else 'cpu')

    def _adoption_intact(self, deep=True):
        """Is the caller's torch.optim.Adam still ONE optimizer with the fused one?  ``optimizer.load_state_dict(...)`` AFTER
        this trainer was built (the usual resume order) replaces ``param_groups[0]`` by a new dict and every ``exp_avg`` /
        ``exp_avg_sq`` by a fresh tensor: lr edits would no longer reach the fused launch and the loaded moments would be
        ignored -- and overwritten at the end of fit()."""
        orig, opt = self._adopted_from, self.optimizer
        if orig.param_groups[0] is not opt.param_groups[0]:
            return False
        for i, p in enumerate(opt._params if deep else opt._params[:1]):
            st = orig.state.get(p)
            if not st:
                return False
            off = 4 * opt._offsets[i]
            if (st['exp_avg'].data_ptr() != opt._exp_avg.data_ptr() + off
                    or st['exp_avg_sq'].data_ptr() != opt._exp_avg_sq.data_ptr() + off):
                return False
        return True

    def _resync_adopted(self, deep=True):
        """An ``optimizer.load_state_dict(...)`` on the caller's object since the last batch (hyper-parameters, moments, step
        count) is loaded into the fused optimizer's flat buffers and re-shared as views.  True if that happened."""
        orig, opt = self._adopted_from, self.optimizer
        if orig is None or self._adoption_intact(deep):
            return False
        if orig.state:
            opt.load_state_dict(orig.state_dict())
        else:                                        # state cleared: a fresh optimizer
            opt._exp_avg.zero_()
            opt._exp_avg_sq.zero_()
            opt._step.zero_()
        opt.param_groups[0] = orig.param_groups[0]
        opt._hyper_host = None
        opt.sync_hyper()
        self._push_adopted_state()
        return True

    def _pull_adopted_step(self):
        """Start of a fit(): whatever the caller did to its own optimizer since the last one reaches the fused one -- a
        ``load_state_dict``, or plain ``step()`` calls (the fused one continues from that count; the caller's per-parameter
        counts are current here: they are written back at the end of every fit() and by close())."""
        orig, opt = self._adopted_from, self.optimizer
        if orig is None or self._resync_adopted():
            return
        steps = {int(st['step']) for st in orig.state.values() if 'step' in st}
        if len(steps) == 1:
            k = steps.pop()
            if k != int(opt._step.item()):
                opt._step.fill_(k)

    # ---- the captured step
    def _capture(self, feats, target, pcm=None):
        """Builds and captures engine.TrainStep for this batch shape (pcm: a PcmBatch -- the step then starts from the
        uploaded PCM and contains the front-end).  Capturing needs one eager warm-up step through the
        slot-bound path (it sizes the workspaces the graph will bake in); that step is rolled back -- parameters, Adam
        moments and step count, BatchNorm running buffers -- so the trajectory is exactly the eager loop's.  The rollback
        also happens when the warm-up or the capture raises: a failed capture leaves the model as it found it."""
        from .engine import TrainStep
        opt, model = self.optimizer, self.model
        if pcm is not None:
            B, K, n, ch = pcm.clips.shape
            step = TrainStep(model, opt, K - 1, n, ch, B, pcm.n_fft, pcm.hop, use_graph=True, device=pcm.clips.device,
                             pcm_dtype=pcm.clips.dtype, track_gains=pcm.gain is not None, normalize=pcm.normalize,
                             copy_mark=_COPY_MARK)
        else:
            B, S, F, T = feats.shape
            step = TrainStep(model, opt, S, batch=B, feature_shape=(F, T), use_graph=True, device=feats.device, copy_mark=_COPY_MARK)
        keep = [t.clone() for t in (opt._flat, opt._exp_avg, opt._exp_avg_sq, opt._step)]
        bufs = [(b, b.clone()) for b in model.buffers()]
        training = model.training
        try:
            if pcm is not None:
                step.bind_clips(pcm.clips, pcm.gain)
            else:
                step.load_features(feats, target)
            step.capture(warmup=1)
        except BaseException:
            step.close()
            raise
        finally:
            for dst, src in zip((opt._flat, opt._exp_avg, opt._exp_avg_sq, opt._step), keep):
                dst.copy_(src)
            for b, saved in bufs:
                b.copy_(saved)
            model.train(training)
        return step

    # ---- batches that arrive as decoded clips in HOST memory (DataLoader(dataset, num_workers>0, pin_memory=True))
    PCM_SLOTS = 2

    def _upload(self, host):
        """data.dataset.HostPcmBatch -> PcmBatch in one of PCM_SLOTS device buffers, the copy on a private stream: batch k + 1
        travels while the step on batch k runs (the host is one batch ahead of the device, see _run), and the step -- whose
        captured front-end reads the clips in place -- waits for it on the device only.  What ``.to(self.device)`` is to the
        reference's loop (model_trainer.py:34), without the 38 MB feature copy and without blocking the training stream."""
        dev = torch.device(self.device)
        B, rest, dtype = host.clips.shape[0], tuple(host.clips.shape[1:]), host.clips.dtype
        st = getattr(self, '_pcm_stage', None)
        if st is None or st['key'] != (rest, dtype) or st['cap'] < B:
            st = self._pcm_stage = {
                'key': (rest, dtype), 'cap': B, 'k': 0, 'stream': torch.cuda.Stream(device=dev),
                'bufs': [torch.empty((B,) + rest, dtype=dtype, device=dev) for _ in range(self.PCM_SLOTS)],
                'ready': [torch.cuda.Event() for _ in range(self.PCM_SLOTS)],
                'consumed': [torch.cuda.Event() for _ in range(self.PCM_SLOTS)]}
        slot = st['k'] % self.PCM_SLOTS
        st['k'] += 1
        cur = torch.cuda.current_stream(dev)
        # the step that read this slot's previous batch must be over: the HOST waits (already true when _run has read that
        # step's loss, one batch late) -- a copy stream waiting for the training stream's event would cost the training stream
        # 0.09 ms per step on this stack (profiles/r05_sync_cost_probe.txt)
        st['consumed'][slot].synchronize()
        # (not timed by the step mark: this loop is host-bound -- the workers' batches arrive every 5.7 ms -- and a blocking wait
        #  here takes the slack it has: 5.74 -> 5.88 ms per step measured; the feeder THREAD of batch_loader() does wait for it)
        with torch.cuda.stream(st['stream']):
            pcm = host.to_device(dev, out=st['bufs'][slot])
            st['ready'][slot].record(st['stream'])
        cur.wait_event(st['ready'][slot])
        if pcm.gain is not None:
            pcm.gain.record_stream(cur)
        st['busy'] = slot
        return pcm

    def _uploaded_batch_enqueued(self):
        """Everything that reads the slot of the last _upload() is on the current stream now."""
        st = getattr(self, '_pcm_stage', None)
        if st is not None and st.get('busy') is not None:
            st['consumed'][st['busy']].record(torch.cuda.current_stream(torch.device(self.device)))
            st['busy'] = None

    def _arm_gate(self, loader=None):
        """Times the loader's uploads by the captured step's mark (engine.TrainStep.copy_mark): a MultitrackAudioDataset's
        feeder thread (``upload_gate``) waits for the step enqueued last to reach its backward pass before it enqueues a
        host-to-device copy -- beside the forward pass the copy slows the latency-bound launches."""
        if loader is not None:
            self._loader = loader
        ds = getattr(getattr(self, '_loader', None), 'dataset', None)
        if ds is not None and hasattr(ds, 'upload_gate'):
            ds.upload_gate = self._step.copy_mark if self._step is not None else None

    def _train_batch(self, batch):
        if hasattr(batch, 'to_device'):                        # data.dataset.HostPcmBatch: decoded clips in host memory
            try:
                return self._train_device_batch(self._upload(batch))
            finally:
                self._uploaded_batch_enqueued()
        return self._train_device_batch(batch)

    def _train_device_batch(self, batch):
        pcm = batch if hasattr(batch, 'clips') else None       # data.dataset.PcmBatch: uploaded PCM, front-end not run yet
        if pcm is not None:
            feats = target = None
            shape = ('pcm', tuple(pcm.clips.shape), pcm.clips.dtype, pcm.gain is not None, pcm.normalize, pcm.n_fft, pcm.hop,
                     self.model.training)
        else:
            feats, target = (t.to(self.device) for t in batch)
            shape = (tuple(feats.shape), tuple(target.shape), feats.dtype, self.model.training)
        self._resync_adopted(deep=False)             # optimizer.load_state_dict() since the last batch?
        if self._graphable and shape == self._shape and (self._step is not None or self._seen >= self.EAGER_BATCHES):
            if self._step is None:
                self._resync_adopted()
                self._step = self._capture(feats, target, pcm)
                self._arm_gate()
            if pcm is not None:
                self._step.bind_clips(pcm.clips, pcm.gain)       # an 8-byte address word (+ the gain table): no PCM / feature copy
            else:
                self._step.load_features(feats, target)
            self.graph_steps += 1
            return self._step()
        if shape != self._shape and self._step is None:
            self._shape, self._seen = shape, 0
        self._seen += shape == self._shape
        self.eager_steps += 1
        self.optimizer.zero_grad()
        loss = self._batch_loss(pcm.features() if pcm is not None else (feats, target))
        loss.backward()
        self.optimizer.step()
        return loss

    def close(self):
        """Drops the captured step and gives the model's parameters back to ordinary autograd (.grad tensors); an adopted
        torch.optim.Adam gets its step counts."""
        if self._step is not None:
            self._step.close()
            self._step = None
            self._arm_gate()
        elif getattr(self.optimizer, 'slots_bound', False):
            self.optimizer.unbind_grad_slots()
        self._push_adopted_state()

    # ---- one batch -> loss tensor (on the device)
    def _batch_loss(self, batch):
        if hasattr(batch, 'to_device'):
            try:
                return self._batch_loss(self._upload(batch).features())
            finally:
                self._uploaded_batch_enqueued()
        if hasattr(batch, 'clips'):
            batch = batch.features()
        feats, target = (t.to(self.device) for t in batch)
        if self._fused:
            return self.model.forward_mse(feats, target)[0]
        return self.criterion(self.model(feats)[0], target)

    def _run(self, loader, train):
        """Mean loss over the loader; one optimisation step per batch when ``train``.

        The reference reads ``loss.item()`` after every batch (model_trainer.py:41,43) -- a device synchronisation per step,
        during which nothing is enqueued.  Every batch's loss is still read and logged here, in order, but ONE BATCH LATE:
        the value of batch j travels to page-locked memory behind an event and is picked up after batch j + 1 has been
        enqueued, so the host's work for the next batch (loader, copies, launch) runs beside the device's work on this one.
        Same numbers, same lines on stdout, same returned mean; the host never runs more than one batch ahead."""
        total, quiet = 0.0, not _rank0()
        n = len(loader) if hasattr(loader, '__len__') else None
        on_gpu = torch.device(self.device).type == 'cuda'
        if on_gpu and getattr(self, '_loss_host', None) is None:
            self._loss_host = torch.zeros(2, dtype=torch.float32, pin_memory=True)
            self._loss_ev = [torch.cuda.Event(), torch.cuda.Event()]
        pending = []                                    # [(step index, slot)]: losses on their way to the host

        def flush():
            nonlocal total
            k, slot = pending.pop(0)
            self._loss_ev[slot].synchronize()
            value = float(self._loss_host[slot])
            if train and k % _LOG_EVERY == 0 and not quiet:
                print('[%d/%4d] loss: %.3f' % (k, n, value) if n is not None else '[%d/   ?] loss: %.3f' % (k, value))
            total += value

        step = 0
        self._arm_gate(loader if train else None)
        ht, clock = self.host_times, time.perf_counter
        t0 = clock()
        it = iter(loader)                 # (a DataLoader with num_workers > 0 forks its workers here, every epoch)
        while True:
            try:
                batch = next(it)
            except StopIteration:
                break
            step += 1
            t1 = clock()
            if train and step == 1:
                ht['epoch_start'] += t1 - t0
            loss = self._train_batch(batch) if train else self._batch_loss(batch)
            if train:
                ht['loader'] += t1 - t0
                ht['enqueue'] += clock() - t1
            if not on_gpu or not loss.is_cuda:
                value = loss.item()
                if train and step % _LOG_EVERY == 0 and not quiet:
                    print('[%d/%4d] loss: %.3f' % (step, n, value) if n is not None else '[%d/   ?] loss: %.3f' % (step, value))
                total += value
                t0 = clock()
                continue
            slot = step & 1
            self._loss_host[slot:slot + 1].copy_(loss.detach().reshape(1), non_blocking=True)
            self._loss_ev[slot].record()
            pending.append((step, slot))
            if len(pending) > 1:
                t2 = clock()
                flush()
                if train:
                    ht['loss_wait'] += clock() - t2
            t0 = clock()
        while pending:
            flush()
        return total / (n if n is not None else step)

    def _train_epoch(self, train_loader):
        return self._run(train_loader, True)

    def _validate_epoch(self, val_loader):
        with torch.no_grad():
            return self._run(val_loader, False)

    def fit(self, train_loader, val_loader, start_epoch, num_epochs):
        history = {'train': [], 'val': []}
        talk = _rank0()
        self._pull_adopted_step()
        for epoch in range(start_epoch, start_epoch + num_epochs):
            if talk:
                print('Epoch {}/{}'.format(epoch, num_epochs - 1))
            history['train'].append(self._train_epoch(train_loader))
            if talk:
                print('Epoch {} train loss: {:.4f}'.format(epoch, history['train'][-1]))
            history['val'].append(self._validate_epoch(val_loader))
            if talk:
                print('Epoch {} val loss: {:.4f}'.format(epoch, history['val'][-1]))
                print('-' * 50)
                name = _CKPT_PATTERN.format(self.model_name, epoch, history['train'][-1])
                torch.save(self.model.state_dict(), os.path.join(self.weights_dir, name))
        self._push_adopted_state()
        return history['train'], history['val']
