"""One full training step of the hot path as a replayable unit:

    interleaved PCM (stems + mix, resident in HBM)
      -> dam_stft_logmag_f32 (features x [B,S,F,T], target gt [B,F,T])
      -> model.forward_mse (conv/BN/ReLU trunk, heads, fused masked-sum + MSE)
      -> backward (BN backward, dgrad, wgrad, heads)            [model_trainer.py:30-37 of the reference]
      -> flat gradient bucket -> (RCCL all-reduce over the data-parallel group) -> Adam(+L2)

The kernel sequence is static, so after a few eager warm-up steps it is captured into a hipGraph
(torch.cuda.CUDAGraph on the ROCm build) and replayed: one host call per step instead of ~400 launches.
With more than one rank the all-reduce stays outside the graph (graph A: front-end+forward+backward+bucket,
eager all-reduce, graph B: Adam).
"""
import torch

from . import features


class TrainStep:
    def __init__(self, model, optimizer, n_stems, n_samples, channels=2, batch=8, n_fft=2048, hop=1024,
                 use_graph=True, device=None):
        self.model, self.opt = model, optimizer
        self.device = device or next(model.parameters()).device
        self.n_fft, self.hop, self.batch, self.n_stems = n_fft, hop, batch, n_stems
        f, t = n_fft // 2 + 1, features.num_frames(n_samples, hop)
        dev = self.device
        self.stems = torch.zeros((batch, n_stems, n_samples, channels), dtype=torch.float32, device=dev)
        self.mix = torch.zeros((batch, n_samples, channels), dtype=torch.float32, device=dev)
        self.x = torch.empty((batch, n_stems, f, t), dtype=torch.float32, device=dev)
        self.gt = torch.empty((batch, f, t), dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.use_graph = use_graph
        self._graph_a = self._graph_b = None
        self._steps_run = 0
        self.frames_per_step = batch * n_stems * t          # BASELINE metric unit: stem-spectrogram frames

    # -- pieces ---------------------------------------------------------------------------------
    def _front_end(self):
        b, s = self.batch, self.n_stems
        features.stft_logmag(self.stems.view(b * s, *self.stems.shape[2:]), self.n_fft, self.hop,
                             out=self.x.view(b * s, *self.x.shape[2:]))
        features.stft_logmag(self.mix, self.n_fft, self.hop, out=self.gt)

    def _fwd_bwd(self):
        self._front_end()
        self.opt.zero_grad(set_to_none=True)
        loss = self.model.forward_mse(self.x, self.gt)[0]
        loss.backward()
        self.loss.copy_(loss.detach())
        self.opt.gather_grads()

    def _update(self):
        from . import ops
        g = self.opt.param_groups[0]
        o = self.opt
        ops.adam_l2_step(o._flat, o._grad, o._exp_avg, o._exp_avg_sq, o._step, o._derived, g['lr'], g['betas'][0],
                         g['betas'][1], g['eps'], g['weight_decay'], 1.0 / o.world_size)

    def _eager(self):
        self._fwd_bwd()
        self.opt.all_reduce_grads()
        self._update()

    # -- public ---------------------------------------------------------------------------------
    def load_batch(self, stems, mix):
        """Copies one batch of PCM (already on the device) into the static input buffers."""
        self.stems.copy_(stems, non_blocking=True)
        self.mix.copy_(mix, non_blocking=True)

    def capture(self, warmup=3):
        """Eager warm-up on a side stream (sizes every workspace), then capture."""
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        if not self.use_graph:
            return self
        self._graph_a = torch.cuda.CUDAGraph()
        if self.opt.world_size > 1:
            self._graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_a):
                self._fwd_bwd()
            with torch.cuda.graph(self._graph_b, pool=self._graph_a.pool()):
                self._update()
        else:
            with torch.cuda.graph(self._graph_a):
                self._fwd_bwd()
                self._update()
        return self

    def __call__(self):
        """Runs one step on the data currently in the static buffers; returns the (device) loss tensor."""
        if self._graph_a is None:
            self._eager()
        elif self._graph_b is None:
            self._graph_a.replay()
        else:
            self._graph_a.replay()
            self.opt.all_reduce_grads()
            self._graph_b.replay()
        self._steps_run += 1
        return self.loss
