"""Full-song inference -- drop-in for the reference's inference_utils.py (interpolate_mask :12-41,
mix_song_smooth :105-145) on the GPU (BASELINE config C5).

mix_song_smooth at the reference HEAD cannot run (it hands [channels, n] slices to torch.stft); this
module implements the intended semantics: features of the channel MEAN, gains applied to the original
multichannel audio.

Everything between the PCM upload and the result download stays on the device and -- for a model in eval mode -- is ONE
hipGraph (``SongMixer``): strided STFT front-end over all chunks of all stems straight out of the planar song
(dam_stft_logmag_strided_f32) -> model forward of the whole chunk batch -> 10 ** (0.5 g) and the Savitzky-Golay
smoothing (dam_gains_smooth) -> sample-rate gain ramp x audio (dam_gain_ramp_apply), or for ``mix_song_to_master`` the
fused stem sum + peak normalisation (dam_mixdown_peak_normalize).  The host sees the song once on the way in (page-locked
double-buffered staging, staging.PinnedPipe) and the result once on the way out.

The model is applied as the reference applies it -- whatever ``model.training`` is, never toggled here:
in eval mode all chunks run as one batch inside the graph; in training mode BatchNorm uses per-call batch statistics
(and updates its running statistics), so chunks run one by one, eagerly, exactly as in the reference loop.
"""
import numpy as np
import torch

from . import features, ops, staging

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')      # inference_utils.py:9

SAVGOL_POLYORDER = 2          # inference_utils.py:140


def interpolate_mask(spec_mask: np.array, tgt_len: int) -> np.array:
    """inference_utils.py:12-41: every gain held for int(tgt_len / len) samples, the last one to the end of the track.
    Host version, kept for API parity (mix_song_smooth applies the same ramp inside dam_gain_ramp_apply)."""
    gains = np.asarray(spec_mask, dtype=np.float64)
    assert len(gains) <= tgt_len, "Target mask should be longer than the initial one"
    if len(gains) < 2:
        return np.zeros(tgt_len)          # the reference's loop body never runs for a single gain: the mask stays zero
    hold = tgt_len // len(gains)
    index = np.minimum(np.arange(tgt_len) // hold, len(gains) - 1)
    return gains[index]


def _savgol_window(num_chunks):
    w = int(num_chunks / 4)          # inference_utils.py:136-139
    return w if w % 2 else w + 1


def predict_chunk_gains(model, pcm, n_stems, n_chunks, chunk_samples, window_size=2048, hop_length=1024, feats=None):
    """pcm: CUDA [n_stems, channels, n] -> raw model outputs [n_chunks-1, n_stems] for chunks 0..n_chunks-2
    (the reference loop ``range(1, num_chunks)`` processes exactly those, inference_utils.py:111-113)."""
    n_proc = n_chunks - 1
    feats = features.stft_logmag_song_chunks(pcm, n_proc, chunk_samples, window_size, hop_length, out=feats)
    feats = feats.view(n_proc, n_stems, feats.shape[1], feats.shape[2])
    if model.training:
        return torch.cat([model.predict_gains(feats[i:i + 1]) for i in range(n_proc)], 0)
    return model.predict_gains(feats)


class SongMixer:
    """Static device buffers and the captured hipGraph of one song geometry: (model, stems, channels, samples, dtype,
    chunk length, output kind).  ``run(tracks)`` uploads, replays, downloads."""

    def __init__(self, model, n_stems, channels, n_samples, dtype, chunk_samples, kind, normalize=True,
                 out_dtype=torch.float64, use_graph=True):
        if kind not in ('stems', 'master'):
            raise ValueError(kind)
        self.model, self.kind, self.normalize = model, kind, normalize
        self.dev = next(model.parameters()).device
        self.n_stems, self.channels, self.n, self.chunk = n_stems, channels, n_samples, chunk_samples
        self.num_chunks = int(n_samples / chunk_samples)
        self.n_proc = self.num_chunks - 1
        if self.n_proc < 1:
            raise ValueError('the song must hold at least two chunks')
        self.window = _savgol_window(self.num_chunks)
        if self.window <= SAVGOL_POLYORDER or self.window > self.n_proc:
            # scipy.signal.savgol_filter raises for these at inference_utils.py:140
            raise ValueError('polyorder must be less than window_length and window_length must not exceed the number '
                             'of gains (window %d, %d gains)' % (self.window, self.n_proc))
        dev = self.dev
        self.pcm = torch.empty((n_stems, channels, n_samples), dtype=dtype, device=dev)
        t = features.num_frames(chunk_samples, 1024)
        self.feats = torch.empty((self.n_proc * n_stems, 1025, t), dtype=torch.float32, device=dev)
        if kind == 'stems':
            self.out = torch.empty((n_stems, channels, n_samples), dtype=out_dtype, device=dev)
            self.ws = None
        else:
            self.out = torch.empty((channels, n_samples), dtype=out_dtype, device=dev)
            self.ws = torch.empty(ops._lib.lib().dam_mixdown_workspace_elems(channels), dtype=out_dtype, device=dev)
        self.gains = torch.empty((2, n_stems, self.n_proc), dtype=torch.float64, device=dev)     # [raw amplitude, smoothed]
        self.graph = None
        self.use_graph = use_graph
        self._key = None

    def _body(self):
        g = predict_chunk_gains(self.model, self.pcm, self.n_stems, self.num_chunks, self.chunk, feats=self.feats)
        _, smooth = ops.gains_smooth(g, self.window, SAVGOL_POLYORDER, out=self.gains)
        if self.kind == 'stems':
            ops.gain_ramp_apply(self.pcm, smooth, out=self.out)
        else:
            ops.mixdown_peak_normalize(self.pcm, smooth, normalize=self.normalize, out=self.out, workspace=self.ws)

    def _model_key(self):
        # the captured forward holds the FOLDED conv + BatchNorm images (layers.FoldedConvBn), computed when it was captured:
        # any change of the parameters or running statistics -- torch-side (version counters) or by this library's own
        # in-place kernels (ops.PARAM_EPOCH) -- needs a new capture, not just a replay
        return (self.model.training, ops.PARAM_EPOCH) + tuple(
            (t.data_ptr(), t._version) for t in self.model.state_dict(keep_vars=True).values())

    def launch(self):
        """Runs the device pipeline on whatever is in self.pcm (graph replay when the model is in eval mode)."""
        if self.model.training or not self.use_graph:
            with torch.no_grad():
                self._body()
            return
        key = self._model_key()
        if self.graph is None or key != self._key:
            with torch.no_grad():
                s = torch.cuda.Stream(device=self.dev)
                s.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(s):
                    self._body()                   # warm-up: sizes every workspace, builds the weight-packing table
                torch.cuda.current_stream(self.dev).wait_stream(s)
                torch.cuda.synchronize(self.dev)
                self.graph = torch.cuda.CUDAGraph()
                with staging.capture_guard, torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                    self._body()           # (thread_local: other threads' GPU calls do not invalidate the capture)
            self._key = key
        self.graph.replay()

    def run(self, tracks):
        """tracks: list of n_stems host arrays [channels, n_samples].  Returns (out ndarray, gains ndarray [2, S, n_proc])."""
        pipe = staging.pipe_for(self.dev)
        for i, a in enumerate(tracks):
            pipe.upload(self.pcm[i], a)
        self.launch()
        out = pipe.download(self.out)
        return out, self.gains.cpu().numpy()


_mixers = {}


def _mixer(model, stems, loaded_tracks, chunk_length, sr, kind, normalize, out_dtype):
    first = np.asarray(loaded_tracks[stems[0]])
    if first.ndim != 2:
        raise ValueError('loaded_tracks[track] must be [channels, n] arrays')
    ch, n = first.shape
    dt = torch.float32 if first.dtype == np.float32 else torch.float64
    key = (id(model), len(stems), ch, n, dt, chunk_length * sr, kind, bool(normalize), out_dtype)
    m = _mixers.get(key)
    if m is None:
        _mixers.clear()                        # one geometry at a time: a song's buffers are hundreds of MB
        m = SongMixer(model, len(stems), ch, n, dt, chunk_length * sr, kind, normalize, out_dtype)
        _mixers[key] = m
    np_dt = np.float32 if dt == torch.float32 else np.float64
    return m, [np.asarray(loaded_tracks[t], dtype=np_dt) for t in stems]


def mix_song_smooth(dataset, model, loaded_tracks: dict, chunk_length=1, sr=44100):
    """Returns (mixed_tracks {track: ndarray[channels, n] float64}, raw_gains {track: [float]}, smooth_gains {track: list})."""
    stems = [t for t in dataset.get_tracklist() if t != 'mix']
    m, arrays = _mixer(model, stems, loaded_tracks, chunk_length, sr, 'stems', False, torch.float64)
    out, gains = m.run(arrays)
    raw_gains = {t: [float(v) for v in gains[0, i]] for i, t in enumerate(stems)}
    smooth_gains = {t: list(gains[1, i]) for i, t in enumerate(stems)}
    return {t: out[i] for i, t in enumerate(stems)}, raw_gains, smooth_gains


def mix_song_to_master(dataset, model, loaded_tracks: dict, chunk_length=1, sr=44100, normalize=True, dtype=np.float64):
    """mix_song_smooth followed by what every caller of the reference does next (inference.ipynb cells 9/11,
    evaluation.py:59-66): ``track_sum = np.sum(list(mixed_tracks.values()), axis=0)`` and, if ``normalize``,
    ``librosa.util.normalize(track_sum, axis=1)`` -- fused into one pass over the song on the GPU (the per-stem mixed
    tracks are never materialised).  Returns (mix ndarray[channels, n], raw_gains, smooth_gains)."""
    stems = [t for t in dataset.get_tracklist() if t != 'mix']
    out_dt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    m, arrays = _mixer(model, stems, loaded_tracks, chunk_length, sr, 'master', normalize, out_dt)
    out, gains = m.run(arrays)
    raw_gains = {t: [float(v) for v in gains[0, i]] for i, t in enumerate(stems)}
    return out, raw_gains, {t: list(gains[1, i]) for i, t in enumerate(stems)}
