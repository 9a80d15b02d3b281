"""GPU feature front-end: batched STFT -> |.| -> dB through ``dam_stft_logmag_f32``.

Mirrors the arithmetic of the reference's ``MultitrackAudioDataset.compute_features``
(data/dataset.py:132-162) with ``_stereo_to_mono`` (:181-183) and ``_augment_audio``
(:164-168) fused in, for all tracks of a batch in one launch.
"""
import ctypes

import numpy as np
import torch

from . import _lib

AMIN = 1e-5      # data/dataset.py:154
_tables = {}
# include/dam_hip.h dam_pcm_dtype.  int16 / int32: integer PCM as the WAV file holds it, scaled by 1/2^15 / 1/2^31 inside the
# kernel exactly as soundfile.read scales it (24-bit files arrive left-justified in int32, dataset_utils.read_wav_native)
PCM_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int16: 2, torch.int32: 3}


def _pcm_code(pcm, planar=False):
    code = PCM_DTYPES.get(pcm.dtype)
    if code is None or (planar and code >= 2):
        raise TypeError('pcm must be float32 or float64%s' % ('' if planar else ', or int16 / int32 integer PCM'))
    return code


def _get_tables(device, n_fft):
    key = (device.type, device.index, n_fft)
    if key not in _tables:
        L = _lib.lib()
        n = L.dam_stft_twiddle_count(n_fft)
        tw = np.empty(2 * n, dtype=np.float32)
        _lib.check(L.dam_stft_fill_twiddles_host(n_fft, tw.ctypes.data_as(ctypes.c_void_p)), 'dam_stft_fill_twiddles_host')
        # the window is torch's own float32 periodic Hann table, computed on the CPU exactly as the
        # reference does (torch.hann_window(window_size), data/dataset.py:148) and uploaded
        win = torch.hann_window(n_fft, dtype=torch.float32)
        _tables[key] = (win.to(device), torch.from_numpy(tw).to(device))
    return _tables[key]


def num_frames(n_samples, hop):
    return 1 + n_samples // hop


def stft_logmag(pcm, n_fft=2048, hop=1024, gain=None, normalize=False, out=None):
    """pcm: CUDA tensor [n_tracks, n_samples, channels] or [n_tracks, n_samples], float32 / float64 in [-1, 1) or int16 /
    int32 integer PCM (scaled by 1/2^15 / 1/2^31 in the kernel), channels in {1, 2} interleaved.
    Returns float32 [n_tracks, n_fft/2+1, T] in dB."""
    _lib.require_cuda(pcm, gain, out)
    if pcm.dim() == 2:
        pcm = pcm.unsqueeze(-1)
    if pcm.dim() != 3:
        raise ValueError('pcm must be [tracks, samples(, channels)]')
    code = _pcm_code(pcm)
    pcm = pcm.contiguous()
    n_tracks, n, ch = pcm.shape
    t = num_frames(n, hop)
    win, tw = _get_tables(pcm.device, n_fft)
    if out is None:
        out = torch.empty((n_tracks, n_fft // 2 + 1, t), dtype=torch.float32, device=pcm.device)
    elif out.shape != (n_tracks, n_fft // 2 + 1, t) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError('bad out tensor')
    if gain is not None:
        gain = gain.to(device=pcm.device, dtype=torch.float32).contiguous()
        if gain.numel() != n_tracks:
            raise ValueError('gain must have one entry per track')
    st = _lib.lib().dam_stft_logmag_f32(_lib.ptr(pcm), code, n_tracks, n, ch,
                                        n * ch, _lib.ptr(win), _lib.ptr(tw), _lib.ptr(gain), n_fft, hop,
                                        AMIN, 1 if normalize else 0, _lib.ptr(out), _lib.stream())
    _lib.check(st, 'dam_stft_logmag_f32')
    return out


def stft_logmag_song_chunks(pcm, n_chunks, chunk_samples, n_fft=2048, hop=1024, out=None):
    """pcm: CUDA [S, channels, n] planar (the reference's loaded_tracks[track] arrays, stacked) -> dB features of the
    channel mean of chunks 0..n_chunks-1 of every stem, [n_chunks * S, n_fft/2+1, T] with row = chunk * S + stem --
    the batch the chunk loop at inference_utils.py:111-123 builds one chunk at a time.  Reads the song in place
    (strided launch, no gather copy)."""
    _lib.require_cuda(pcm, out)
    if pcm.dim() != 3:
        raise ValueError('pcm must be a float32/float64 [stems, channels, samples] tensor')
    _pcm_code(pcm, planar=True)
    pcm = pcm.contiguous()
    S, ch, n = pcm.shape
    if n_chunks * chunk_samples > n:
        raise ValueError('the song is shorter than n_chunks chunks')
    t = num_frames(chunk_samples, hop)
    win, tw = _get_tables(pcm.device, n_fft)
    shape = (n_chunks * S, n_fft // 2 + 1, t)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=pcm.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError('bad out tensor')
    st = _lib.lib().dam_stft_logmag_strided_f32(_lib.ptr(pcm), 0 if pcm.dtype == torch.float32 else 1, n_chunks, chunk_samples,
                                                S, ch * n, chunk_samples, ch, 1, n, _lib.ptr(win), _lib.ptr(tw), None,
                                                n_fft, hop, AMIN, 0, _lib.ptr(out), None, 0, _lib.stream())
    _lib.check(st, 'dam_stft_logmag_strided_f32')
    return out


def stft_logmag_clips(pcm, n_fft=2048, hop=1024, gain=None, normalize=False, out_stems=None, out_mix=None, pcm_word=None,
                      pcm_table=None):
    """pcm: CUDA [B, S+1, n, channels] (or [B, S+1, n]) float32/float64 -- a batch of clips, every clip's S stems
    followed by its mix, interleaved channels (what data/dataset.py:192-196 reads per item).  ONE launch for all
    B*(S+1) tracks; returns (x [B, S, F, T], gt [B, F, T]) float32 dB -- the collated (train_features, gt_features) of
    data/dataset.py:207-210.  gain: optional [B, S+1] augmentation draws (data/dataset.py:198-199, the mix included).
    pcm_word: optional CUDA int64[1] holding the ADDRESS of the batch to read (DAM_PCM_INDIRECT, include/dam_hip.h); `pcm`
    then only describes shape and dtype -- a captured launch follows the word.  pcm_table: optional CUDA int64 table
    {address of a step counter, n, offset, addr[0..n)} (DAM_PCM_ROTATE): the launch reads addr[(counter + offset) % n] -- a
    captured step walks n batches without anything being re-pointed between the replays (engine.TrainStep.bind_clips /
    bind_rotation)."""
    _lib.require_cuda(pcm, gain, out_stems, out_mix)
    if pcm.dim() == 3:
        pcm = pcm.unsqueeze(-1)
    if pcm.dim() != 4:
        raise ValueError('pcm must be a [clips, tracks, samples(, channels)] tensor')
    code = _pcm_code(pcm)
    pcm = pcm.contiguous()
    B, K, n, ch = pcm.shape
    if K < 2:
        raise ValueError('a clip needs at least one stem and the mix')
    f, t = n_fft // 2 + 1, num_frames(n, hop)
    win, tw = _get_tables(pcm.device, n_fft)
    if out_stems is None:
        out_stems = torch.empty((B, K - 1, f, t), dtype=torch.float32, device=pcm.device)
    if out_mix is None:
        out_mix = torch.empty((B, f, t), dtype=torch.float32, device=pcm.device)
    for o, shape in ((out_stems, (B, K - 1, f, t)), (out_mix, (B, f, t))):
        if tuple(o.shape) != shape or o.dtype != torch.float32 or not o.is_contiguous():
            raise ValueError('bad out tensor')
    if gain is not None:
        gain = gain.to(device=pcm.device, dtype=torch.float32).contiguous()
        if gain.numel() != B * K:
            raise ValueError('gain must have one entry per track')
    src, flag = _lib.ptr(pcm), 0
    if pcm_word is not None:
        _lib.require_cuda(pcm_word)
        if pcm_word.dtype != torch.int64 or pcm_word.numel() != 1:
            raise ValueError('pcm_word: one int64 on the device')
        src, flag = _lib.ptr(pcm_word), 0x100
    if pcm_table is not None:
        _lib.require_cuda(pcm_table)
        if pcm_word is not None or pcm_table.dtype != torch.int64 or pcm_table.numel() < 4 or not pcm_table.is_contiguous():
            raise ValueError('pcm_table: a contiguous int64 table on the device (and no pcm_word beside it)')
        src, flag = _lib.ptr(pcm_table), 0x200
    st = _lib.lib().dam_stft_logmag_strided_f32(src, code | flag, B, K * n * ch, K,
                                                n * ch, n, ch, ch, 1, _lib.ptr(win), _lib.ptr(tw), _lib.ptr(gain), n_fft,
                                                hop, AMIN, 1 if normalize else 0, _lib.ptr(out_stems), _lib.ptr(out_mix), 1,
                                                _lib.stream())
    _lib.check(st, 'dam_stft_logmag_strided_f32')
    return out_stems, out_mix


def augment_gains(seed, n_tracks, items=None, first_item=0, n_items=None, lo=0.6, hi=1.4, device=None, out=None):
    """[n_items, n_tracks] float32 gains in [lo, hi): the reference's per-track augmentation draw (data/dataset.py:164-168,
    198-199) as a reproducible device-side function of (seed, global item index, track).  items: int64 tensor / list of
    global item indices, or (first_item, n_items) for a consecutive run."""
    if items is not None:
        items = torch.as_tensor(items, dtype=torch.int64)
        n_items = items.numel()
        device = device or (items.device if items.is_cuda else None)
    device = torch.device(device or 'cuda')
    if items is not None:
        items = items.to(device).contiguous()
    if out is None:
        out = torch.empty((n_items, n_tracks), dtype=torch.float32, device=device)
    _lib.require_cuda(out)
    _lib.check(_lib.lib().dam_augment_gains_f32(int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(items), int(first_item), int(n_items),
                                                int(n_tracks), float(lo), float(hi), _lib.ptr(out), _lib.stream()),
               'dam_augment_gains_f32')
    return out


def batch_features(batch, device=None):
    """One call for caller-owned training loops (training_ignite.ipynb cell 12: ``train_features, gt_features = batch``):
    whatever a loader over MultitrackAudioDataset yields -> ``(train_features [B,S,F,T], gt_features [B,F,T])`` float32 dB
    on the GPU.  A HostPcmBatch (``DataLoader(dataset, num_workers>0)``: decoded clips in host memory) is uploaded and goes
    through ONE front-end launch; a PcmBatch (``batch_loader(pcm=True)``) through the launch only; a pair of feature
    tensors (``num_workers=0``, the feature cache, ``iter_batches``) is moved to the device if it is not there."""
    if hasattr(batch, 'to_device'):
        return batch.to_device(device).features()
    if hasattr(batch, 'clips'):
        return batch.features()
    x, gt = batch
    dev = torch.device(device if device is not None else 'cuda')
    return x.to(dev, torch.float32), gt.to(dev, torch.float32)
