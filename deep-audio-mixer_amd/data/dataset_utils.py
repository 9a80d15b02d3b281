"""Mirror of the reference's data/dataset_utils.py helpers that sit on the path."""
import os
import wave

import numpy as np


def scalar_amplitude_to_dB(x):
    """data/dataset_utils.py:39-43: 20 * log10(x)."""
    return 20 * np.log10(x)


def scalar_dB_to_amplitude(x):
    """data/dataset_utils.py:46-50: 10 ** (0.5 * x)  (sic -- not x/20; kept for parity with trained models)."""
    return np.power(10.0, 0.5 * x)


def read_wav(path, start=0, stop=None):
    """PCM WAV -> (float64 [frames, channels] in [-1, 1), sample rate), with a partial read like
    ``soundfile.read(path, start=, stop=)`` (data/dataset.py:194).  stdlib only (soundfile is not in the image);
    8/16/24/32-bit integer PCM, the formats MedleyDB / MUSDB18-HQ ship."""
    with wave.open(path, 'rb') as w:
        n, ch, width, sr = w.getnframes(), w.getnchannels(), w.getsampwidth(), w.getframerate()
        stop = n if stop is None else min(stop, n)
        start = min(max(start, 0), stop)
        w.setpos(start)
        raw = w.readframes(stop - start)
    if width == 2:
        a = np.frombuffer(raw, dtype='<i2').astype(np.float64) / 32768.0
    elif width == 4:
        a = np.frombuffer(raw, dtype='<i4').astype(np.float64) / 2147483648.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        a = (v - ((v & 0x800000) << 1)).astype(np.float64) / 8388608.0
    elif width == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    else:
        raise ValueError('unsupported sample width %d in %s' % (width, path))
    return a.reshape(-1, ch), sr


def wav_num_frames(path):
    with wave.open(path, 'rb') as w:
        return w.getnframes(), w.getframerate()


def load_tracks(base_path, song_name, tracklist=('bass', 'drums', 'vocals', 'other', 'mix')):
    """data/dataset_utils.py:53-68 for the MedleyDB layout: {track: ndarray[channels, n]}."""
    out = {}
    for track in tracklist:
        if track == 'mix':
            p = os.path.join(base_path, song_name, '%s_MIX.wav' % song_name)
        else:
            p = os.path.join(base_path, song_name, '%s_STEMS_JOINED' % song_name,
                             '%s_STEM_%s.wav' % (song_name, track.upper()))
        a, _ = read_wav(p)
        out[track] = a.T.copy()
    return out
