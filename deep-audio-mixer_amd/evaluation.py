"""Loudness evaluation of a mix against a reference mix -- the numeric core of the reference's evaluation.py:20-75
(``LoudnessEvaluator``): per-stem BS.1770 loudness relative to the stems' mean, and the mean absolute difference of two
such profiles (the paper's "loudness error").  Same method names and argument meaning; the meter is the HIP one
(loudness.Meter), stems stay on the device.  The reference's spreadsheet / WAV export and its experiment driver
(evaluation.py:77-end: openpyxl, soundfile, MUSDB loaders) are callers of this and are not part of the path.
"""
from collections import OrderedDict
from statistics import mean

import numpy as np
import torch

from .loudness import Meter, normalize_loudness


class LoudnessEvaluator:
    def __init__(self, sr=44100, keys=('bass', 'drums', 'vocals', 'other')):
        self.sr = sr
        self.meter = Meter(sr)
        self.keys = tuple(keys)

    def evaluate_loudness(self, tracks: dict) -> list:
        """evaluation.py:39-46: loudness of every stem ([channels, samples] each) minus the mean over the stems."""
        per_track_loudness = [self.meter.integrated_loudness(tracks[name].T) for name in self.keys]
        avg_loudness = mean(per_track_loudness)
        return [l - avg_loudness for l in per_track_loudness]

    @staticmethod
    def _calculate_diff_between_loudness_dicts(l_dict1: OrderedDict, l_dict2: OrderedDict):
        """evaluation.py:48-53."""
        a1 = np.array(list(l_dict1.values()))
        a2 = np.array(list(l_dict2.values()))
        return float(np.mean(np.abs(a1 - a2)))

    def sum_tracks_to_target(self, track_dict: dict, target_lufs: float = -20.0):
        """evaluation.py:59-66 without the file write: stem sum, measured, brought to target_lufs."""
        stems = [track_dict[k] for k in track_dict]
        if torch.is_tensor(stems[0]):
            track_sum = torch.stack(stems).sum(dim=0)
        else:
            track_sum = np.sum(np.array(stems), axis=0)
        loudness = self.meter.integrated_loudness(track_sum.T)
        return normalize_loudness(track_sum.T, loudness, target_lufs)

    def _sum_and_evaluate_tracks(self, track_dict, reference_dict):
        """evaluation.py:55-75: (loudness profile, error against reference_dict or None)."""
        loudness_dict = OrderedDict(zip(self.keys, self.evaluate_loudness(track_dict)))
        if reference_dict:
            return loudness_dict, self._calculate_diff_between_loudness_dicts(loudness_dict, reference_dict)
        return loudness_dict, None
