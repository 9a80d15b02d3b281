"""Loudness-normalisation baseline: every stem is brought to the mean loudness of that stem over the training set.
Same surface as the reference's models/baselines/mean_loudness_model.py:6-22; the meter is the HIP BS.1770 meter."""
from ...loudness import Meter, normalize_loudness


class MeanLoudnessModel:
    def __init__(self, d_mean_loudness: dict, sr=44100):
        self.mean_loudness = d_mean_loudness
        self.meter = Meter(sr)
        self.tracklist = ('bass', 'drums', 'vocals', 'other')

    def forward(self, x: dict) -> dict:
        result = {}
        for name in self.tracklist:
            track = x[name]                                  # [channels, samples], as load_tracks_* returns it
            measured = self.meter.integrated_loudness(track.T)
            result[name] = normalize_loudness(track.T, measured, self.mean_loudness[name]).T
        return result
