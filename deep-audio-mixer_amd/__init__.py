"""deep-audio-mixer hot path on AMD MI355X (gfx950).

Host-side mirror of the reference interface (same module paths, names and argument
meaning as apelykh/deep-audio-mixer) over the C-ABI library ``libdam_hip.so``
(include/dam_hip.h).  PyTorch-ROCm supplies device memory, streams, autograd plumbing
and torch.distributed; every kernel on the path is hand-written HIP.

    from deep_audio_mixer_amd.data.dataset import MultitrackAudioDataset
    from deep_audio_mixer_amd.models.model_resnet import ResNet18
    from deep_audio_mixer_amd.model_trainer import ModelTrainer
    from deep_audio_mixer_amd.inference_utils import mix_song_smooth
"""
__version__ = '0.1.0'
