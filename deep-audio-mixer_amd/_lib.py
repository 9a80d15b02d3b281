"""ctypes binding of libdam_hip.so (include/dam_hip.h).  There is no CPU fallback: if the
library is missing or a call fails, the product path raises."""
import ctypes
import os

import torch
import torch.utils.data

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DAM_LIB_PATH') or os.path.join(_HERE, 'libdam_hip.so')   # override: diagnostic builds only

# include/dam_hip.h: bumped whenever a C signature changes (together with dam_abi_version() in csrc/dam_api.hip and
# DAM_ABI_VERSION in the header).  libdam_hip.so is git-ignored and travels prebuilt: a stale one would read device pointers
# as streams, so lib() refuses it instead of launching.
EXPECTED_ABI = 15

_STATUS = {0: 'DAM_OK', -1: 'DAM_ERR_BAD_ARG', -2: 'DAM_ERR_UNSUPPORTED', -3: 'DAM_ERR_LAUNCH',
           -4: 'DAM_ERR_WORKSPACE'}

c_i, c_i64, c_f, c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
c_d = ctypes.c_double

# name -> (restype, argtypes); kept in the order of include/dam_hip.h
SIGNATURES = {
    'dam_arch': (ctypes.c_char_p, []),
    'dam_abi_version': (c_i, []),
    'dam_host_dontfork_pinned': (c_i, [c_p, c_p]),
    'dam_step_mark_create': (c_i, [c_p]),
    'dam_step_mark_record': (c_i, [c_p, c_p]),
    'dam_step_mark_wait': (c_i, [c_p, c_p]),
    'dam_step_mark_synchronize': (c_i, [c_p]),
    'dam_step_mark_destroy': (c_i, [c_p]),
    'dam_stft_twiddle_count': (c_i64, [c_i]),
    'dam_stft_fill_twiddles_host': (c_i, [c_i, c_p]),
    'dam_stft_logmag_f32': (c_i, [c_p, c_i, c_i64, c_i64, c_i, c_i64, c_p, c_p, c_p, c_i, c_i, c_f, c_i, c_p, c_p]),
    'dam_stft_logmag_strided_f32': (c_i, [c_p, c_i, c_i64, c_i64, c_i64, c_i64, c_i64, c_i, c_i64, c_i64, c_p, c_p, c_p,
                                          c_i, c_i, c_f, c_i, c_p, c_p, c_i, c_p]),
    'dam_augment_gains_f32': (c_i, [ctypes.c_uint64, c_p, c_i64, c_i, c_i, c_f, c_f, c_p, c_p]),
    'dam_conv_packed_weight_count': (c_i64, [c_i, c_i, c_i, c_i]),
    'dam_conv_pack_weights_f32': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'dam_conv_pack_weights_multi_f32': (c_i, [c_p, c_i, c_i64, c_p]),
    'dam_conv2d_tapgrid_f32': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_p] +
                               [c_i] * 17 + [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p]),
    'dam_conv_batch_bytes': (c_i64, []),
    'dam_conv_batch_init': (c_i, [c_p]),
    'dam_conv_batch_flush': (c_i, [c_p, c_p]),
    'dam_conv2d_wgrad_workspace_floats': (c_i64, [c_i, c_i, c_i, c_i]),
    'dam_conv2d_wgrad_f32': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p] + [c_i] * 9 +
                             [c_p, c_i, c_p, c_i64, c_p, c_p]),
    'dam_wgrad_queue_bytes': (c_i64, []),
    'dam_wgrad_queue_init': (c_i, [c_p]),
    'dam_wgrad_queue_pending': (c_i, [c_p]),
    'dam_wgrad_queue_flush': (c_i, [c_p, c_p]),
    'dam_wgrad_queue_set_batching': (c_i, [c_p, c_i]),
    'dam_bn_workspace_floats': (c_i64, [c_i]),
    'dam_bn_stats_f32': (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'dam_bn_finalize_f32': (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p]),
    'dam_conv1x1_pair_f32': (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'dam_dgrad_s2_3x3_f32': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    'dam_conv_s2_pair_fwd_f32': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    'dam_bn_finalize_pair_f32': (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p]),
    'dam_strip_diag_counters': (c_i, [c_p, c_i]),
    'dam_bn_stats_partial_f32': (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p]),
    'dam_bn_finalize_apply_f32': (c_i, [c_p, c_i, c_i, c_p, c_p, c_i64, c_p, c_p, c_p, c_i, c_p, c_p, c_p]),
    'dam_bn_eval_affine_f32': (c_i, [c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p]),
    'dam_bn_apply_f32': (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p]),
    'dam_bn_stats_pair_f32': (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p, c_p, c_p]),
    'dam_bn_pair_workspace_floats': (c_i64, [c_i]),
    'dam_bn_backward_pair_f32': (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i] + [c_p] * 16),
    'dam_bn_backward_f32': (c_i, [c_p, c_p, c_p, c_i64, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    'dam_channel_sum_f32': (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p, c_p]),
    'dam_dropout_tick': (c_i, [c_p, c_i64, c_p, c_p]),
    'dam_dropout_apply_f32': (c_i, [c_p, c_i64, c_f, ctypes.c_uint64, c_p, c_p, c_p]),
    'dam_heads_fwd_f32': (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'dam_heads_bwd_workspace_floats': (c_i64, [c_i, c_i, c_i, c_i]),
    'dam_heads_bwd_f32': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'dam_masksum_fwd_f32': (c_i, [c_p, c_p, c_i, c_i, c_i64, c_p, c_p]),
    'dam_masksum_workspace_floats': (c_i64, [c_i, c_i]),
    'dam_masksum_bwd_f32': (c_i, [c_p, c_p, c_i, c_i, c_i64, c_p, c_p, c_p]),
    'dam_masksum_mse_f32': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i64, c_p, c_p, c_p, c_p, c_p]),
    'dam_adam_l2_step_f32': (c_i, [c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p]),
    'dam_gains_smooth': (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    'dam_gain_ramp_apply': (c_i, [c_p, c_i, c_p, c_i64, c_i64, c_i64, c_i, c_p, c_i, c_p]),
    'dam_mixdown_workspace_elems': (c_i64, [c_i64]),
    'dam_mixdown_peak_normalize': (c_i, [c_p, c_i, c_p, c_i, c_i64, c_i64, c_i, c_i, c_p, c_i, c_p, c_p]),
    'dam_nchw_to_nhwc16_f32': (c_i, [c_p, c_i, c_i, c_i64, c_p, c_p]),
    'dam_loudness_kweight_coeffs': (c_i, [c_d, c_p]),
    'dam_loudness_workspace_bytes': (c_i64, [c_i64, c_i]),
    'dam_loudness_block_energy': (c_i, [c_p, c_i, c_i64, c_i, c_i64, c_i64, c_p, c_p, c_p, c_i, c_d, c_p, c_p, c_p]),
}



class BnFin(ctypes.Structure):
    """struct dam_bn_fin (include/dam_hip.h): device pointers for an in-kernel BatchNorm finalize."""
    _fields_ = [('gamma', c_p), ('beta', c_p), ('running_mean', c_p), ('running_var', c_p), ('num_batches_tracked', c_p),
                ('momentum', c_f), ('eps', c_f), ('save_mean', c_p), ('save_invstd', c_p), ('scale', c_p), ('shift', c_p),
                ('counter', c_p)]


class BnBwdSums(ctypes.Structure):
    """struct dam_bn_bwd_sums (include/dam_hip.h): BatchNorm-backward sums taken in a data-gradient epilogue."""
    _fields_ = [('x', c_p), ('mean', c_p), ('invstd', c_p), ('mask_scale', c_p), ('mask_shift', c_p), ('res_mask_bits', c_p),
                ('mask_bits', c_p)]


_lib = None


def lib():
    """Loads libdam_hip.so once; raises (never falls back) if it is not built -- and in a process that must not touch the
    GPU (a DataLoader worker, any fork of a GPU-initialised parent): the HIP runtime does not survive a fork, a launch from
    there would hang or fault instead of failing."""
    global _lib
    if torch.cuda._is_in_bad_fork() or torch.utils.data.get_worker_info() is not None:
        raise RuntimeError('deep_audio_mixer_amd: a GPU kernel was asked for in a DataLoader worker / forked child process '
                           '(pid %d).  The GPU belongs to the parent: workers of MultitrackAudioDataset return host PCM '
                           '(PcmItem / HostPcmBatch) and the front-end runs in the main process -- see '
                           'features.batch_features(batch)' % os.getpid())
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libdam_hip.so is not built (%s): run `python __graft_entry__.py` or '
                               '`python deep-audio-mixer_amd/build.py`; there is no CPU fallback' % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        l.dam_abi_version.restype, l.dam_abi_version.argtypes = c_i, []
        have = l.dam_abi_version()
        if have != EXPECTED_ABI:
            raise RuntimeError('%s speaks ABI version %d, this package expects %d: the library is stale -- rebuild it '
                               '(`python deep-audio-mixer_amd/build.py --force`)' % (LIB_PATH, have, EXPECTED_ABI))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(status, what):
    if status != 0:
        raise RuntimeError('%s failed: %s (%d)' % (what, _STATUS.get(status, '?'), status))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('deep_audio_mixer_amd kernels run on the GPU only (got a %s tensor); '
                               'there is no CPU fallback' % t.device)
