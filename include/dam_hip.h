/*
 * dam_hip.h -- C ABI of libdam_hip.so, the MI355X (gfx950) implementation of the
 * deep-audio-mixer hot path.
 *
 * The reference (apelykh/deep-audio-mixer) has no FFI: its hot path is reached through
 * torch.stft / torch.nn on one device.  Each entry point below names the reference
 * code (file:line, relative to the reference root) whose arithmetic it replaces; the
 * Python mirror of the reference interface (deep-audio-mixer_amd/) binds these through
 * ctypes, see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in
 *     _host; the library allocates nothing and keeps no state;
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*), never
 *     synchronises, and is hipGraph-capturable;
 *   - returns DAM_OK (0) or a negative dam_status; never throws across the ABI;
 *   - activations are NHWC float32 ("pixels x channels", channel count a multiple of 16
 *     unless stated); tensors the reference API exposes (features, masked, gains) are in
 *     the reference's own layout ([B,S,F,T] / [B,F,T] / [B,S]).
 */
#ifndef DAM_HIP_H
#define DAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dam_status {
    DAM_OK = 0,
    DAM_ERR_BAD_ARG = -1,        /* null pointer, non-positive size, N <= n_fft/2 ... */
    DAM_ERR_UNSUPPORTED = -2,    /* shape/dtype outside what the gfx950 kernels implement */
    DAM_ERR_LAUNCH = -3,         /* hipGetLastError() != hipSuccess after the launch */
    DAM_ERR_WORKSPACE = -4       /* caller-provided workspace too small */
} dam_status;

/* Sample types the front-end reads.  DAM_PCM_S16 / DAM_PCM_S32: integer PCM exactly as the WAV file holds it (interleaved or
 * mono), scaled by 1/2^15 / 1/2^31 as soundfile.read does (data/dataset.py:194) -- no host conversion, half the PCIe bytes
 * for 16-bit files; results are bit-identical to DAM_PCM_F32 on host-converted samples.  24-bit files: the reader unpacks
 * them to left-justified DAM_PCM_S32 (value << 8).  Strides are counted in samples of the given type. */
typedef enum dam_pcm_dtype { DAM_PCM_F32 = 0, DAM_PCM_F64 = 1, DAM_PCM_S16 = 2, DAM_PCM_S32 = 3 } dam_pcm_dtype;
/* OR-ed into pcm_dtype: `pcm` is a DEVICE word holding the address of the PCM (const void* const*), read when the kernel
 * starts.  A launch captured in a hipGraph then follows whichever resident batch the word points at -- the caller re-points
 * the word (8 bytes) instead of copying a batch into the graph's fixed input buffer.  Layout arguments describe the pointee. */
#define DAM_PCM_INDIRECT 0x100
/* OR-ed into pcm_dtype instead: `pcm` is a DEVICE table of int64 words {address of an int64 step counter, n, offset, addr[0] ...
 * addr[n-1]} and the launch reads the batch at addr[(counter + offset) % n].  With the optimizer's device-side step count as the
 * counter (dam_adam_l2_step_f32 advances it inside the captured step) a replayed step walks n resident batches -- or the n
 * staging buffers of an uploader -- with NO launch between the replays: re-pointing the DAM_PCM_INDIRECT word was a 4.8 us fill
 * launch behind an 8 us gap per step (profiles/r05_C3_step_timeline.txt, the last line).  n >= 1, counter + offset >= 0. */
#define DAM_PCM_ROTATE 0x200
struct dam_bn_fin;      /* defined in the BatchNorm section */
struct dam_bn_bwd_sums; /* defined in the BatchNorm section */

/* Library / build identification ("gfx950").  DAM_ABI_VERSION is bumped whenever a signature below changes; a binding
 * compares dam_abi_version() of the library it loaded with the version it was written against and refuses a stale one
 * (deep-audio-mixer_amd/_lib.py: EXPECTED_ABI). */
#define DAM_ABI_VERSION 15
const char* dam_arch(void);
int dam_abi_version(void);

/* Host-side fork guard for the loaders of the reference's notebooks: training.ipynb cell 6 / training_ignite.ipynb cell 6 build
 * DataLoader(d_train, num_workers=6, pin_memory=True) over data/dataset.py:270-292, i.e. the process that owns the GPU forks
 * six workers at the start of EVERY epoch.  On this stack page-locked host memory (hipHostMalloc: torch's pinned tensors,
 * the runtime's staging buffers) is anonymous memory registered with the GPU as user pointers and NOT marked MADV_DONTFORK:
 * a fork write-protects it for copy-on-write, the driver's MMU notifier evicts the process's GPU queues and restores them
 * only later -- measured 0.2-0.6 s of GPU stall per fork batch in a toy process, 4-5 s per epoch in the C3 training loop
 * (profiles/r05_fork_stall.txt).  This call marks the private anonymous mappings of the calling process that ARE page-locked
 * data buffers handed out by the HIP layer (hipPointerGetAttributes: hipMemoryTypeHost; extent from hsa_amd_pointer_info)
 * MADV_DONTFORK: children forked afterwards do not inherit those buffers (they cannot use the GPU anyway) and the fork no
 * longer write-protects them.  The runtime's internal pools are left alone (a child that runs a destructor of an inherited
 * GPU object reads them).  No device pointers, no stream, nothing enqueued; call it from the GPU-owning process only, after
 * the runtime is initialised.
 *   n_mappings_host / n_bytes_host : optional outputs, ranges and bytes marked by this call (re-marking is harmless). */
int dam_host_dontfork_pinned(int64_t* n_mappings_host, int64_t* n_bytes_host);

/* Step marks (ABI 15): a point INSIDE a captured training step that another stream can wait for.  The reference uploads every
 * batch on the training stream, `train_features.to(self.device)` at model_trainer.py:34 and `gt_features.to(...)` at :35, from
 * the page-locked batches of DataLoader(pin_memory=True) (training.ipynb cell 6).  Here the step is ONE hipGraph and the upload
 * of batch k+1 runs on a copy stream beside step k; measured (profiles/r05_pcie_trace.txt), a 1.3 ms host-to-device copy beside
 * the FORWARD half of the step costs its latency-bound launches (the BatchNorm finalize kernels: 4.7 -> 7.7 us each) 0.13 ms per
 * step, beside the shallow layers' backward it costs a third of that.  A mark is a HIP event recorded with
 * hipEventRecordExternal while the step is being captured (an event-record node of the graph, re-recorded by every replay;
 * outside a capture: an ordinary record), so the copy stream waits for "step k has reached its backward pass" -- the graph is
 * not cut.  The front-end has read the step's PCM long before any mark of the step, so the same wait also says the previous
 * staging buffer is free.
 *   dam_step_mark_create : *mark_host receives the handle (timing disabled).
 *   dam_step_mark_record : on `stream`; captured as an external event-record node when the stream is capturing.
 *   dam_step_mark_wait   : makes `stream` (never a capturing one: DAM_ERR_BAD_ARG) wait for the latest record that has been
 *                          enqueued -- directly or through a graph launch -- before this call; a no-op if there is none.
 *   dam_step_mark_synchronize: the HOST waits for that record.  This is the form the uploader uses: measured on this stack
 *                          (profiles/r05_sync_cost_probe.txt), ANY stream that waits for an event of the training stream costs
 *                          the training stream 0.09 ms per step (4.125 -> 4.212 ms; event flags make no difference), a host
 *                          wait costs nothing (4.138-4.143 against 4.139-4.145 ms) -- so the host waits for the mark and then
 *                          enqueues the copy, and only the cheap direction (the training stream waiting for the copy stream's
 *                          event) stays on the device.
 *   dam_step_mark_destroy: after the graphs that hold the mark are gone. */
int dam_step_mark_create(void** mark_host);
int dam_step_mark_record(void* mark, void* stream);
int dam_step_mark_wait(void* mark, void* stream);
int dam_step_mark_synchronize(void* mark);
int dam_step_mark_destroy(void* mark);

/* ---------------------------------------------------------------------------------
 * Feature front-end.  Replaces data/dataset.py:132-162 (compute_features: torch.stft ->
 * abs -> amplitude_to_DB), :181-183 (_stereo_to_mono), :164-168 (_augment_audio) and the
 * stacking at :207-210, for a whole batch of tracks in one launch.
 * --------------------------------------------------------------------------------- */

/* Host helper: number of float2 entries of the twiddle table for n_fft (n_fft entries:
 * W_{n_fft}^k = exp(-2 pi i k / n_fft)), and a routine that fills it (computed in double). */
int64_t dam_stft_twiddle_count(int n_fft);
int dam_stft_fill_twiddles_host(int n_fft, float* table_host /* [2*count] re,im */);

/* out[track][f][t] = 20*log10(max(|STFT(mean_c pcm[track][:, c] * gain[track])|, amin))
 *   pcm      : [n_tracks][n_samples][channels], dtype per pcm_dtype, interleaved channels;
 *              track k starts at pcm + k*pcm_track_stride elements
 *   window   : [n_fft] float32 (the caller uploads torch.hann_window(n_fft))
 *   twiddles : table from dam_stft_fill_twiddles_host
 *   gain     : optional [n_tracks] float32 (nullptr = 1), the augmentation draw
 *   normalize: 0, or 1 = divide every frame (column) by its max-abs over bins
 *              (librosa.util.normalize semantics of the call disabled at data/dataset.py:159-160)
 *   out      : [n_tracks][n_fft/2+1][T] float32, T = 1 + n_samples/hop
 * center=True / reflect padding / onesided / unnormalised, as torch.stft's defaults.
 * Supported: channels 1 or 2, n_samples > n_fft/2, hop >= 1; n_fft == 2048 with an even hop (every call site of the
 * reference: data/dataset.py:132-133 defaults) runs the tuned in-register kernel, any other power of two from 64 to 16384
 * (or an odd hop) a generic one-workgroup-per-frame kernel; window then has n_fft entries (torch.hann_window(n_fft)). */
int dam_stft_logmag_f32(const void* pcm, int pcm_dtype, int64_t n_tracks, int64_t n_samples, int channels,
                        int64_t pcm_track_stride, const float* window, const float* twiddles,
                        const float* gain, int n_fft, int hop, float amin, int normalize,
                        float* out, void* stream);

/* The same front-end over strided PCM, so that tracks need not be gathered first.  Track (o, i), o < n_outer,
 * i < n_inner, is output row o*n_inner + i and starts at pcm + o*outer_stride + i*inner_stride (elements); sample p of
 * channel c is at + p*sample_stride + c*channel_stride.  Two layouts are implemented: interleaved channels
 * (sample_stride == channels, channel_stride == 1: what a WAV decoder / soundfile hands over, data/dataset.py:194) and
 * planar channels (sample_stride == 1, channel_stride >= n_samples: the [channels, n] arrays that
 * inference_utils.py:116-119 slices chunk by chunk -- outer = chunk (stride chunk_samples), inner = stem).
 * gain (optional) has one entry per track, index o*n_inner + i.
 * n_tail > 0 splits the output the way a dataset item is split (data/dataset.py:207-210: train_features = the stems,
 * gt_features = the mix): tracks i < n_inner - n_tail go to out [n_outer][n_inner - n_tail][F][T], the last n_tail
 * tracks of every outer group to out_tail [n_outer][n_tail][F][T] -- one launch for a whole batch of clips. */
int dam_stft_logmag_strided_f32(const void* pcm, int pcm_dtype, int64_t n_outer, int64_t outer_stride, int64_t n_inner,
                                int64_t inner_stride, int64_t n_samples, int channels, int64_t sample_stride,
                                int64_t channel_stride, const float* window, const float* twiddles, const float* gain,
                                int n_fft, int hop, float amin, int normalize, float* out, float* out_tail, int n_tail,
                                void* stream);

/* The augmentation draw of data/dataset.py:164-168,198-199 (one uniform gain per track of an item, the mix included)
 * made on the device and reproducibly: gains[i][k] = lo + (hi - lo) * u(seed, item_i, k), u in [0, 1) a counter-based
 * hash (splitmix64) of (seed, GLOBAL item index, track).  items: optional DEVICE int64[n_items] of global item (chunk)
 * indices; nullptr = first_item, first_item + 1, ...  gains [n_items][n_tracks] feeds dam_stft_logmag_*'s `gain`. */
int dam_augment_gains_f32(uint64_t seed, const int64_t* items, int64_t first_item, int n_items, int n_tracks,
                          float lo, float hi, float* gains, void* stream);

/* ---------------------------------------------------------------------------------
 * Convolutions.  Replace nn.Conv2d forward and its autograd data-gradient in
 * models/model_resnet.py:11-21,64 (BasicBlock / stem convs) and
 * models/model_scalar_1s.py:167-172, models/model_scalar_2s.py:25-30 (ConvBlock2d.conv).
 * Implicit GEMM on the fp32 matrix cores; activations NHWC float32.
 * --------------------------------------------------------------------------------- */

/* Number of floats of the packed weight image for an operator with n_out output and k_in
 * reduction channels: [kh*kw][ceil(k_in/16)][ceil(n_out/16)][4][16][4]. */
int64_t dam_conv_packed_weight_count(int n_out, int k_in, int kh, int kw);

/* Pack torch-layout weights [O][I][KH][KW] for the forward (transpose=0: n_out=O, k_in=I)
 * or the data-gradient operator (transpose=1: n_out=I, k_in=O); channels are zero-padded to 16. */
int dam_conv_pack_weights_f32(const float* w_oihw, int O, int I, int KH, int KW, int transpose,
                              float* packed, void* stream);

/* The same packing for many tensors in one launch.  desc_dev: DEVICE array [n_tensors][8] of int64 =
 * {w_oihw pointer, packed pointer, O, I, KH, KW, transpose, packed float count}; max_total = largest count. */
int dam_conv_pack_weights_multi_f32(const int64_t* desc_dev, int n_tensors, int64_t max_total, void* stream);

/* y[b, oh*out_stride+out_off_h, ow*out_stride+out_off_w, :] = g( res [* (res_mask > 0)] +
 *     bias + sum_{a<nA, b<nB, k} Wp[wt_base + a*wt_sa + b*wt_sb][k][:] *
 *            f(x[b, oh*in_stride + off_h + a*step_h, ow*in_stride + off_w + b*step_w, k])
 * for oh < Ho, ow < Wo, with x = 0 outside [0,H)x[0,W) and f(v) = v, or v*in_scale[k]+in_shift[k]
 * (then ReLU if relu_in; without it the loaders clamp at -inf, so a NaN input reads as -inf) -- the producer's BatchNorm apply fused into the load -- and g(v) = v, or max(v, 0) if relu_out:
 * with an eval-mode BatchNorm folded into the weights (scale) and `bias` (shift), one launch is the reference's
 * relu(bn(conv(x)) [+ shortcut]) (models/model_resnet.py:23-28,97).
 *   x : NHWC [B][H][W][C] (C % 16 == 0, k_chunks == C/16), or with in_nchw=1 the reference's
 *       [B][C][H][W] planes with C <= 16 (first layer, k_chunks == 1)
 *   y : NHWC [B][OHt][OWt][n_out], n_out % 16 == 0;  res/res_mask (optional): same shape as y
 * Forward conv: in_stride=stride, off=-pad, step=dilation, taps (kh,kw) as is.  Stride-1 dgrad:
 * off=+pad, step=-dilation on dy with transposed packing.  Stride-2 dgrad: one call per output parity
 * class (out_stride=2).  in_stride must be 1 or 2.
 * bn_partial (optional, >= dam_bn_workspace_floats(n_out) floats): if the launch can also produce the BatchNorm
 * partial statistics of y (records (n, mean, M2) per workgroup and channel) it does so and stores the record count
 * in *bn_parts_host (a HOST int); 0 there means "not produced" and the caller runs dam_bn_stats_f32 instead.
 * Feed the records to dam_bn_finalize_f32 -- or pass bn_fin (see dam_bn_fin below): the launch's last workgroup then
 * merges them itself and writes save_mean / save_invstd / scale / shift (+ running statistics), no finalize launch.
 * bn_bwd (optional, see dam_bn_bwd_sums below; excludes bn_fin, needs bn_partial; with res only if res_mask_bits is set there): the launch is a data gradient whose
 * output dy feeds the backward pass of y = relu(bn(x)); if it can, it also writes that pass's two per-channel sums as records
 * [*bn_parts_host][n_out][2] into bn_partial (then hand them to dam_bn_backward_f32 as partials_given); 0 records = not produced.
 * workspace (optional): scratch for split-K over the input channels (used for small-spatial, wide layers; at most
 * 8 * B*OHt*OWt*n_out floats are used, fewer if less is given); partial slabs are summed in a fixed order. */
int dam_conv2d_tapgrid_f32(const float* x, int B, int H, int W, int C, int in_nchw, const float* w_packed,
                           int k_chunks, int n_out, const float* bias, const float* in_scale,
                           const float* in_shift, int relu_in, int relu_out, float* y, int OHt, int OWt, int Ho, int Wo,
                           int out_stride, int out_off_h, int out_off_w, int in_stride, int nA, int nB,
                           int off_h, int step_h, int off_w, int step_w, int wt_base, int wt_sa, int wt_sb,
                           const float* res, const float* res_mask, float* bn_partial, int* bn_parts_host,
                           const struct dam_bn_fin* bn_fin, const struct dam_bn_bwd_sums* bn_bwd, float* workspace,
                           int64_t workspace_floats, void* batch, void* stream);

/* Two single-tap operators into the same output pixels in ONE launch:
 *   y[b, oh*out_stride+out_off_h, ow*out_stride+out_off_w, :] = W1p[tap1] . x1[b, oh, ow, :] + W2p[tap2] . x2[b, oh, ow, :]
 * x1, x2: NHWC [B][H][W][C]; W1p / W2p: packed images (dam_conv_pack_weights_f32) with k_in = C and n_out outputs, of which tap
 * tap1 / tap2 is used.  The data gradient of a down-sampling block's input (models/model_resnet.py:17-21,26: x feeds the 3x3 /
 * stride-2 conv1 AND the 1x1 / stride-2 shortcut convolution) receives, at its (even, even) pixels, exactly these two terms: the
 * centre tap of conv1's transposed operator applied to d conv1 and the shortcut's transposed operator applied to d shortcut. */
int dam_conv1x1_pair_f32(const float* x1, const float* w1_packed, int tap1, const float* x2, const float* w2_packed, int tap2,
                         int B, int H, int W, int C, int n_out, float* y, int OHt, int OWt, int out_stride, int out_off_h,
                         int out_off_w, void* stream);

/* The whole data gradient of a 3x3 / stride-2 / pad-1 convolution's input in ONE launch -- all four output parity classes from
 * one read of dy -- optionally with a second, 1x1 / stride-2 operator of the same input added at the (even, even) pixels (a
 * down-sampling block: conv1 and the shortcut convolution both read x, models/model_resnet.py:17-21,26):
 *   dx[b, u, v, :] = sum_{a,b} W[a][b]^T dy[b, (u+1-a)/2, (v+1-b)/2, :]  (integer quotients only)  [+ Wp^T dy_pair[b, u/2, v/2, :]]
 * dy, dy_pair: NHWC [B][Hd][Wd][Co] with Hd = (H+1)/2, Wd = (W+1)/2; w_packed_t / w_pair_packed_t: dam_conv_pack_weights_f32
 * images with transpose = 1 (nine taps / one tap); dx: NHWC [B][H][W][Ci].  Every byte of dx is written.
 * Co = 32 -> Ci = 16 and Co = 64 -> Ci = 32 keep the packed weights in LDS for the life of a persistent workgroup; other layers
 * with Co % 32 == 0 and Ci % 16 == 0 stream the weight fragments from L2 (one (pixel block, channel block) unit per wave).
 * DAM_ERR_UNSUPPORTED for anything else; the caller then runs the parity classes through dam_conv2d_tapgrid_f32.
 * bn_bwd / bn_partial / bn_parts_host (optional, all three or none; see dam_bn_bwd_sums below, `mask_bits` form only): dx is the
 * gradient reaching relu(bn(u) + shortcut), the output of the block in front (models/model_resnet.py:26-27); the two thin layers'
 * kernels then also write that BatchNorm's two backward sums as records [*bn_parts_host][Ci][2] into bn_partial (hand them to
 * dam_bn_backward_f32 as partials_given); *bn_parts_host == 0: not produced, run the separate pass. */
int dam_dgrad_s2_3x3_f32(const float* dy, const float* w_packed_t, const float* dy_pair, const float* w_pair_packed_t, int B,
                         int Hd, int Wd, int Co, int Ci, float* dx, int H, int W, const struct dam_bn_bwd_sums* bn_bwd,
                         float* bn_partial, int* bn_parts_host, void* stream);

/* Forward of the same block: conv1 (3x3 / stride 2 / pad 1, no bias) and the shortcut convolution (1x1 / stride 2) of one input
 * in ONE launch, with the BatchNorm statistics records of both outputs (models/model_resnet.py:17-21,24-26: both feed a
 * training-mode BatchNorm).  x: NHWC [B][H][W][Ci]; w_packed / wsc_packed: dam_conv_pack_weights_f32 images with transpose = 0;
 * y, ysc: NHWC [B][(H+1)/2][(W+1)/2][Co].  partial / partial_sc (both or neither; >= dam_bn_workspace_floats(Co) floats each):
 * records [*parts_host][Co][3] = (n, mean, M2), one per workgroup, for dam_bn_finalize_pair_f32 / dam_bn_finalize_f32 /
 * dam_bn_finalize_apply_f32.  Ci = 16 -> Co = 32 and Ci = 32 -> Co = 64 keep both weight images resident in LDS (one record per
 * workgroup); other layers with Ci % 32 == 0 and Co % 16 == 0 stream the weight fragments from L2 (one (16 pixels, 16 channels) unit
 * per wave, one record per 64 pixels); DAM_ERR_UNSUPPORTED otherwise: the caller runs dam_conv2d_tapgrid_f32 twice and
 * dam_bn_stats_pair_f32. */
int dam_conv_s2_pair_fwd_f32(const float* x, const float* w_packed, const float* wsc_packed, int B, int H, int W, int Ci, int Co,
                             float* y, float* ysc, float* partial, float* partial_sc, int* parts_host, void* stream);

/* Sibling launches in one.  The parity classes of a strided data gradient are up to four small launches over the same
 * tensors, weights and tile that differ only in their tap grid.  With a batch (caller-owned HOST memory of
 * dam_conv_batch_bytes() bytes, dam_conv_batch_init() once) dam_conv2d_tapgrid_f32 records a launch that would take the tile
 * kernel instead of making it (launches that take another kernel run at once, as without a batch);
 * dam_conv_batch_flush packs consecutive records that differ only in geometry into ONE launch (class in blockIdx.z).
 * Every tensor of a recorded call must stay valid until the flush; batch = NULL launches at once. */
int64_t dam_conv_batch_bytes(void);
int dam_conv_batch_init(void* batch);
int dam_conv_batch_flush(void* batch, void* stream);

/* Weight gradient of the same convolutions (autograd of nn.Conv2d reached from loss.backward(),
 * model_trainer.py:36):  dw[n][k][kh][kw] = sum_{b,oh,ow} dy[b,oh,ow,n] * f(x[b, oh*stride+kh*dil-pad, ow*stride+kw*dil-pad, k])
 *   x  : as for dam_conv2d_tapgrid_f32 (NHWC, or in_nchw planes for the first layer), same fused f()
 *   dy : NHWC [B][Ho][Wo][n_chan], n_chan % 16 == 0; only the first n_out channels are real
 *   dw : torch layout [n_out][c_real][kh][kw], fully overwritten; c_real <= C is the number of REAL input channels
 *        (0 = C): the zero-padded planes of the re-laid stem input (dam_nchw_to_nhwc16_f32) drop out, so the result can
 *        be written straight into the parameter's slice of a flat gradient buffer
 *   workspace : at least dam_conv2d_wgrad_workspace_floats(...) floats; holds the split-K slabs that a
 *               second kernel sums in a fixed order (bitwise reproducible, no float atomics)
 * Supported kernels: 3x3, 1x1, and kw in {5,7,9} with any kh; stride 1 or 2. */
int64_t dam_conv2d_wgrad_workspace_floats(int n_out, int c_in, int kh, int kw);
int dam_conv2d_wgrad_f32(const float* x, int B, int H, int W, int C, int in_nchw, const float* in_scale,
                         const float* in_shift, int relu_in, const float* dy, int Ho, int Wo, int n_chan,
                         int n_out, int kh, int kw, int stride, int pad, int dil, float* dw, int c_real,
                         float* workspace, int64_t workspace_floats, void* reduce_queue, void* stream);

/* Deferred slab reductions.  Nothing in loss.backward() (model_trainer.py:36) reads a weight gradient before
 * optimizer.step() (model_trainer.py:37), and the reduction that turns a convolution's slabs into dw is a few
 * microseconds of mostly launch latency -- twenty of them in a ResNet18 step.  With a queue,
 * dam_conv2d_wgrad_f32 launches only the slab kernel and records the reduction; dam_wgrad_queue_flush runs every
 * recorded reduction in ONE launch (descriptors travel as kernel arguments: nothing to upload, hipGraph-capturable).
 *   queue : caller-owned HOST memory of dam_wgrad_queue_bytes() bytes, dam_wgrad_queue_init() once;
 *           reduce_queue = NULL in dam_conv2d_wgrad_f32 reduces at once (dw valid when the call's work is done)
 *   Until the flush, every queued call's `workspace` (its slabs) and `dw` must stay allocated and untouched, so
 *   queued calls need distinct workspaces; dw is written by the flush.  A full queue flushes itself on `stream`. */
int64_t dam_wgrad_queue_bytes(void);
int dam_wgrad_queue_init(void* queue);
int dam_wgrad_queue_pending(const void* queue);     /* recorded, not yet flushed; -1: not an initialised queue */
int dam_wgrad_queue_flush(void* queue, void* stream);
/* Batched slab launches (ABI 14).  The deep ResNet stages (models/model_resnet.py:70-71: 128 / 256 channels on 65x9 / 33x5 pixels)
 * have three weight gradients of ONE geometry each, 20-28 us launches of which ~13 us are launch + pipeline fill, and each split
 * eight ways over the pixels to fill 256 CUs (eight 2.4 MB slabs for a 2.4 MB gradient).  With batching on, a queued
 * dam_conv2d_wgrad_f32 that takes the tile kernel only RECORDS its launch; recorded launches of the same instantiation and geometry
 * (at most 4) run as ONE launch (blockIdx.z = job) when another geometry arrives or at dam_wgrad_queue_flush -- with a third of
 * the pixel splits, i.e. a third of the slab bytes the reduction reads back.  Calls that take the DIRECT kernel (the 1x1 / stride-2
 * shortcut convolutions, models/model_resnet.py:18-21: 11-16 us launches of a latency-bound kernel) are recorded too and run, up
 * to 8 of one instantiation with whatever geometries, as one flat launch whose jobs share the chip (each a fraction of the pixel
 * splits it would take alone).  Off by default.
 *   Contract while on: x, dy, in_scale / in_shift of a queued call must stay valid and unmodified until the flush as well
 *   (not only its workspace and dw).  Switching needs an empty queue (DAM_ERR_BAD_ARG otherwise). */
int dam_wgrad_queue_set_batching(void* queue, int on);

/* Diagnostic builds of the library only (libdam_hip_diag.so: dam_conv_strip.hip compiled with -DDAM_STRIP_DIAG_TAGS): the row-ring
 * convolution's loader waves tag every geometry-table entry with the tile it describes and its compute waves check the tag of every
 * entry they consume (the slip-bound argument in dam_conv_strip.hip, checked on the device).  out2_host receives {checks made,
 * mismatches seen} since the last reset; synchronises the device.  The shipped library returns DAM_ERR_UNSUPPORTED. */
int dam_strip_diag_counters(uint32_t* out2_host, int reset);

/* ---------------------------------------------------------------------------------
 * BatchNorm2d on NHWC float32 [n_pixels][C], C % 16 == 0.  Replaces nn.BatchNorm2d + F.relu (+ the
 * residual add) of models/model_resnet.py:12-27,65,97 and models/model_scalar_1s.py:174-186,
 * models/model_scalar_2s.py:32-44, forward and backward.
 * --------------------------------------------------------------------------------- */
int64_t dam_bn_workspace_floats(int C);

/* "The last workgroup finalizes": the kernels that produce per-workgroup partial records (BatchNorm statistics from a
 * convolution epilogue or from dam_bn_stats_f32, the two sums of dam_bn_backward_f32) can merge them themselves in the
 * workgroup that happens to finish last, instead of a separate 5-7 us finalize launch.  That needs one device word:
 * `counter` -- ZERO-INITIALISED by the caller, returned to zero by every launch, shared only by launches that are
 * ordered on one stream.  counter == NULL keeps the two-launch form. */
typedef struct dam_bn_fin {       /* host struct of device pointers: what dam_bn_finalize_f32 takes, for an in-kernel finalize */
    const float* gamma; const float* beta;
    float* running_mean; float* running_var; int64_t* num_batches_tracked;    /* may be NULL */
    float momentum, eps;
    float* save_mean; float* save_invstd; float* scale; float* shift;
    uint32_t* counter;
} dam_bn_fin;

/* BatchNorm-backward sums from a data-gradient epilogue (dam_conv2d_tapgrid_f32's bn_bwd): the backward pass of
 * y = relu(bn(x)) (models/model_resnet.py:24 of the reference, reached from loss.backward()) needs sum(dz) and sum(dz * xhat)
 * per channel, dz = dy * (x*mask_scale + mask_shift > 0), xhat = (x - mean) * invstd.  dy is the convolution data gradient
 * that was just computed, so the launch that produces it takes the sums from its output registers and one read of x.
 * Host struct of device pointers (x: the BatchNorm's input, shape of the launch's output; the rest: per channel). */
typedef struct dam_bn_bwd_sums {
    const float* x; const float* mean; const float* invstd; const float* mask_scale; const float* mask_shift;
    /* A data gradient that also carries an identity shortcut (res / res_mask given: dy = conv^T(..) + res * (res_mask > 0), the
     * gradient reaching the block INPUT) can take the sums of the BatchNorm that produced that input -- the ResNet stem's
     * relu(bn1(conv1(x))), models/model_resnet.py:97 -- if the shortcut's mask is also available as the sign bytes
     * dam_bn_apply_f32 wrote (one byte per channel quad); res_mask stays the fallback of launches that cannot. */
    const uint8_t* res_mask_bits;
    /* mask_bits (instead of mask_scale / mask_shift): the BatchNorm's ReLU mask as sign bytes -- the upstream layer is
     * relu(bn(x) + shortcut), a residual block's bn2 (models/model_resnet.py:26-27), whose mask is that of the block output.
     * Only together with res / res_mask_bits. */
    const uint8_t* mask_bits;
} dam_bn_bwd_sums;

/* Training-mode statistics of x: save_mean, save_invstd = 1/sqrt(biased var + eps), the fused affine
 * scale = gamma*invstd, shift = beta - mean*scale, and torch's running-stat update
 * (running = (1-momentum)*running + momentum*stat, unbiased variance; ++*num_batches_tracked).
 * running_mean/running_var/num_batches_tracked may be NULL. */
int dam_bn_stats_f32(const float* x, int64_t n_pixels, int C, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     float momentum, float eps, float* save_mean, float* save_invstd, float* scale,
                     float* shift, float* workspace, uint32_t* counter, void* stream);

/* dam_bn_stats_f32 for TWO tensors of one shape in one partial + one finalize launch (a residual block's conv1 output and its
 * shortcut convolution's output, models/model_resnet.py:17-21,24-26: two independent BatchNorms that become ready together).
 * a, b: the parameters / outputs of each (struct dam_bn_fin below; `counter` is ignored); workspace:
 * 2 * dam_bn_workspace_floats(C) floats. */
int dam_bn_stats_pair_f32(const float* x_a, const float* x_b, int64_t n_pixels, int C, const struct dam_bn_fin* a,
                          const struct dam_bn_fin* b, float* workspace, void* stream);

/* Second half of dam_bn_stats_f32 on its own: merges `parts` partial records [parts][C][3] = (n, mean, M2) (as written
 * by dam_conv2d_tapgrid_f32's bn_partial output) and produces the same outputs / running-stat update. */
int dam_bn_finalize_f32(const float* partial, int parts, int C, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked,
                        float momentum, float eps, float* save_mean, float* save_invstd, float* scale,
                        float* shift, void* stream);

/* dam_bn_finalize_f32 for the two BatchNorms of a pair in one launch (records of equal count and channel number, e.g. from
 * dam_conv_s2_pair_fwd_f32).  a, b: struct dam_bn_fin (`counter` is ignored). */
int dam_bn_finalize_pair_f32(const float* partial_a, const float* partial_b, int parts, int C, const struct dam_bn_fin* a,
                             const struct dam_bn_fin* b, void* stream);

/* First half of dam_bn_stats_f32 on its own: the partial records [*parts_host][C][3] (a HOST int receives the count), sized
 * for dam_bn_finalize_apply_f32 (or dam_bn_finalize_f32).  workspace: dam_bn_workspace_floats(C) floats. */
int dam_bn_stats_partial_f32(const float* x, int64_t n_pixels, int C, float* workspace, int* parts_host, void* stream);

/* dam_bn_finalize_f32 + dam_bn_apply_f32 in ONE launch (relu(bn2(conv2(..)) + shortcut), models/model_resnet.py:26-27; the stem's
 * and ConvBlock2d's relu(bn(conv(x))), models/model_resnet.py:97, models/model_scalar_1s.py:184-186): every workgroup merges the
 * records of the 16 or 32 channels it applies in its prologue -- no finalize launch (4.6-5 us each, whatever it does).
 * fin: the BatchNorm's parameters and outputs as dam_bn_finalize_f32 takes them (all four outputs and the running statistics
 * are written; `counter` is ignored); the remaining arguments as dam_bn_apply_f32.  DAM_BN_FUSED_FIN=0 in the environment
 * makes this (and the backward entry points below) run the separate launches instead (A/B switch). */
int dam_bn_finalize_apply_f32(const float* partial, int parts, int C, const struct dam_bn_fin* fin, const float* x,
                              int64_t n_pixels, const float* res, const float* res_scale, const float* res_shift, int relu,
                              float* y, uint8_t* sign_bits, void* stream);

/* Eval-mode equivalent: the same four outputs from the running statistics. */
int dam_bn_eval_affine_f32(int C, const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, float* save_mean, float* save_invstd,
                           float* scale, float* shift, void* stream);

/* y = relu?( x*scale + shift  [+ res]  or  [+ res*res_scale + res_shift] ).
 * sign_bits (optional, n_pixels * C/4 bytes): one byte per channel quad, bit i = (y[4q + i] > 0) -- the ReLU mask the
 * backward passes need from the block output, in 1/16 of its bytes (mask_bits of dam_bn_backward*_f32). */
int dam_bn_apply_f32(const float* x, int64_t n_pixels, int C, const float* scale, const float* shift,
                     const float* res, const float* res_scale, const float* res_shift, int relu, float* y,
                     uint8_t* sign_bits, void* stream);

/* Backward of y = [relu](bn(x) [+ ...]): dz = dy * mask, where mask is (y_mask > 0) if y_mask (the saved output) is given,
 * the sign bytes dam_bn_apply_f32 wrote if mask_bits is given, (x*mask_scale + mask_shift > 0) if the forward's fused affine
 * is given instead (plain relu(bn(x)): the saved output is then not read at all), 1 if all are NULL (at most one form);
 * dgamma = sum dz*xhat, dbeta = sum dz, dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat))
 * (training) or gamma*invstd*dz (training == 0, running statistics).
 * partials_given > 0: `workspace` already holds that many records [partials_given][C][2] of (sum dz, sum dz*xhat) -- written by
 * a data-gradient launch with bn_bwd -- and the pass over dy and x that would produce them is skipped. */
int dam_bn_backward_f32(const float* dy, const float* y_mask, const float* x, int64_t n_pixels, int C,
                        const float* gamma, const float* save_mean, const float* save_invstd, int training,
                        const float* mask_scale, const float* mask_shift, const uint8_t* mask_bits, float* dx,
                        float* dgamma, float* dbeta, float* workspace, int partials_given, uint32_t* counter, void* stream);

/* The same for TWO BatchNorms that share dy and the mask (exactly one of y_mask / mask_bits) -- a residual block's bn2 and the BatchNorm of its shortcut
 * convolution, both fed by the gradient of relu(bn2(..) + bn_sc(..)) (models/model_resnet.py:23-28): dy and the mask are
 * read once per pass instead of twice, three launches instead of six; bitwise the results of two dam_bn_backward_f32 calls.
 * workspace: dam_bn_pair_workspace_floats(C) floats. */
int64_t dam_bn_pair_workspace_floats(int C);
int dam_bn_backward_pair_f32(const float* dy, const float* y_mask, const uint8_t* mask_bits, int64_t n_pixels, int C,
                             int training,
                             const float* x_a, const float* gamma_a, const float* mean_a, const float* invstd_a,
                             float* dx_a, float* dgamma_a, float* dbeta_a,
                             const float* x_b, const float* gamma_b, const float* mean_b, const float* invstd_b,
                             float* dx_b, float* dgamma_b, float* dbeta_b, float* workspace, void* stream);

/* out[c] = sum_p x[p][c] for c < n_real (gradient of a convolution bias, models/model_scalar_1s.py:167). */
int dam_channel_sum_f32(const float* x, int64_t n_pixels, int C, int n_real, float* out, float* workspace,
                        void* stream);

/* Inverted dropout of ConvBlock2d (models/model_scalar_1s.py:177,187-188; applied only in training mode).  The mask is a
 * counter-based function of (seed, call offset, element index): dam_dropout_tick snapshots and advances the DEVICE
 * call counter by n (graph replays draw fresh masks), dam_dropout_apply_f32 computes y = keep ? x/(1-p) : 0 for the
 * snapshot -- called again on the gradient with the same snapshot in backward.  n % 4 == 0. */
int dam_dropout_tick(int64_t* counter, int64_t n, int64_t* snapshot, void* stream);
int dam_dropout_apply_f32(const float* x, int64_t n, float p, uint64_t seed, const int64_t* snapshot, float* y,
                          void* stream);

/* ---------------------------------------------------------------------------------
 * Gain heads, gain-weighted sum and MSE.  Replace models/model_resnet.py:75-85,108-126 (same code in
 * models/model_scalar_1s.py:222-273, models/model_scalar_2s.py:79-132) and nn.MSELoss at
 * model_trainer.py:21,35.  trunk: NHWC [B][P][C]; conv_w [S][C], conv_b [S], fc_w [S][P], fc_b [S];
 * h [B][S][P] (post-ReLU head activations, kept for backward); gains [B][S]; x [B][S][FT]; masked/gt [B][FT].
 * --------------------------------------------------------------------------------- */
int dam_heads_fwd_f32(const float* trunk, int B, int P, int C, int S, const float* conv_w, const float* conv_b,
                      const float* fc_w, const float* fc_b, float* h, float* gains, void* stream);
int64_t dam_heads_bwd_workspace_floats(int B, int P, int C, int S);
int dam_heads_bwd_f32(const float* dgains, const float* h, const float* trunk, int B, int P, int C, int S,
                      const float* conv_w, const float* fc_w, float* dtrunk, float* dconv_w, float* dconv_b,
                      float* dfc_w, float* dfc_b, float* workspace, void* stream);
int dam_masksum_fwd_f32(const float* x, const float* gains, int B, int S, int64_t FT, float* masked, void* stream);
int64_t dam_masksum_workspace_floats(int B, int S);
int dam_masksum_bwd_f32(const float* dmasked, const float* x, int B, int S, int64_t FT, float* dgains,
                        float* workspace, void* stream);
/* Fused: masked (optional output), loss = mean((masked-gt)^2), dgains = d loss / d gains, one pass over x. */
int dam_masksum_mse_f32(const float* x, const float* gains, const float* gt, int B, int S, int64_t FT,
                        float* masked, float* loss, float* dgains, float* workspace, void* stream);

/* ---------------------------------------------------------------------------------
 * Optimizer.  Replaces optimizer.step() at model_trainer.py:37 for torch.optim.Adam(params,
 * weight_decay=wd) (training.ipynb cell 11; L2 folded into the gradient, not AdamW) over one flat
 * parameter buffer.  step: device int64 counter (incremented here); derived2: device float[2] scratch.
 * grads are multiplied by grad_scale first (1/world_size after a sum all-reduce).
 * hyper_dev (optional): DEVICE float[8] = {lr, beta1, beta2, eps, weight_decay, grad_scale, 1-beta1, 1-beta2} read at
 * run time instead of the scalar arguments, so that a captured graph follows param_groups edits / LR schedulers
 * (1-beta formed in double by the caller, as torch does).
 * All four buffers must be 16-byte aligned.
 * --------------------------------------------------------------------------------- */
int dam_adam_l2_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                         int64_t* step, float* derived2, float lr, float beta1, float beta2, float eps,
                         float weight_decay, float grad_scale, const float* hyper_dev, void* stream);

/* ---------------------------------------------------------------------------------
 * Full-song inference tail (BASELINE config C5), all on the device so that the whole of
 * inference_utils.py:105-145 + the callers' sum / normalise is one hipGraph-capturable sequence.
 * --------------------------------------------------------------------------------- */

/* inference_utils.py:125-130 and :136-141.  raw_db [n_chunks][n_stems] float32 = the model's gain outputs per chunk;
 *   amp[s][c]    = 10 ** (0.5 * raw_db[c][s])                       (data/dataset_utils.py:46-50, float64)
 *   smooth[s][:] = scipy.signal.savgol_filter(amp[s], window, polyorder)   (default mode 'interp': the first / last
 *                  window/2 outputs come from the polynomial fitted to the first / last window)
 * amp, smooth: [n_stems][n_chunks] float64; smooth_f32 (optional): the same values rounded to float32.
 * Argument rules are scipy's (odd window, polyorder < window <= n_chunks -> otherwise DAM_ERR_BAD_ARG);
 * polyorder <= 5 and n_chunks <= 8192 are supported. */
int dam_gains_smooth(const float* raw_db, int n_chunks, int n_stems, int window, int polyorder, double* amp,
                     double* smooth, float* smooth_f32, void* stream);

/* inference_utils.py:12-41 (interpolate_mask) fused with :143 (loaded_tracks[track] * mask):
 *   out[r][n] = audio[r][n] * gains[r / rows_per_gain][min(n / (n_samples / n_gains), n_gains-1)]
 * audio [rows][n_samples] float32 or float64 (audio_is_f64); gains [ceil(rows/rows_per_gain)][n_gains] float64 (one gain
 * sequence per stem = per `rows_per_gain` channel rows); out float32 or float64 (out_is_f64; the reference's numpy
 * product of a float32 track with the float64 mask is float64). */
int dam_gain_ramp_apply(const void* audio, int audio_is_f64, const double* gains, int64_t rows, int64_t rows_per_gain,
                        int64_t n_samples, int n_gains, void* out, int out_is_f64, void* stream);

/* The caller's next step (inference.ipynb cells 9/11, evaluation.py:59-66) fused with the gain ramp:
 *   mix[r][n] = sum_s audio[s][r][n] * gains[s][min(n / (n_samples / n_gains), n_gains-1)]
 * and, if normalize, each row divided by its max-abs (librosa.util.normalize(track_sum, axis=1)).
 * audio [n_stems][rows][n_samples] float32/float64, gains [n_stems][n_gains] float64, mix [rows][n_samples]
 * float32/float64 (mix_is_f64).  workspace: dam_mixdown_workspace_elems(rows) elements of mix's dtype. */
int64_t dam_mixdown_workspace_elems(int64_t rows);
int dam_mixdown_peak_normalize(const void* audio, int audio_is_f64, const double* gains, int n_stems, int64_t rows,
                               int64_t n_samples, int n_gains, int normalize, void* mix, int mix_is_f64,
                               void* workspace, void* stream);

/* ---------------------------------------------------------------------------------
 * ITU-R BS.1770 loudness.  Replaces what the reference gets from the third-party pyloudnorm
 * package (`pyln.Meter(sr).integrated_loudness`): data/dataset.py:115-130, evaluation.py:39-46,59-66,
 * models/baselines/mean_loudness_model.py:10-20.
 *   dam_loudness_kweight_coeffs (host): the two normalised biquads of pyloudnorm's "K-weighting" at `rate`,
 *     coef12 = {b0,b1,b2,1,a1,a2} for the high shelf then the high pass.
 *   dam_loudness_block_energy (device): y = lfilter(high pass, lfilter(high shelf, x)) per channel in float64 and
 *     z[ch][j] = sum_{n in [blk_lo[j], min(blk_hi[j], n_samples))} y[n]^2 / block_len  (the caller supplies the block
 *     bounds exactly as the reference's int() truncations produce them; blk_lo/blk_hi/z are device pointers).
 *     x: float32 or float64 (x_is_f64), sample n of channel ch at x[ch*channel_stride + n*sample_stride].
 *     workspace: dam_loudness_workspace_bytes(n_samples, channels) bytes.
 * The gating of the block energies (absolute -70 LUFS, relative -10 LU) is host logic (loudness.py).
 * --------------------------------------------------------------------------------- */
int dam_loudness_kweight_coeffs(double rate, double* coef12);
int64_t dam_loudness_workspace_bytes(int64_t n_samples, int channels);
int dam_loudness_block_energy(const void* x, int x_is_f64, int64_t n_samples, int channels, int64_t sample_stride,
                              int64_t channel_stride, const double* coef12_host, const int64_t* blk_lo,
                              const int64_t* blk_hi, int n_blocks, double block_len, double* z, void* workspace,
                              void* stream);

/* ---------------------------------------------------------------------------------
 * Stem input layout: x [B][C][HW] (C <= 16 planes, the reference's [B,S,F,T] feature stack) -> y [B][HW][16] with
 * channels C..15 zero, so that the first convolution (models/model_resnet.py:64,97) runs the NHWC kernels.
 * --------------------------------------------------------------------------------- */
int dam_nchw_to_nhwc16_f32(const float* x, int B, int C, int64_t HW, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DAM_HIP_H */
