// Geometry descriptors shared by the convolution kernels (passed by value as kernel arguments).
#pragma once
#include "dam_common.h"

namespace dam {

struct ConvGeo {
    int B, H, W, C;          // input tensor dims; C = channel stride of an NHWC pixel (S planes if in_nchw)
    int Ho, Wo;              // output pixel grid enumerated by this launch
    int N;                   // output channels (multiple of 16) = channel stride of the output tensor
    int OHt, OWt;            // output tensor spatial dims
    int os, oo_h, oo_w;      // output pixel (oh, ow) -> tensor position (oh*os+oo_h, ow*os+oo_w)
    int s;                   // input stride per output pixel (1 or 2)
    int nA, nB;              // tap grid
    int off_h, step_h, off_w, step_w;   // tap (a,b) reads input (oh*s + off_h + a*step_h, ow*s + off_w + b*step_w)
    int wt_base, wt_sa, wt_sb;          // packed weight tap index = wt_base + a*wt_sa + b*wt_sb
    int r0, c0;              // min tap offsets: patch origin (row oh_first*s + r0, col c0)
    int PR, PWin, PWs, PWT;  // patch rows, input columns covered, slots per parity, slots per row (s*PWs)
    int nchunks, CG;         // 16-channel K chunks in total / per LDS group
    int NBtot;               // 16-channel output blocks in the packed weights
    int tiles_m;             // M tiles per image
    int in_nchw;             // 1: input is [B][C][H][W] with C <= 16 planes (first layer)
    int relu_in;             // with in_scale: apply relu(x*scale+shift) while staging
    int relu_out;            // ReLU on the result after bias / residual (eval-mode BatchNorm folded into weights + bias)
    int ksplit;              // split-K over channel groups (tile kernel): raw partial sums go to a workspace slab
    int gps;                 // channel groups per split
    int epi_bwd;             // 1: BatchNorm-backward sums epilogue (BnBwdEpi): `res` is the BatchNorm's input, nothing is added
};

// BatchNorm-backward sums from a data-gradient epilogue: the launch's output IS the gradient dy that reaches relu(bn(x)), so
// the two per-channel sums of the BatchNorm's backward pass  sum(dz), sum(dz * xhat)  with  dz = dy * (x*mscale + mshift > 0),
// xhat = (x - mean) * invstd  are taken from the output registers + one read of x instead of a pass over dy and x.
// Records [workgroup][channel][2] (the layout of dam_bn_backward_f32's workspace).  All pointers: device, per channel.
struct BnBwdEpi {
    const float* x;          // null: off
    const float* mean; const float* invstd; const float* mscale; const float* mshift;
    const unsigned char* res_bits;   // with a residual operand: its ReLU mask as sign bytes (the float mask stays the fallback)
    const unsigned char* mask_bits;  // the BatchNorm's own ReLU mask as sign bytes instead of mscale / mshift (relu(bn(x) + shortcut))
};

struct StripGeo {
    int NR;                  // ring rows (power of two)
    int RH;                  // input rows touched by one output row
    int tiles_m, tpw, strips;  // M tiles per image, tiles per workgroup, workgroups per image
    int ring_off;            // multiple of NR added to row indices before masking
    int w_lds;               // 1: the packed weights of this N tile are resident in LDS behind the ring
    int w_taps;              // taps held in LDS (max used tap index + 1)
    unsigned wo_magic;       // floor(2^32 / Wo) + 1: p / Wo == umulhi(p, wo_magic) for p < 2^22 (scalar-ALU division)
};

// dam_conv_strip.hip
struct BnFinArgs;
int conv_strip_try(ConvGeo& g, int h_lo, int h_hi, const float* X, const float* Wp, const float* bias, float* Y,
                   const float* res, const float* res_mask, float* stats, int* stats_parts, const BnFinArgs* fin,
                   const float* in_scale, const float* in_shift, const BnBwdEpi& bwd, hipStream_t st);

// dam_conv_pipe.hip: the persistent tile kernel with loader waves (thick 3x3 layers); picks its own tile;
// DAM_ERR_UNSUPPORTED = the caller falls back to conv_igemm_kernel
int conv_pipe_try(ConvGeo g, int row_span, const float* X, const float* Wp, const float* bias, const float* sc, const float* sh,
                  float* Y, const float* res, const float* res_mask, float* workspace, float* stats, int* stats_parts,
                  const BnBwdEpi& bwd, hipStream_t st);

}  // namespace dam
