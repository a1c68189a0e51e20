/*
 * asr_hip.h -- C ABI of libasr_hip.so: MI355X (gfx950) kernels for the
 * BLSTM / VGG-BLSTM -> CTC / attention training + decode hot path of
 * hirofumi0810/tensorflow_end2end_speech_recognition.
 *
 * The reference is 100 % Python over TensorFlow 1.x; it has no FFI of its own.
 * The boundary below therefore stands in for the TensorFlow op call sites the
 * hot path dispatches to.  Each entry point cites the
 * reference call site it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns every tensor and every per-call workspace argument (asr_*_workspace_bytes);
 *     the handle owns ONE scratch arena, allocated once by asr_create / asr_create_ex and reported by
 *     asr_scratch_bytes: deterministic split-K slabs and chunk partials of the calls that say so, and
 *     (its top 64 MiB) the exchange slots + error word of the multi-CU recurrence kernels.  A call
 *     that needs more than the arena holds returns ASR_ERR_WORKSPACE -- create the handle larger.
 *     No allocation and no synchronisation inside any call; everything is enqueued on `stream`;
 *   - row-major, contiguous unless a leading dimension (ld*) is given;
 *   - returns 0 on success, a negative asr_status otherwise;
 *     asr_last_error_string() explains the last failure on that handle;
 *   - one handle per GPU per process; calls on one handle are not thread-safe.
 *   - `dtype` selects the MFMA operand type of the matmul-shaped work:
 *     ASR_F32 = exact fp32 (v_mfma_f32_16x16x4_f32), ASR_BF16 = bf16 operands
 *     with fp32 accumulation (v_mfma_f32_16x16x32_bf16).  Gate math, cell
 *     state, CTC recursions, reductions and optimizer state are always fp32.
 */
#ifndef ASR_HIP_H_
#define ASR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct asr_handle asr_handle;
typedef void* asr_stream; /* hipStream_t */

typedef enum {
  ASR_OK = 0,
  ASR_ERR_INVALID_ARG = -1,
  ASR_ERR_UNSUPPORTED = -2,
  ASR_ERR_HIP = -3,
  ASR_ERR_WORKSPACE = -4
} asr_status;

typedef enum { ASR_F32 = 0, ASR_BF16 = 1 } asr_dtype;

/* tf.train.*Optimizer table of models/model_base.py:12-20 */
typedef enum {
  ASR_OPT_SGD = 0, ASR_OPT_MOMENTUM = 1, ASR_OPT_NESTEROV = 2, ASR_OPT_ADAGRAD = 3,
  ASR_OPT_ADADELTA = 4, ASR_OPT_RMSPROP = 5, ASR_OPT_ADAM = 6
} asr_optimizer;

/* ---- lifetime ------------------------------------------------------------ */
int asr_abi_version(void);                                         /* 5: round 6, additive (asr_lstm_bwd_ex).  4: round 5, additive (asr_conv3x3_bwd_weight_bias, asr_conv3x3_smallc_bwd_weight_bias, asr_debug_* hooks).  3: round 4 (2: additions + the two size changes noted at asr_create_ex / asr_ctc_beam_workspace_bytes; 3: asr_att_decoder grew a trailing field) */
int asr_create(asr_handle** out, int device);                        /* 192 MiB scratch arena */
int asr_create_ex(asr_handle** out, int device, size_t scratch_bytes); /* >= 96 MiB (64 MiB of it: recurrence exchange areas) */
size_t asr_scratch_bytes(asr_handle* h);
int asr_destroy(asr_handle* h);
const char* asr_last_error_string(asr_handle* h);
/* Handles used for SIDE-stream work (weight-gradient GEMMs issued beside a recurrence kernel): the lean reduction-major
 * GEMM and its split-K reduction leave the first n XCDs (0 <= n <= 6) to the recurrence clusters, which sit on XCD
 * 0 .. (B/16)*ndir-1 and whose per-step hand-off goes through those XCDs' L2 (workgroup b of a 1-D grid runs on XCD
 * b % 8; workgroups that land on a skipped XCD retire at once).  Results are identical for every n. */
int asr_set_xcd_skip(asr_handle* h, int n);
/* Workgroups (output tiles x split-K slabs) the lean reduction-major GEMM (X^T dG, K = T*B) aims at on this handle;
 * 0 = the default (~2 per CU).  Weight-gradient GEMMs issued beside a recurrence kernel are given few (32: no split-K
 * slabs at all): they have a millisecond to finish and their slab traffic is what slows the recurrence down.  Results
 * differ only in the summation order of the K slabs (fixed for a given n: run-to-run deterministic). */
int asr_set_gemm_tn_workgroups(asr_handle* h, int n);
/* number of CUs / device name (for bench reporting) */
int asr_device_info(asr_handle* h, int* num_cu, char* name, int name_len);

/* ---- layout / elementwise ----------------------------------------------- */
/* [B,T,D] fp32 batch-major -> [T,B,D] time-major in `dtype`
 * (tf.transpose(inputs,[1,0,2]) at models/encoders/core/blstm.py:277-279). */
int asr_bt_to_tb(asr_handle* h, int dtype, const float* in_btd, void* out_tbd,
                 int B, int T, int D, asr_stream s);
/* same with output rows of ld_out >= D elements, columns D..ld_out-1 zero (the first layer's reduction width padded
 * to the GEMM's k tile, e.g. 120 -> 128) */
int asr_bt_to_tb_ld(asr_handle* h, int dtype, const float* in_btd, void* out_tbd,
                    int B, int T, int D, int ld_out, asr_stream s);
/* Batch assembly on the device.  Frame stacking / skipping of a zero-padded batch
 * x[B,T,F] (utils/io/inputs/frame_stacking.py:14-85): out[B, ceil(T/num_skip), F*num_stack], output frame k
 * = input frames k*num_skip .. +num_stack-1 side by side, zero where they run past the utterance;
 * out_len[b] = ceil(seq_len[b]/num_skip) (may be NULL).  num_stack < num_skip is an error as in the
 * reference (:30-31). */
int asr_stack_frames(asr_handle* h, const float* x, const int32_t* seq_len, int B, int T, int F,
                     int num_stack, int num_skip, float* out, int32_t* out_len, asr_stream s);
/* Context splicing per utterance over its own seq_len[b] frames (utils/io/inputs/splicing.py:9-73 as coded:
 * frames t-splice .. t-1 with first/last-frame replication, output laid out [channels][splice*num_stack][3]);
 * x[B,T,D], D = channels*3*num_stack; out[B,T,D*splice], zero for t >= seq_len[b]. */
int asr_splice(asr_handle* h, const float* x, const int32_t* seq_len, int B, int T, int D, int splice,
               int num_stack, float* out, asr_stream s);
/* out[c*ld_out + r] = in[r*ld_in + c] for an [rows, cols] matrix in `dtype` (LDS-tiled, both
 * sides coalesced).  The MFMA GEMM wants both operands reduction-contiguous; the k-major ones
 * (W_x as stored by TF: [Din, 4H], models/encoders/core/blstm.py:286-320 via LSTMBlockCell's
 * kernel) are transposed once per step instead of through bank-conflicting LDS scatter stores. */
int asr_transpose2d(asr_handle* h, int dtype, const void* in, int rows, int cols, int ld_in,
                    void* out, int ld_out, asr_stream s);
/* fp32 -> dtype cast / dtype -> fp32 of n elements (weight copies for the MFMA path) */
int asr_cast_from_f32(asr_handle* h, int dtype, const float* in, void* out, size_t n, asr_stream s);
int asr_cast_to_f32(asr_handle* h, int dtype, const void* in, float* out, size_t n, asr_stream s);
/* out[i] = in[i] * mask[i] (DropoutWrapper(output_keep_prob), blstm.py:308-311;
 * mask already holds 0 or 1/keep_prob).  in/out in `dtype`, mask fp32. */
int asr_apply_mask(asr_handle* h, int dtype, const void* in, const float* mask, void* out,
                   size_t n, asr_stream s);
/* Bernoulli(keep_prob)/keep_prob mask from a counter-based generator (seed, offset) */
/* Read pass over a device buffer (nothing is written): brings it into the memory-side cache ahead of a kernel whose
 * loads are latency-critical (the BPTT kernels' saved activations). */
int asr_touch(asr_handle* h, const void* p, size_t bytes, asr_stream s);
int asr_dropout_mask(asr_handle* h, float* mask, size_t n, float keep_prob,
                     uint64_t seed, uint64_t offset, asr_stream s);
/* The same mask formed where it is used (no mask tensor): out = in * mask(seed, offset) in `dtype`, bit-identical to
 * asr_dropout_mask + asr_apply_mask; and its backward through a ReLU, dpre = (out > 0 ? dout * mask : 0) =
 * asr_relu_bwd with that mask.  tf.nn.dropout on the gigabyte-sized activations of the VGG front-end
 * (vgg_blstm.py:121-157), where an fp32 mask would be three times the activation's own traffic. */
int asr_dropout_apply(asr_handle* h, int dtype, const void* in, void* out, size_t n, float keep_prob, uint64_t seed,
                      uint64_t offset, asr_stream s);
int asr_relu_bwd_drop(asr_handle* h, int dtype, const float* dout, const void* out, size_t n, float keep_prob,
                      uint64_t seed, uint64_t offset, void* dpre, asr_stream s);
/* out[N] = sum over rows of a[M,N] (bias gradients); a in `dtype`, out fp32 */
int asr_colsum(asr_handle* h, int dtype, const void* a, int M, int N, int lda,
               float* out, asr_stream s);

/* ---- GEMM (input projections, output FC, all backward contractions) ------ *
 * C[M,N] = op(A)[M,K] * op(B)[K,N] (+ bias[N]) (+ C if accumulate)
 * transA=0: A is [M,K] with lda; transA=1: A is stored [K,M] with lda.  Same for B.
 * A, B in `dtype`; C in `out_dtype`; bias fp32 or NULL.
 * Replaces fully_connected / the [x,h]W product of LSTMBlockCell hoisted over T
 * (models/ctc/ctc.py:198-233, models/encoders/core/blstm.py:286-320). */
int asr_gemm(asr_handle* h, int dtype, int out_dtype, int transA, int transB,
             int M, int N, int K, const void* A, int lda, const void* B, int ldb,
             void* C, int ldc, const float* bias, int accumulate, asr_stream s);

/* asr_gemm with a fused epilogue activation: act 0 = none, 1 = ReLU (conv_layer /
 * fully_connected(activation_fn=relu): models/encoders/core/cnn_util.py:78-84, vgg_blstm.py:165-173) */
int asr_gemm_act(asr_handle* h, int dtype, int out_dtype, int transA, int transB,
                 int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                 void* C, int ldc, const float* bias, int accumulate, int act, asr_stream s);
/* asr_gemm_act with fp32 output and an elementwise multiplier mul [M, N] (row stride ldmul, fp32) applied last:
 * C = act(op(A) op(B) + bias [+ C]) * mul.  The gradient of a layer input that went through a DropoutWrapper
 * (blstm.py:308-311: dx = dG W_x^T, then times the keep mask of the layer below) in one pass; same bits as
 * asr_gemm_act followed by asr_apply_mask. */
int asr_gemm_mul(asr_handle* h, int dtype, int transA, int transB, int M, int N, int K,
                 const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                 const float* bias, int accumulate, int act, const float* mul, int ldmul, asr_stream s);
/* asr_gemm_mul whose multiplier is the dropout mask asr_dropout_mask(M*N, keep_prob, seed, offset) would produce, formed
 * in the epilogue from its Philox counter instead of read from a mask tensor (C contiguous: ldc == N, N % 4 == 0):
 * a recurrent layer's dx times the DropoutWrapper mask of the layer below (blstm.py:308-311), without that tensor. */
int asr_gemm_drop(asr_handle* h, int dtype, int transA, int transB, int M, int N, int K, const void* A, int lda,
                  const void* B, int ldb, float* C, int ldc, const float* bias, int accumulate, int act, float keep_prob,
                  uint64_t seed, uint64_t offset, asr_stream s);

/* ---- VGG front-end (models/encoders/core/vgg_blstm.py:107-177) --------------- *
 * Images are NHWC: [N = B*T frames, H = channels(40), W = splice*stack, C].
 * 3x3 SAME convolution = asr_im2col3x3 (patches[m, tap*Cin+ci], row stride ldp >= 9*Cin,
 * m = (n*H+h)*W+w) followed by asr_gemm_act(patches, weight[9*Cin, Cout], bias, relu);
 * backward: dW = patches^T dpre, dpatches = dpre W^T, asr_col2im3x3 gathers dpatches back. */
int asr_im2col3x3(asr_handle* h, int dtype, const void* in, int N, int H, int W, int Cin, int ldp,
                  void* patches, asr_stream s);
int asr_col2im3x3(asr_handle* h, const float* dpatches, int N, int H, int W, int Cin, int ldp,
                  float* din, asr_stream s);
/* General SAME convolution (NHWC x HWIO, any kernel / stride) as im2col + asr_gemm: conv_layer of the CLDNN front-end
 * (models/encoders/core/cldnn_wang.py:141-177: 11x21 stride (3,2), 11x11 stride (1,2), 3x3; cnn_util.py:50-84).
 * patches [N*Ho*Wo, ldp] (`dtype`), column (ky*kw + kx)*Cin + ci, Ho = ceil(H/sh), Wo = ceil(W/sw), TensorFlow's SAME
 * padding (the odd cell goes after).  asr_col2im: gradient w.r.t. the input from the gradient of the patches (fp32). */
int asr_im2col(asr_handle* h, int dtype, const void* in_nhwc, int N, int H, int W, int Cin, int kh, int kw,
               int sh, int sw, int ldp, void* patches, asr_stream s);
int asr_col2im(asr_handle* h, const float* dpatches, int N, int H, int W, int Cin, int kh, int kw,
               int sh, int sw, int ldp, float* din_nhwc, asr_stream s);
/* Implicit-GEMM form of the same 3x3 SAME convolution for bf16 operands (no patch matrix: a 64-wide
 * k-tile is one tap x 64 input channels, read straight from the NHWC image; conv_layer of
 * models/encoders/core/cnn_util.py:14-84, VGG blocks of vgg_blstm.py:113-157).
 *   prep:  from the HWIO fp32 master [3,3,Cin,Cout]: wt_fwd[Cout][9*Cin] and the flipped-tap image
 *          wt_bwd[Cin][9*Cout] (both bf16, reduction-contiguous);
 *   fwd:   out[N,H,W,Cout] bf16 = relu?(conv(x) + bias)             (Cin, Cout multiples of 64);
 *   bwd_data:   dx[N,H,W,Cin] fp32 = conv of dy (bf16) with wt_bwd;
 *   bwd_weight: dw[9*Cin, Cout] fp32 (+= if accumulate) = sum over pixels of x(shifted) (x) dy,
 *               deterministic split-K over the pixels through the handle scratch. */
int asr_conv3x3_prep_weights(asr_handle* h, const float* w_hwio, int Cin, int Cout, void* wt_fwd,
                             void* wt_bwd, asr_stream s);
int asr_conv3x3_fwd(asr_handle* h, const void* x, int N, int H, int W, int Cin, const void* wt_fwd,
                    const float* bias, int Cout, int relu, void* out, asr_stream s);
/* conv_layer + tf.nn.dropout (vgg_blstm.py:113-157) in one launch (round 4): out = dropout(round(relu(conv + bias))) with the
 * mask of asr_dropout_apply(keep_prob, seed, offset) formed in the epilogue -- bit for bit asr_dropout_apply of
 * asr_conv3x3_fwd's output; the undropped activation is never written (the backward needs only its sign where the mask
 * kept it: use_drop == 2 below). */
int asr_conv3x3_fwd_drop(asr_handle* h, const void* x, int N, int H, int W, int Cin, const void* wt_fwd,
                         const float* bias, int Cout, float keep_prob, uint64_t seed, uint64_t offset, void* out,
                         asr_stream s);
int asr_conv3x3_bwd_data(asr_handle* h, const void* dy, int N, int H, int W, int Cout,
                         const void* wt_bwd, int Cin, float* dx, asr_stream s);
/* asr_conv3x3_bwd_data followed by asr_relu_bwd(_drop) of the layer below, in the epilogue: dpre_below (bf16
 * [N,H,W,Cin]) = (act_below > 0) ? dx * dropout mask(seed, offset) : 0 -- the fp32 dx is never written.
 * use_drop: 0 no dropout, 1 the mask is formed from (keep_prob, seed, offset), 2 act_below IS the dropped activation
 * (asr_conv3x3_fwd_drop / asr_conv3x3_smallc_fwd_drop): it is > 0 exactly where active and kept, dx is scaled by 1 / keep. */
int asr_conv3x3_bwd_data_relu(asr_handle* h, const void* dy, int N, int H, int W, int Cout, const void* wt_bwd, int Cin,
                              const void* act_below, float keep_prob, uint64_t seed, uint64_t offset, int use_drop,
                              void* dpre_below, asr_stream s);
int asr_conv3x3_bwd_weight(asr_handle* h, const void* x, const void* dy, int N, int H, int W,
                           int Cin, int Cout, float* dw, int accumulate, asr_stream s);
/* the weight AND the bias gradient of a convolution layer (tf.nn.bias_add's gradient, cnn_util.py:49-84: dbias[Cout] =
 * column sums of dy over all N H W pixels) in one call: where the image-resident weight-gradient kernel applies, the bias
 * sums come from the dy images it has staged in LDS anyway (fixed summation order); otherwise this is
 * asr_conv3x3_bwd_weight followed by asr_colsum.  Both outputs are overwritten. */
int asr_conv3x3_bwd_weight_bias(asr_handle* h, const void* x, const void* dy, int N, int H, int W,
                                int Cin, int Cout, float* dw, float* dbias, asr_stream s);
/* tf.nn.max_pool 2x2 stride 2 SAME (cnn_util.py:13-28): out [N, ceil(H/2), ceil(W/2), C];
 * argmax (uint8, 0..3 = position in the window) drives the backward pass. */
int asr_maxpool2x2_fwd(asr_handle* h, int dtype, const void* in, int N, int H, int W, int C,
                       void* out, uint8_t* argmax, asr_stream s);
/* max_pool + tf.nn.dropout in one launch: out = dropout(round(max)) (== asr_dropout_apply of asr_maxpool2x2_fwd's output).
 * C % 4 == 0, 16-byte aligned arrays. */
int asr_maxpool2x2_fwd_drop(asr_handle* h, int dtype, const void* in, int N, int H, int W, int C, void* out,
                            uint8_t* argmax, float keep_prob, uint64_t seed, uint64_t offset, asr_stream s);
int asr_maxpool2x2_bwd(asr_handle* h, const float* dout, const uint8_t* argmax, int N, int H, int W,
                       int C, float* din, asr_stream s);
/* dpre = dout * (out > 0) (* mask if given), written in `dtype` (ReLU + dropout backward) */
/* 3x3 SAME convolution with few input channels (9 Cin <= 32, Cout == 64: the first VGG layer, vgg_blstm.py:113-121),
 * direct -- no patch matrix: out (bf16 [N,H,W,64]) = act(conv(x bf16 [N,H,W,Cin], w2d bf16 [9 Cin, 64]) + bias), and
 * its weight gradient dw fp32 [9 Cin, 64] = patches(x)^T dpre (dpre bf16 [N,H,W,64]; deterministic two-stage sum). */
int asr_conv3x3_smallc_fwd(asr_handle* h, const void* x, int N, int H, int W, int Cin, const void* w2d,
                           const float* bias, int Cout, int relu, void* out, asr_stream s);
int asr_conv3x3_smallc_fwd_drop(asr_handle* h, const void* x, int N, int H, int W, int Cin, const void* w2d,
                                const float* bias, int Cout, float keep_prob, uint64_t seed, uint64_t offset, void* out,
                                asr_stream s);      /* ReLU + dropout in the epilogue, as asr_conv3x3_fwd_drop */
int asr_conv3x3_smallc_bwd_weight(asr_handle* h, const void* x, const void* dpre, int N, int H, int W, int Cin,
                                  int Cout, float* dw, asr_stream s);
/* + the bias gradient dbias[64] = column sums of dpre, from a column of ones in the patch operand (as
 * asr_conv3x3_bwd_weight_bias) */
int asr_conv3x3_smallc_bwd_weight_bias(asr_handle* h, const void* x, const void* dpre, int N, int H, int W, int Cin,
                                       int Cout, float* dw, float* dbias, asr_stream s);
/* asr_dropout_apply (on the pooled gradient, when use_drop) -> asr_maxpool2x2_bwd -> asr_relu_bwd of the convolution
 * under the pool, as one pass without the full-resolution fp32 gradient in between: dpre[n,h,w,c] (operand dtype) =
 * (act[n,h,w,c] > 0 && argmax[o] == 2 (h & 1) + (w & 1)) ? dout[o] * mask(o) : 0, o = pooled cell (n, h/2, w/2, c).
 * C % 4 == 0, 16-byte aligned arrays.
 * use_drop == 2: `act` is the POOLED activation after its dropout ([N, ceil(H/2), ceil(W/2), C], asr_maxpool2x2_fwd_drop, or
 * the plain pooled output with keep_prob 1): dpre = (act[o] > 0 && argmax[o] == position) ? dout[o] / keep_prob : 0 -- the
 * full-resolution ReLU output is not read. */
int asr_maxpool2x2_relu_bwd(asr_handle* h, int dtype, const float* dout, const uint8_t* argmax, const void* act, int N,
                            int H, int W, int C, void* dpre, float keep_prob, uint64_t seed, uint64_t offset,
                            int use_drop, asr_stream s);
int asr_relu_bwd(asr_handle* h, int dtype, const float* dout, const void* out, const float* mask,
                 size_t n, void* dpre, asr_stream s);
/* dpre = (out > 0) ? dout * (1 / keep) : 0 with `out` a DROPPED ReLU output (asr_conv3x3_fwd_drop etc.): the same values as
   asr_relu_bwd_drop over the undropped output, without forming the mask */
int asr_relu_bwd_scaled(asr_handle* h, int dtype, const float* dout, const void* out, size_t n, float keep, void* dpre,
                        asr_stream s);

/* ---- LSTM recurrence ------------------------------------------------------ *
 * One layer, `ndir` directions (1 = LSTMEncoder, 2 = BLSTMEncoder), all T steps:
 * tf.contrib.rnn.LSTMBlockCell(forget_bias, clip_cell, use_peephole) under
 * tf.nn.(bidirectional_)dynamic_rnn(sequence_length) --
 * models/encoders/core/blstm.py:286-323, lstm.py:253-285.
 *
 * Gate layout.  The TF variable `kernel` [Din+H, 4H] has gate-major columns q*H + j
 * (q = i, ci, f, o).  The device-side activation tensors keep the four gates of a unit
 * INTERLEAVED (column j*4 + q) so that the recurrence kernels move one 16-byte (fp32) or
 * 8-byte (bf16) quadruple per (utterance, unit) pair:
 *     xproj / gates / dgates : [T, B, ndir, H, 4]
 *
 * asr_lstm_prep_weights: once per (layer, direction) per step, from kernel/bias (fp32):
 *   wx_il   [Din, 4H] `dtype`  W_x with interleaved columns (B operand of the hoisted GEMM)
 *   bias_il [4H] fp32          bias, interleaved
 *   packed_fwd / packed_bwd    W_h = kernel[Din:] in MFMA B-fragment order for
 *                              h[16,H] x W_h  and  dG[16,4H] x W_h^T   (H*4H `dtype` each) */
int asr_lstm_prep_weights(asr_handle* h, int dtype, const float* kernel, const float* bias,
                          int Din, int H, void* wx_il, float* bias_il,
                          void* packed_fwd, void* packed_bwd, asr_stream s);
/* asr_lstm_prep_layer: the same images for ALL directions of a layer in one launch, in the form the layer's GEMMs
 * consume (no transposes / concatenations afterwards).  vars[ndir*5] (host array of device pointers), per direction:
 * {kernel [Din+H,4H], bias [4H], w_i_diag, w_f_diag, w_o_diag [H]} -- the five tf.get_variable()s of one
 * LSTMBlockCell (blstm.py:287-305; SURVEY App. C names); the peephole pointers may be NULL iff peep_out is NULL.
 *   wxT     [ndir*4H, ldk] `dtype`  W_x^T, interleaved rows, directions stacked, columns Din..ldk-1 zero
 *                                   (xproj = x [T*B, ldk] . wxT^T: one GEMM for both directions)
 *   wx_cat  [Din, ndir*4H] `dtype`  W_x, interleaved columns, directions side by side (dx = dG . wx_cat^T)
 *   bias_cat[ndir*4H] fp32, packed_fwd / packed_bwd [ndir][H*4H] `dtype`, peep_out [ndir][3][H] fp32 */
int asr_lstm_prep_layer(asr_handle* h, int dtype, int ndir, const float* const* vars, int Din, int ldk, int H,
                        void* wxT, void* wx_cat, float* bias_cat, void* packed_fwd, void* packed_bwd,
                        float* peep_out, asr_stream s);
/* asr_lstm_grad_finish: the way back for one layer's gradients, one launch.  dw_il [ndir][rows = Din+H][4H] fp32 with
 * interleaved columns (output of the weight-gradient GEMMs) and dpeep_dbias [ndir][7][H] (asr_lstm_bwd) are written
 * to grads[ndir*5] = the gradient buffers of the same five variables, in TF's layouts (kernel gate-major). */
int asr_lstm_grad_finish(asr_handle* h, int ndir, float* const* grads, int rows, int H, const float* dw_il,
                         const float* dpeep_dbias, int has_peep, asr_stream s);
/* [rows, 4H] fp32 with interleaved columns -> gate-major columns (dW after the GEMMs) */
int asr_gate_deinterleave(asr_handle* h, const float* in, int ld_in, float* out, int ld_out,
                          int rows, int H, asr_stream s);

/* Forward.  xproj[T,B,ndir,H,4] fp32 = x*wx_il + bias_il (from asr_gemm).
 * wh_packed: ndir fwd-packed buffers back to back.  peep: [ndir][3][H] fp32
 * (w_i_diag, w_f_diag, w_o_diag) or NULL (use_peephole=False).
 * gates[T,B,ndir,H,4] `dtype`: post-activation i, ci, f, o (valid frames only);
 * hout[T,B,ndir*H] `dtype`: cell outputs, zero for t >= seq_len[b];
 * cs[T,B,ndir*H] fp32: cell state per frame (after clipping);
 * c_final/h_final [ndir][B][H] fp32 (may be NULL): state at the last valid step.
 * cell_clip <= 0 disables clipping.  B must be a multiple of 16 (pad with
 * seq_len 0 rows); H in {64,128,192,256,320,512}. */
int asr_lstm_fwd(asr_handle* h, int dtype, int T, int B, int H, int ndir,
                 const float* xproj, const void* wh_packed, const float* peep,
                 const int32_t* seq_len, float forget_bias, float cell_clip,
                 void* gates, void* hout, float* cs, float* c_final, float* h_final,
                 asr_stream s);

/* Backward through time.  dhout[T,B,ndir*H] fp32: gradient w.r.t. hout (already
 * multiplied by the dropout mask if any).  d_c_final/d_h_final [ndir][B][H] or NULL.
 * gates/cs from forward.  wh_packed_bwd: ndir bwd-packed buffers.
 * dgates[T,B,ndir,H,4] in `dtype`: gradient w.r.t. the pre-activations (interleaved, zero
 * at padded frames) -- feeds the dW_x, dW_h, dx GEMMs.
 * dpeep_dbias [ndir][7][H] fp32 (may be NULL; overwritten): rows 0-2 = gradients of the
 * peepholes (w_i_diag, w_f_diag, w_o_diag), rows 3-6 = gradient of the bias, i.e. the column
 * sums of dgates per gate block (i, ci, f, o) -- accumulated in registers during BPTT.
 * dpeep_workspace: (B/16)*ndir*7*H floats (per batch-tile partials, reduced in a fixed
 * order so the result is run-to-run deterministic); required iff dpeep_dbias != NULL. */
int asr_lstm_bwd(asr_handle* h, int dtype, int T, int B, int H, int ndir,
                 const float* dhout, const void* gates, const float* cs,
                 const void* wh_packed_bwd, const float* peep, const int32_t* seq_len,
                 const float* d_c_final, const float* d_h_final,
                 void* dgates, float* dpeep_dbias, float* dpeep_workspace, asr_stream s);
/* The same with clip_no_grad: 0 = asr_lstm_bwd (LSTMBlockCell: the fused gradient op does not know cell_clip, the clip is
 * straight-through).  > 0 = the forward ran with cell_clip = clip_no_grad and the cell is tf.contrib.rnn.LSTMCell
 * (models/encoders/core/blstm.py:187-230 of the reference, the num_proj cell), whose tf.clip_by_value passes NO gradient
 * through a clamped state: a frame whose saved cs has |c| >= clip_no_grad sends nothing to its gates or to c_prev
 * (its output gate still receives its gradient).  Either operand dtype. */
int asr_lstm_bwd_ex(asr_handle* h, int dtype, int T, int B, int H, int ndir,
                    const float* dhout, const void* gates, const float* cs,
                    const void* wh_packed_bwd, const float* peep, const int32_t* seq_len,
                    const float* d_c_final, const float* d_h_final, float clip_no_grad,
                    void* dgates, float* dpeep_dbias, float* dpeep_workspace, asr_stream s);

/* The recurrences run as clusters of H/32 (or H/64) workgroups per (direction, 16-utterance tile) that hand their
 * slices of h / dh to each other inside the launch (bounded spins).  This synchronises the device and reports the
 * sticky error word (ASR_ERR_HIP when non-zero) -- call at a sync point.  Bits: 1 = a forward hand-off timed out,
 * 2 = a backward hand-off timed out (never expected), 4 = a bf16 forward recurrence published a NON-FINITE hidden state
 * (the model diverged: its 4-byte self-tagged exchange words would otherwise hand the peers a finite value).
 * Saved activations (gates, cs) of frames past an utterance's length are UNSPECIFIED (rows past their length park
 * their accesses): only asr_lstm_bwd consumes them and it never reads those positions; hout and dgates ARE
 * zero there. */
int asr_check_async_errors(asr_handle* h, unsigned* flags_out);
/* Non-blocking form for training loops: enqueues ONE 4-byte device->host copy of the sticky error word on `s` into
 * host_flags (pinned host memory owned by the caller; valid once work on `s` up to here has completed) -- the Python
 * front end arms it after every optimizer step and inspects the previous step's word (ops.ErrorWatch), so a timed-out
 * hand-off raises within one step instead of silently training on garbage. */
int asr_peek_async_errors(asr_handle* h, unsigned* host_flags, asr_stream s);
/* Resets the sticky error word (after it has been reported). */
int asr_clear_async_errors(asr_handle* h, asr_stream s);
/* Debug / test switches of the cluster kernels (process-wide, same bits as the environment variable ASR_LSTM_DFLAGS):
 * 16 = force the placement-independent write-through exchange, 64 = TEST ONLY, make every hand-off time out;
 * 32 / 128 / 256 = A-B switches of kernel variants (own-slice products ahead of the poll, the BPTT kernel's fetch ahead of
 * its poll loop, MFMA priority in the fp32 BPTT kernel), result-neutral; 512 = clusters of H/64 CUs x eight waves instead
 * of H/32 CUs x four (H = 256 / 512). */
int asr_debug_set_lstm_flags(int flags);
/* 1 (default; env ASR_GRU_PERSISTENT): asr_gru_fwd / asr_gru_bwd run ONE launch per call -- for 64 / 128 / 256 / 320 units on
 * clusters of H/32 CUs (round 6: the recurrent blocks as three-bf16-term fragments in registers, two all-gathers per step;
 * env ASR_GRU_CLUSTER=0 keeps the single-CU form), else on one CU per (direction, tile) with the state in LDS and both
 * products on exact-fp32 MFMA where the LDS images fit; 0: the launch-per-step kernels. */
int asr_debug_set_gru_persistent(int on);

/* Debug / measurement: are 16-byte per-lane stores seen whole by 16-byte loads of another CU?  Workgroup 0 stores
 * {i, i, i, i}, i = 1 .. iters, into buf[16 lane .. ] (64 lanes; plain stores, or write-through when write_through != 0);
 * workgroup `peer` (8: the same XCD as workgroup 0, 1: another one) polls them with L1-bypassing loads.
 * out[3 lane .. +2] = {loads, loads whose four dwords differed, distinct values seen}.  Evidence for DESIGN section 8
 * item 1-i (the product path exchanges 8-byte granules and per-word tags only). */
int asr_debug_tear_probe(asr_handle* h, unsigned* buf, unsigned iters, int peer, int write_through,
                         unsigned long long* out, asr_stream s);

/* Debug: records, per workgroup of a probe grid launched on `s`, {XCC id, HW_ID register} into out[2*nblocks]
 * (device memory); every workgroup stays resident for spin_cycles so that the grid spreads over the CUs. */
int asr_debug_placement(asr_handle* h, unsigned* out, int nblocks, int spin_cycles, asr_stream s);
/* test aid: fills every CU's LDS with NaN bit patterns (a kernel that reads LDS words nobody wrote then fails every time
 * instead of once in a few cold starts); tests/conftest.py runs it in front of every test under ASR_POISON_LDS=1 */
int asr_debug_poison_lds(asr_handle* h, asr_stream s);

/* ---- GRU recurrence ------------------------------------------------------- *
 * tf.contrib.rnn.GRUCell under tf.nn.(bidirectional_)dynamic_rnn(sequence_length) -- the reference's GRUEncoder /
 * BGRUEncoder (models/encoders/core/gru.py:58-76, :126-152).  fp32.  One layer, ndir directions, all steps:
 *     [r, u] = sigmoid(xg_t + h W_gh),  c = tanh(xc_t + (r * h) W_ch),  h' = u * h + (1 - u) * c
 * xg [T,B,ndir,2H] = x W_g[:D] + b_g and xc [T,B,ndir,H] = x W_c[:D] + b_c are the caller's hoisted GEMMs;
 * wgh [ndir][H][2H] / wch [ndir][H][H] the recurrent blocks of the two kernels (row-major), wghT / wchT their
 * transposes ([ndir][2H][H] / [ndir][H][H]).  tmax = max(seq_len) (host value: the step loop is issued by the host).
 * Saved for the backward pass, all [T,B,ndir,H] at the frame a row worked on: r, u, c, rh = r * h_prev.
 * hout [T,B,ndir*H] (zero past seq_len); hstate2: 2*ndir*B*H floats of work space, the final state [ndir,B,H] is
 * left in its first half.  r, u, c past a row's length are unspecified, rh there is left as the caller set it (zeros:
 * the candidate kernel's weight gradient contracts it over all T*B rows); a timed-out cluster hand-off sets the sticky
 * error word of asr_check_async_errors (bit 1 forward, bit 2 backward) as the LSTM clusters do.
 * asr_gru_bwd: dout [T,B,ndir*H] (+ d_h_final [ndir,B,H] or NULL) -> dgate [T,B,ndir,2H] (d r_pre | d u_pre) and
 * dcand [T,B,ndir,H] (d c_pre), zero where no row is active; the weight / input gradients are GEMMs over them
 * (dW_g = [x; h_prev]^T dgate, dW_c = [x; rh]^T dcand, dx = dgate W_gx^T + dcand W_cx^T).  work2: 2*ndir*B*H floats. */
int asr_gru_fwd(asr_handle* h, int T, int B, int H, int ndir, const float* xg, const float* xc,
                const float* wgh, const float* wch, const int32_t* seq_len, int tmax, float* r, float* u,
                float* c, float* rh, float* hout, float* hstate2, asr_stream s);
int asr_gru_bwd(asr_handle* h, int T, int B, int H, int ndir, const float* dout, const float* d_h_final,
                const float* hout, const float* r, const float* u, const float* c, const float* wghT,
                const float* wchT, const int32_t* seq_len, int tmax, float* dgate, float* dcand,
                float* work2, asr_stream s);

/* ---- CTC ------------------------------------------------------------------ *
 * tf.nn.ctc_loss(labels, logits, seq_len, preprocess_collapse_repeated=False,
 * ctc_merge_repeated=True, time_major=True) -- models/ctc/ctc.py:289-297,
 * models/attention/joint_ctc_attention.py:308-316.  blank = C-1.
 * logits[T,B,C] fp32; labels_flat: concatenated label ids (int32), label_offsets[B+1];
 * loss[B] fp32; grad[T,B,C] fp32 = d loss[b] / d logits * grad_scale (pass 1/B to get the
 * gradient of the batch mean of ctc.py:298); zero for t >= seq_len[b].
 * Infeasible utterances give loss 0 and grad 0 (ignore_longer_outputs_than_inputs=True);
 * their count is written to *num_infeasible (device int32, may be NULL).
 * workspace: asr_ctc_workspace_bytes(T,B,max label length). */
size_t asr_ctc_workspace_bytes(int T, int B, int max_label_len);
int asr_ctc_loss(asr_handle* h, const float* logits, int T, int B, int C,
                 const int32_t* labels_flat, const int32_t* label_offsets,
                 const int32_t* seq_len, int max_label_len, float grad_scale,
                 float* loss, float* grad, int32_t* num_infeasible,
                 void* workspace, size_t workspace_bytes, asr_stream s);

/* tf.nn.ctc_greedy_decoder(merge_repeated=True) -- models/ctc/ctc.py:341-342 and
 * models/ctc/decoders/greedy_decoder.py:19-50.  logits[T,B,C]; out_labels[B,T] int32
 * padded with -1; out_len[B]. */
int asr_ctc_greedy_decode(asr_handle* h, const float* logits, int T, int B, int C,
                          const int32_t* seq_len, int blank,
                          int32_t* out_labels, int32_t* out_len, asr_stream s);

/* Prefix beam search -- models/ctc/decoders/beam_search_decoder.py:53-152
 * (also stands in for tf.nn.ctc_beam_search_decoder, models/ctc/ctc.py:344-346).
 * logits[T,B,C] fp32 (log-softmax is taken inside); out_labels[B,T] padded -1;
 * out_len[B]; out_score[B] = -log p of the best prefix (float64).
 * beam_width <= 128.  Scores are fp64, ties are broken like the reference's stable sort. */
size_t asr_ctc_beam_workspace_bytes(int T, int B, int C, int beam_width);
int asr_ctc_beam_decode(asr_handle* h, const float* logits, int T, int B, int C,
                        const int32_t* seq_len, int blank, int beam_width,
                        int32_t* out_labels, int32_t* out_len, double* out_score,
                        void* workspace, size_t workspace_bytes, asr_stream s);

/* row softmax: CTC.posteriors (models/ctc/ctc.py:354-380) */
int asr_softmax_rows(asr_handle* h, const float* in, float* out, int rows, int C, asr_stream s);

/* ---- attention decoder (models/attention/...) -------------------------------- *
 * Encoder outputs and keys are TIME-MAJOR: enc[T,B,E], keys[T,B,A]; per-utterance vectors [B,*].
 *
 * One LSTMBlockCell step of the decoder RNN (attention_seq2seq.py:352-371): pre[B,4U] fp32 =
 * [x,h]W + b with TF gate-major columns (i, ci, f, o); peep [3][U] or NULL.  live[B] (1/0):
 * rows already finished keep their state (dynamic_decode impute_finished,
 * decoders/dynamic_decoder.py:171-193).  Outputs: gates[B,4U] post-activation, c_raw/h_raw = new
 * cell/output before the finished-carry, c_out/h_out = carried state. */
int asr_lstm_cell_fwd(asr_handle* h, const float* pre, const float* c_prev, const float* h_prev,
                      const float* peep, const float* live, int B, int U, float forget_bias,
                      float cell_clip, float* gates, float* c_raw, float* c_out, float* h_out,
                      float* h_raw, asr_stream s);
/* The same step with its results also written where the NEXT kernels of the decoder step read them, so that no copy
 * launch sits between them (attention_decoder.py:142-229: the cell output feeds the attention layer and the
 * attentional vector, the carried h the next step's cell input):
 *   cell_out [B,U] (may be NULL) = h_raw * out_mask (out_mask [B,U] fp32 or NULL: DropoutWrapper(output_keep_prob));
 *   cell_out2 (may be NULL): the same values at cell_out2[b*ld_c2 + j]  (a column block of a wider row-major array);
 *   h_out2   (may be NULL): h_out at h_out2[b*ld_h2 + j]. */
int asr_lstm_cell_fwd_ex(asr_handle* h, const float* pre, const float* c_prev, const float* h_prev,
                         const float* peep, const float* live, int B, int U, float forget_bias,
                         float cell_clip, float* gates, float* c_raw, float* c_out, float* h_out,
                         float* h_raw, const float* out_mask, float* cell_out, float* h_out2, int ld_h2,
                         float* cell_out2, int ld_c2, asr_stream s);
/* asr_gemm_act(x [B,K] (row stride ldx) x W [K,4U] + b) -> asr_lstm_cell_fwd_ex as ONE launch (a decoder step's first two
 * kernels): the product runs on the gate-INTERLEAVED weight image W_il [(K + 1), 4U] (column 4 u + g = gate g of unit u, the
 * last row the bias; written by asr_lstm_cell_gemm_prep once per weight update), so a workgroup holds all four gates of its
 * units and applies the cell in its epilogue; the pre-activations are never written.  Same arithmetic in the same order
 * as the two separate calls: bit-identical outputs.  asr_lstm_cell_gemm_ok: B <= 32, K % 64 == 0, U % 8 == 0, ldx % 4 == 0. */
int asr_lstm_cell_gemm_ok(int B, int K, int U, int ldx);
int asr_lstm_cell_gemm_prep(asr_handle* h, const float* W, const float* bias, int K, int U, float* W_il, asr_stream s);
int asr_lstm_cell_gemm_fwd(asr_handle* h, const float* x, int ldx, int K, const float* W_il, int has_bias,
                           const float* c_prev, const float* h_prev, const float* peep, const float* live,
                           int B, int U, float forget_bias, float cell_clip, float* gates, float* c_raw,
                           float* c_out, float* h_out, float* h_raw, const float* out_mask, float* cell_out,
                           float* h_out2, int ld_h2, float* cell_out2, int ld_c2, asr_stream s);
/* The same pair with bf16 WEIGHTS (fp32 activations, exact fp32 products of an fp32 value and a bf16-valued one): `img`
 * (asr_lstm_cell_gemm_h_bytes) holds, written by asr_lstm_cell_gemm_prep_h, the gate-interleaved kernel in the order the
 * multiplies consume it (four 16-byte loads per lane and 64-row unit), the interleaved fp32 bias, and a plain bf16 copy of
 * W whose rows asr_lstm_cell_gemm_bwd_h (dx [B,K] = dpre [B,4U] W^T, the backward product of a decoder step) streams. */
size_t asr_lstm_cell_gemm_h_bytes(int K, int U);
int asr_lstm_cell_gemm_prep_h(asr_handle* h, const float* W, const float* bias, int K, int U, void* img, asr_stream s);
int asr_lstm_cell_gemm_fwd_h(asr_handle* h, const float* x, int ldx, int K, const void* img,
                             const float* c_prev, const float* h_prev, const float* peep, const float* live,
                             int B, int U, float forget_bias, float cell_clip, float* gates, float* c_raw,
                             float* c_out, float* h_out, float* h_raw, const float* out_mask, float* cell_out,
                             float* h_out2, int ld_h2, float* cell_out2, int ld_c2, asr_stream s);
int asr_lstm_cell_gemm_bwd_h(asr_handle* h, const float* dpre, int B, int K, int U, const void* img, float* dx, int lddx,
                             asr_stream s);
/* dh_use: gradient on the cell output of live rows; dc_next/dh_next: gradient on the carried state.
 * dpre[B,4U]; dc_prev; dh_prev_carry (pass-through part, add (dpre W^T)[h] for live rows);
 * dpeep_rows[B][3][U] per-row peephole gradient terms (sum over rows/steps on the caller side). */
int asr_lstm_cell_bwd(asr_handle* h, const float* dh_use, const float* dc_next, const float* dh_next,
                      const float* gates, const float* c_raw, const float* c_prev, const float* peep,
                      const float* live, int B, int U, float* dpre, float* dc_prev,
                      float* dh_prev_carry, float* dpeep_rows, asr_stream s);
/* The same with the forward's cell clip: a state the forward clamped (tf.clip_by_value inside LSTMCell,
 * models/recurrent/layers/lstm.py:152-157) passes no gradient to the gates or to c_prev; c_raw holds the clamped value,
 * |c_raw| >= cell_clip marks it.  cell_clip <= 0: identical to asr_lstm_cell_bwd.  (asr_att_decoder_bwd does NOT do
 * this: the attention decoder's cell is tf.contrib.rnn.LSTMBlockCell, attention_seq2seq.py:354-363, whose gradient op has
 * no clip attribute -- the clamp is transparent to the gradient there, as in the encoders' fused cells.) */
int asr_lstm_cell_bwd_ex(asr_handle* h, const float* dh_use, const float* dc_next, const float* dh_next,
                         const float* gates, const float* c_raw, const float* c_prev, const float* peep,
                         const float* live, int B, int U, float cell_clip, float* dpre, float* dc_prev,
                         float* dh_prev_carry, float* dpeep_rows, asr_stream s);
/* Attention energies (attention_layer.py:115-347).  mode 0 (bahdanau_content / location / hybrid):
 * energy[b,t] = sum_a v[a] * tanh(keys[t,b,a] + qz[b,a])   (keys NULL for 'location');
 * mode 1 (dot_product / luong_dot / luong_general): energy[b,t] = sum_a keys[t,b,a] * qz[b,a].
 * qz = W_query s (+ W_filter bias where the type has location features). */
int asr_att_energy_fwd(asr_handle* h, const float* keys, const float* qz, const float* v, int T,
                       int B, int A, int mode, float* energy, asr_stream s);
/* dkeys[T,B,A] += (may be NULL); dqz[B,A] = ; dv_rows[B,A] = per-utterance v gradient (may be NULL) */
int asr_att_energy_bwd(asr_handle* h, const float* denergy, const float* keys, const float* qz,
                       const float* v, int T, int B, int A, int mode, float* dkeys, float* dqz,
                       float* dv_rows, asr_stream s);
/* energy*mask + (1-mask)*float32.min, * sharpening, softmax over t, context = sum_t alpha enc
 * (attention_layer.py:75-111).  alpha[B,T], ctx[B,E].  enc[T,B,E] in `enc_dtype`: a bf16-operand model passes
 * the bf16 copy of the encoder output it already keeps, halving the two per-step streams over enc.
 * sigmoid_norm NULL: softmax.  Non-NULL ([B], written): `sigmoid_smoothing` (attention_layer.py:92-96),
 * alpha = sigmoid(e) / sum_t sigmoid(e), and sigmoid_norm[b] = that sum -- hand it back to the backward. */
int asr_att_softmax_ctx_fwd(asr_handle* h, const float* energy, const int32_t* seq_len,
                            float sharpening, const void* enc, int enc_dtype, int T, int B, int E,
                            float* alpha, float* ctx, float* sigmoid_norm, asr_stream s);
/* the same with the context also written to up to two column blocks of wider row-major arrays
 * (ctx2[b*ld2 + e], ctx3[b*ld3 + e]; NULL = not wanted): the next step's cell input and the attentional vector's */
int asr_att_softmax_ctx_fwd_ex(asr_handle* h, const float* energy, const int32_t* seq_len,
                               float sharpening, const void* enc, int enc_dtype, int T, int B, int E,
                               float* alpha, float* ctx, float* sigmoid_norm, float* ctx2, int ld2,
                               float* ctx3, int ld3, asr_stream s);
/* denergy[B,T] = ; denc[T,B,E] += alpha * dctx.  denc may be NULL: a decoder loop then keeps alpha and
 * dctx of every step and forms d_enc = sum_steps alpha (x) dctx with ONE GEMM per utterance at the end
 * instead of a read-modify-write of the whole [T,B,E] tensor per step. */
int asr_att_softmax_ctx_bwd(asr_handle* h, const float* dctx, const float* alpha,
                            const int32_t* seq_len, float sharpening, const void* enc, int enc_dtype,
                            int T, int B, int E, float* denergy, float* denc,
                            const float* sigmoid_norm, const float* dalpha_extra, asr_stream s);
/* dalpha_extra [B,T] (may be NULL): a further gradient w.r.t. alpha, added before the softmax gradient -- with
 * carried location features the NEXT decoder step's conv1d hands one back (asr_att_loc_energy_bwd).
 *
 * Location / hybrid attention with the previous step's weights carried (attention_layer.py:191-265; the recurrence
 * the reference wrote down -- its graph feeds zeros at every step, SURVEY Appendix A Q1, which stays the default of
 * the Python model: prev_alpha='zeros' | 'carry'):
 *   f = tf.nn.conv1d(alpha_prev[B,T,1], filter[taps,1,10], stride 1, 'SAME')    taps = 201 (location) / 200 (hybrid),
 *       zero padding (taps-1)/2 frames before, the rest after; cross-correlation
 *   energy[b,t] = sum_a v[a] tanh( keys[t,b,a] + qz[b,a] + (f W_filter)[b,t,a] )     keys NULL for 'location';
 *       qz = W_query s + b_filter as for asr_att_energy_fwd.
 * filt [taps,10] (the [taps,1,10] variable), wfil [10,A] (W_filter/weights). */
int asr_att_loc_energy_fwd(asr_handle* h, const float* alpha_prev, const float* filt, const float* wfil,
                           const float* keys, const float* qz, const float* v, int T, int B, int A, int taps,
                           float* energy, asr_stream s);
/* Gradients of the above for one decoder step.  dkeys[T,B,A] += (may be NULL); dqz[B,A], dv_rows[B,A] written;
 * dwfil_rows[B,10,A] and dfilt_rows[B,taps,10]: per-utterance gradients of W_filter/weights and filter, overwritten
 * when accumulate == 0 and added to otherwise (a decoder loop accumulates over its steps and sums over B once);
 * dalpha_prev[B,T] written: gradient w.r.t. the previous step's weights (-> dalpha_extra of that step).
 * Partials go through the handle scratch in a fixed order: run-to-run deterministic. */
int asr_att_loc_energy_bwd(asr_handle* h, const float* denergy, const float* alpha_prev, const float* filt,
                           const float* wfil, const float* keys, const float* qz, const float* v, int T, int B,
                           int A, int taps, float* dkeys, float* dqz, float* dv_rows, float* dwfil_rows,
                           float* dfilt_rows, float* dalpha_prev, int accumulate, asr_stream s);
/* ---- the decoder loop, native ----------------------------------------------------------------------------- *
 * dynamic_decode over a TrainingHelper (decoders/dynamic_decoder.py:68-218, attention_decoder.py:142-229): all To steps
 * of the attention decoder issued from ONE call -- the per-step sequence cell-input GEMM -> asr_lstm_cell_fwd_ex ->
 * query GEMM -> asr_att_(loc_)energy_fwd -> asr_att_softmax_ctx_fwd_ex, i.e. exactly the entry points above in the
 * order a host loop would call them (a Python host spends ~15 us per call, ~9 calls per step, 400 steps: the loop was
 * host-bound), and the reverse sequence for the gradients (where adjacent steps of that sequence have a fused kernel --
 * the dctx add inside the d-alpha kernel, the softmax backward inside the energy backward, mask and carried-dh add
 * inside the cell backward -- the loop uses it; the arithmetic is that of the separate entry points).  Every array is the caller's; per-step arrays are [To, ...]
 * row blocks.  Layouts: dec_in [To,B,Em+E2+U] = embedded input | previous context | previous h (the embedding columns
 * and row 0 filled by the caller, the rest by the loop), av_in [To,B,U+E2] = cell output (after its dropout mask) |
 * context; c_all / h_all [To+1,B,U]: carried state BEFORE step k at row k (row 0 = initial state from the bridge). */
typedef struct asr_att_decoder {
  int To, B, T, U, Em, E2, A;       /* A: width of keys / qz */
  int att_mode;                     /* 0 additive (v tanh(keys + qz)), 1 dot */
  int has_query_fc;                 /* qz = cell_out W_q (+ b_q); 0: qz = cell_out (then A == U) */
  int carry_alpha, taps;            /* location features of the previous step's weights (asr_att_loc_energy_*) */
  int enc_dtype;
  float forget_bias, cell_clip, sharpening;
  const float *W_cell, *b_cell, *peep;          /* [Em+E2+U,4U], [4U], [3,U] or NULL */
  const float *W_q, *b_q, *v;                   /* [U,A] (row stride ld_wq) or NULL, [A] or NULL, [A] or NULL */
  int ld_wq;
  const float *keys;                            /* [T,B,A] or NULL */
  const void *enc;                              /* [T,B,E2] in enc_dtype */
  const int32_t *seq_len;                       /* [B] */
  const float *filt, *wfil, *alpha_zero;        /* carry_alpha: [taps,1,10], [10,A], zeros [B,T] */
  const float *live, *dmask;                    /* [To,B]; [To,B,U] or NULL */
  float *dec_in, *av_in, *alpha_all, *snorm_all;/* snorm_all [To,B] or NULL (sigmoid smoothing) */
  float *gates_all, *craw_all, *c_all, *h_all, *qz_all;   /* [To,B,4U], [To,B,U], [To+1,B,U] x2, [To,B,A] */
  float *work;                                  /* forward: B*(4U + U + T + E2) floats; backward: see asr_att_decoder_bwd */
  /* backward only */
  const float *dav_cell, *dav_ctx;              /* [To,B,U] (used as work space: overwritten), [To,B,E2] */
  float *dctx_all, *dpre_all, *dqz_all, *dv_all, *dpeep_all, *d_in_all;   /* [To,B,E2|4U|A|A or NULL|3U or NULL|Em+E2+U] */
  float *dkeys, *dwfil_rows, *dfilt_rows;       /* [T,B,A] += or NULL; carry_alpha: [B,10,A], [B,taps,10] */
  float *dc0, *dh0;                             /* [B,U] out: gradient w.r.t. the initial state */
  /* forward / inference, optional (abi 3): (Em+E2+U+1) x 4U floats of work space.  When given (and asr_lstm_cell_gemm_ok)
   * the loop writes the gate-interleaved image of W_cell | b_cell there once (asr_lstm_cell_gemm_prep) and every step runs
   * the cell-input product and the LSTM cell as ONE launch (asr_lstm_cell_gemm_fwd) -- same values, bit for bit. */
  float *W_cell_il;
  /* forward / inference / backward, optional: asr_lstm_cell_gemm_h_bytes(Em+E2+U, U) bytes of work space.  When given (and
   * asr_lstm_cell_gemm_ok, U % 16 == 0) the loops write the bf16 images of W_cell there (asr_lstm_cell_gemm_prep_h) and the
   * cell-input product (+ cell) of every forward step and the dpre W_cell^T product of every backward step stream bf16
   * weights: the rounding point of a bf16-operand model's decoder kernel (oracle: operand_round on lstm_cell/kernel).
   * Takes precedence over W_cell_il. */
  void *W_cell_h;
} asr_att_decoder;
int asr_att_decoder_fwd(asr_handle* h, const asr_att_decoder* a, asr_stream s);
/* ---- greedy inference, native ------------------------------------------------------------------------------ *
 * dynamic_decode over a GreedyEmbeddingHelper with impute_finished (attention_seq2seq.py:462-509,
 * decoders/dynamic_decoder.py:148-197): the reference's in-graph while_loop, here ONE call that issues, per step, the
 * forward step of asr_att_decoder_fwd, the attentional-vector FC + tanh, the output layer, and a selection kernel
 * (argmax -> emitted id, per-row finished flag, embedding of the id into the next step's input row, imputed context).
 * `a` as for asr_att_decoder_fwd with To = max_decode_length, dmask NULL, a->live == f->live ([To+1,B], row 0 set by the
 * caller: 1 for rows that decode), gates_all / craw_all / qz_all ONE step long (reused), snorm_all [To,B] or NULL;
 * dec_in row 0 = embedding(start token) | zero context | initial h.  Outputs: ids_all [To,B] (0 once a row has
 * finished), logits_all [To,B,C2] and av_all [To,B,U] (NOT imputed: multiply by live[k] for the reference's emitted
 * fields), a->alpha_all [To,B,T], live [To+1,B], live_count [To+1] (live rows at the START of step k; entry 0 is the
 * caller's).  The number of decoded steps is the first k with live_count[k] == 0 (or To).
 * Early exit: with host_live_count (pinned host int32 [To+1]) and check_every > 0 the call stops issuing steps once a
 * check point's asynchronous copy shows no live row; it waits only for the copy of two check points ago (the issue loop
 * stays 2-3 intervals ahead of the device, the pipeline is never drained); *steps_issued tells how many steps were
 * enqueued (>= the number decoded). */
typedef struct asr_att_infer {
  const float *W_av, *W_out, *b_out;            /* [U+E2,U], [U,C2], [C2] or NULL */
  const float *embedding;                       /* [vocabulary, Em] */
  int C2, eos;
  float *live;                                  /* [To+1,B] */
  float *av_all, *logits_all;                   /* [To,B,U], [To,B,C2] */
  int32_t *ids_all, *live_count;                /* [To,B], [To+1] */
  int32_t *host_live_count;                     /* pinned host [To+1] or NULL */
  int check_every;
} asr_att_infer;
int asr_att_decoder_infer(asr_handle* h, const asr_att_decoder* a, const asr_att_infer* f, int* steps_issued,
                          asr_stream s);
/* work: B*(5*U + 3*T + E2) floats.  dav_cell is consumed (its rows accumulate the query-path gradient in place). */
int asr_att_decoder_bwd(asr_handle* h, const asr_att_decoder* a, asr_stream s);
/* out[b, j] = x[b*ldx + j] + y[b*ldy + j], j < W (row blocks of wider arrays; out may alias x) */
int asr_add_cols(asr_handle* h, const float* x, int ldx, const float* y, int ldy, float* out, int ldo, int B, int W,
                 asr_stream s);
int asr_tanh_fwd(asr_handle* h, const float* x, float* y, size_t n, asr_stream s);
int asr_tanh_bwd(asr_handle* h, const float* dy, const float* y, float* dx, size_t n, asr_stream s);
/* tf.nn.embedding_lookup (attention_seq2seq.py:439) and its gradient (deterministic) */
int asr_embedding_gather(asr_handle* h, const float* W, const int32_t* ids, int rows, int E,
                         float* out, asr_stream s);
int asr_embedding_scatter(asr_handle* h, const float* dout, const int32_t* ids, int rows, int E,
                          int vocab, float* dW, asr_stream s);
/* sparse softmax cross-entropy per row of logits[rows,C] (+eps), weighted (sequence_loss,
 * attention_seq2seq.py:625-637): row_loss[r] = w[r]*xent; dlogits = (softmax-onehot)*w*dscale */
int asr_seq_xent(asr_handle* h, const float* logits, const int32_t* targets, const float* weights,
                 int rows, int C, float eps, float dscale, float* row_loss, float* dlogits,
                 asr_stream s);
int asr_argmax_rows(asr_handle* h, const float* x, int rows, int C, int32_t* out, asr_stream s);

/* ---- gradient clipping + optimizers -------------------------------------- *
 * Multi-tensor over one flat fp32 parameter buffer; tensor i is
 * [offsets[i], offsets[i+1]).  tf.clip_by_norm per variable
 * (models/model_base.py:148-152): g *= clip / max(||g||_2, clip). */
/* Norms are reduced in fixed 4096-element chunks (deterministic).  asr_clip_plan (host
 * helper) fills chunk_start_host[num_tensors+1] from HOST offsets; upload it once.
 * partial_ws: chunk_start[num_tensors] (= total_chunks) floats of device scratch. */
int asr_clip_plan(asr_handle* h, const int64_t* offsets_host, int num_tensors,
                  int64_t* chunk_start_host);
int asr_clip_by_norm_multi(asr_handle* h, float* grads, const int64_t* offsets,
                           const int64_t* chunk_start, int num_tensors, int64_t total_chunks,
                           float clip_norm, float* partial_ws, asr_stream s);
/* weight decay term of ctc.py:280-286: grads += wd * params on tensors with decay_mask[i]!=0;
 * l2_out (device fp32, may be NULL) receives wd * sum(0.5*||p||^2). */
int asr_weight_decay(asr_handle* h, float* grads, const float* params, const int64_t* offsets,
                     const uint8_t* decay_mask, int num_tensors, float wd, float* l2_out,
                     asr_stream s);
/* One optimizer step over n fp32 parameters with TF1 default hyper-parameters
 * (models/model_base.py:68-95).  slot0/slot1: optimizer state
 * (momentum / accumulators / m,v), `step` = 1-based step count (Adam bias correction). */
int asr_optimizer_step(asr_handle* h, int optimizer, float* params, const float* grads,
                       float* slot0, float* slot1, size_t n, float lr, int64_t step,
                       asr_stream s);
/* p[i] = (a[i] + b[i]) * scale helpers for the tower mean of utils/training/multi_gpu.py:39-40 */
int asr_scale(asr_handle* h, float* x, size_t n, float scale, asr_stream s);

/* ---- data-parallel collective (RCCL over xGMI) ------------------------------------------------ *
 * The tower mean of utils/training/multi_gpu.py:13-48 (average_gradients: per variable, stack the N towers'
 * already-clipped gradients and reduce_mean) and its caller examples/librispeech/training/train_ctc.py:112-147, as
 * ONE all-reduce(sum) over the flat fp32 gradient buffer + x 1/world -- one process per GPU.  librccl.so is opened
 * with dlopen (asr_comm_set_library names it; default: the loader's librccl.so.1) so that a host that already has
 * one loaded (PyTorch) shares the instance.  Bootstrap as for NCCL: rank 0 calls asr_comm_unique_id (128 bytes,
 * host memory), the host program hands the bytes to every rank, every rank calls asr_comm_init. */
typedef struct asr_comm asr_comm;
int asr_comm_set_library(const char* path);
int asr_comm_unique_id(void* id128_host);
int asr_comm_init(asr_comm** out, asr_handle* h, int rank, int world, const void* id128_host);
int asr_comm_destroy(asr_comm* c);
int asr_comm_info(asr_comm* c, int* rank, int* world);
/* buf[0..n) <- mean over ranks, in place, asynchronous on `s`; every rank calls it with the same n.  buf: any 4-byte
 * aligned device pointer. */
int asr_allreduce_mean(asr_comm* c, float* buf, size_t n, asr_stream s);

#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H_ */
