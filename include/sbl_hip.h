/* libsbl_hip.so — C ABI of the MI355X-native (gfx950) SBL lip-reading hot path.
 *
 * The reference (VIPL SBL_For_Multilingual_Lip_Reading) has no FFI: its boundary is the
 * Python class surface of SBL_Multilingual_Lip_reading/transformer/ ("SBL/..." below).
 * Each entry point here replaces the torch ops one reference call site issues; the Python
 * mirror (sbl_for_multilingual_lip_reading_amd/transformer/) binds them through ctypes
 * (see INTEGRATION.md for the stub a maintainer adds to the reference).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless marked host.
 *   - the CALLER allocates every output and workspace; nothing is allocated inside.
 *   - every call only enqueues work on `stream` (hipStream_t passed as void*): no sync, no
 *     host read-back, so any sequence of calls can be captured into a hipGraph.
 *   - return 0 on success, otherwise a hipError_t value or SBL_ERR_INVALID (bad shape /
 *     alignment, checked on the host BEFORE any launch); text via sbl_last_error()
 *     (thread-local).  No global mutable state: safe from several host threads on distinct
 *     streams/devices (nn.DataParallel's threading model, SBL/train.py:115).
 *   - activations of the visual trunk are NHWC ("channels last"): (image, h, w, c), image =
 *     n*T + t — the reference's transpose/contiguous/view (SBL/transformer/video_frontend.py:113-115)
 *     is folded into the layout.  Transformer tensors are row-major (B, L, 512).
 *   - tensors are fp32 in memory everywhere (the reference's dtype).  The arithmetic of the matrix products of the
 *     tile engine is a process-wide setting, sbl_set_matmul_precision() below: exact fp32 MFMA
 *     (v_mfma_f32_32x32x2_f32, the library default) or split-bf16 MFMA with fp32 accumulation.  The setting is read
 *     when a launch is ENQUEUED: a captured hipGraph keeps the mode it was captured under, whatever is set later.
 */
#ifndef SBL_HIP_H
#define SBL_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SBL_ERR_INVALID (-22)
#define SBL_ABI_VERSION 1

typedef void* sbl_stream_t; /* hipStream_t */

const char* sbl_last_error(void);
int sbl_abi_version(void);

/* Bench instrumentation (debug; process-wide): between begin/end every GEMM / convolution launch gets a slot
 * stamps[2*slot] = min start, stamps[2*slot+1] = max end of its workgroups, in 100 MHz s_memrealtime ticks
 * (caller pre-fills starts with ~0 and ends with 0).  The slot pointer is baked into the launch, so stamps are
 * taken inside hipGraph replays too.  sbl_profile_last_slot/kernel report the launch just enqueued by this thread
 * (slot -1 = not instrumented; kernel 1 skinny GEMM, 2 tiled 64x64, 3 tiled 128x128, 4/5/6 conv fwd/dgrad/wgrad,
 * 7 merged decoder/encoder weight gradients); sbl_profile_used = slots handed out so far by ALL threads (autograd
 * runs backward on its own thread, so a caller brackets a call with it to learn which slots the call used). */
int sbl_profile_begin(uint64_t* stamps, int capacity);
int sbl_profile_end(void);
int sbl_profile_used(void);
int sbl_profile_last_slot(void);
int sbl_profile_last_kernel(void);

/* Precision of the matrix products of the tile engine (dense GEMMs and trunk convolutions), process-wide, default 0:
 *   0  exact fp32 MFMA (v_mfma_f32_32x32x2_f32), bitwise an fmaf chain — the reference's arithmetic;
 *   6  every operand split exactly into three bf16 planes on the way into LDS, six bf16 MFMA products per fp32
 *      product (all plane pairs of weight >= 2^-16), fp32 accumulation: fp32-grade results (dropped terms <= 2^-26
 *      per product) at 6 x 32 instead of 8 x 64 MFMA cycles per 32x32x16 block;
 *   3  two planes, three products (~2^-17 per product);   1  plain bf16 inputs with fp32 accumulation — BASELINE
 *      config 5 "mixed bf16" (fp32 master weights, fp32 accumulate).
 * Inputs and outputs stay fp32 in memory in every mode.  Returns SBL_ERR_INVALID for any other value.
 * The value is read at enqueue time and baked into captured hipGraphs (re-capture after changing it).  Mode 6's
 * "exact split" holds for |x| >= 2^-110 or x == 0; residual planes of smaller magnitudes underflow bf16's range. */
int sbl_set_matmul_precision(int terms);
int sbl_get_matmul_precision(void);
/* Measurement knobs (process-wide, read at enqueue time like the precision; results are the same either way).
 * knob 0: wave-group K split of the dense 64x64 split-bf16 tiles (512-thread workgroups), 1 = on (default), 0 = off;
 * knob 1: number of 64x64 output tiles from which a dense product takes 128x128 tiles (default 4096);
 * knob 2: largest tile count of a launch that takes the wave-group K split (default 320);
 * knob 3: stride-2 convolution weight gradients on 64x64 tiles (1, default) or by the general rule (0);
 * knob 4: workgroup target of their split-K (default 1536; 0 = the general rule);
 * knob 5: patch-resident 3x3 / stride-1 convolution kernel for the 22x22 and 11x11 trunk maps: 2 (default) swizzled 32-channel LDS rows,
 *         two workgroups per CU; 1 padded 64-channel rows, one workgroup per CU; 0 the per-tap gather kernels;
 * knob 6: cap on the workgroups of the grouped weight-gradient launch (0 = one per tile, default);
 * knob 7: position-major convolution weight gradients with Cout <= value on 64x64 tiles (default 512; 0 = 128x128 tiles);
 * knob 8: most images per tile of the patch-resident kernel (default 0 = as many as fit, i.e. two 11x11 maps; 1 = one,
 *         which leaves the 11x11 layer on the position-major kernels);
 * knob 9: patch-resident weight gradient of the 3x3 / stride-1 convolutions for maps of at least `value` pixels (default 30:
 *         the 22x22, 11x11 and 6x6 layers; 0 = the implicit-GEMM weight gradients everywhere);
 * knob 10 / 11: workgroup target (default 256) and largest split count (default 8) of the in-launch split-K of sbl_gemm2_f32;
 * knob 12: stem weight gradient with operands split once into LDS planes and transposed LDS reads (1, default: 635 us) or
 *          the split-per-use kernel (0: 801 us);
 * knob 13: phase ablation of that variant (measurement only: results are WRONG while it is non-zero);
 * knob 14: stem forward convolution with eight wavefronts (two tiles in flight) on one copy of the weight planes (1, default:
 *          434 us) or the four-wavefront kernel (0: 497 us). */
int sbl_set_tuning(int knob, int value);

/* ---------------------------------------------------------------- dense GEMM / Linear
 * C[M,N] (+)= opA(A)[M,K] * opB(B)[K,N], row-major; opA(A)[m,k] = transA ? A[k*lda+m] : A[m*lda+k],
 * opB(B)[k,n] = transB ? B[n*ldb+k] : B[k*ldb+n].  Epilogue: +bias[n], ReLU, or multiply by
 * (relu_mask[m*ldm+n] > 0) (ReLU backward fused into the producing GEMM).
 * accumulate: 0 = overwrite, 1 = C += (in place: gradients accumulate straight into the flat .grad buffer).
 * a_colsum (transA=1 only, may be NULL): float[M] += sum_k opA(A)[m,k], float atomics — the bias gradient
 *   db = sum_rows dY rides on the weight-gradient GEMM dW = dY^T X.
 * ws / ws_bytes (may be NULL/0): split-K workspace = int[4096] tile counters (zero on first hand-over, left
 *   zero by every call) followed by fp32 partial slabs.  With it, small-M problems split K across workgroups
 *   and the last-arriving workgroup of each tile reduces the slabs and runs the epilogue inside the same
 *   launch; without it split-K falls back to float atomics on C (plain epilogue only).  One workspace per
 *   stream: concurrent calls must not share it.
 * Replaces nn.Linear forward/backward: SBL/transformer/attention.py:16-18,27,41-43,57;
 * module.py:42-43,49; encoder.py:27,54; decoder.py:59-60,166-167. */
int sbl_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                 float* C, long ldc, const float* bias, int relu, const float* relu_mask, long ldm, int accumulate,
                 float* a_colsum, void* ws, long ws_bytes, sbl_stream_t stream);
/* Two products of one shape in ONE launch: C_d[M,N] = A_d[M,K] * B_d[N,K]^T (+ bias_d) (ReLU), d = 0, 1 - the nn.Linear
 * forward of layer_stack_l2r[i] and layer_stack_r2l[i], which SBL/transformer/decoder.py:121-156 runs back to back on
 * same-shape inputs.  Kernel boundaries cost ~5 us and small launches on two streams do not overlap on this GPU, so
 * the two decoder directions share launches.  Same kernels and workspace convention as sbl_gemm_f32. */
int sbl_gemm2_f32(int M, int N, int K, const float* A0, const float* A1, long lda, const float* B0, const float* B1,
                  long ldb, float* C0, float* C1, long ldc, const float* bias0, const float* bias1, int relu, void* ws,
                  long ws_bytes, sbl_stream_t stream);
/* Deferred weight gradient of one decoder weight over all stages of a step:
 * C[M,N] += sum_s A_s^T B_s with A_s (seg_rows[s] x M, row stride lda) = dY of stage s and B_s (seg_rows[s] x N) = its
 * input; a_colsum[m] += column sums of the A_s (bias gradient).  A_ptrs / B_ptrs / seg_rows are HOST arrays of nseg
 * <= 16 entries (device pointers inside).  C and a_colsum are accumulated with float atomics (split-K). */
int sbl_wgrad_seg_f32(int nseg, const float* const* A_ptrs, long lda, const float* const* B_ptrs, long ldb,
                      const int* seg_rows, int M, int N, float* C, long ldc, float* a_colsum, sbl_stream_t stream);
/* The same for MANY weights in one launch (all deferred decoder weights of a step): every 128x128 tile of every
 * C_p (+)= sum_s A_{p,s}^T B_{p,s} is owned by one workgroup over the whole K = sum(seg_rows) (no split-K, no
 * atomics on C: deterministic).  All problems share nseg / seg_rows (multiples of 16 rows); A_ptrs / B_ptrs are HOST
 * arrays of nprob*nseg device pointers (problem-major); lda/ldb/M/N/C/ldc/colsum HOST arrays of nprob entries
 * (colsum[p] may be NULL).  table: device scratch of sbl_wgrad_group_table_bytes(nprob) bytes (problem descriptors,
 * written by the call on `stream`). */
long sbl_wgrad_group_table_bytes(int nprob);
int sbl_wgrad_group_f32(int nprob, int nseg, const int* seg_rows, const float* const* A_ptrs, const long* lda,
                        const float* const* B_ptrs, const long* ldb, const int* M, const int* N, float* const* C,
                        const long* ldc, float* const* colsum, void* table, long table_bytes, sbl_stream_t stream);
/* out[n] (+)= sum_m X[m*ldx + n]   (bias gradients) */
int sbl_colsum_f32(const float* X, long ldx, float* out, int M, int N, int accumulate, sbl_stream_t stream);

/* ---------------------------------------------------------------- stem (frontend3D)
 * Conv3d(1,64,(5,7,7),s(1,2,2),p(2,3,3),bias=False) -> BatchNorm3d(64) -> ReLU ->
 * MaxPool3d((1,3,3),s(1,2,2),p(0,1,1)): SBL/transformer/video_frontend.py:99-104.
 * x: (N,T,H,W) fp32 (C=1).  conv_out: (N*T,Ho,Wo,64), Ho=H/2, Wo=W/2.  pooled: (N*T,Ho/2,Wo/2,64).
 * stats: double[128] = per-channel (sum, sumsq) of conv_out; zeroed by the call. */
int sbl_stem_conv_fwd(const float* x, const float* w /*[64][245]*/, float* conv_out, double* stats, int N, int T, int H,
                      int W, sbl_stream_t stream);
int sbl_stem_bn_relu_pool_fwd(const float* conv_out, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, float* pooled, uint8_t* argmax, int NT, int Ho, int Wo,
                              sbl_stream_t stream);
/* backward pass 1: sums = double[128] (sum g, sum g*xhat), g = grad wrt BN output after pool/ReLU adjoints */
int sbl_stem_bwd_reduce(const float* conv_out, const float* dpooled, const uint8_t* argmax, const float* mean,
                        const float* invstd, const float* gamma, const float* beta, double* sums, int NT, int Ho,
                        int Wo, sbl_stream_t stream);
/* backward pass 2: dconv recomputed on the fly and contracted with the input patches:
 * dw[64][245] (zeroed by the call), dgamma[64], dbeta[64].  No input gradient (x is data). */
int sbl_stem_wgrad(const float* x, const float* conv_out, const float* dpooled, const uint8_t* argmax,
                   const float* mean, const float* invstd, const float* gamma, const float* beta, const double* sums,
                   float* dw, float* dgamma, float* dbeta, int N, int T, int H, int W, sbl_stream_t stream);

/* ---------------------------------------------------------------- BatchNorm (train / eval)
 * nn.BatchNorm{2,3}d defaults: SBL/transformer/video_frontend.py:21,24,71,101. */
int sbl_bn_finalize(const double* stats /*[2C]*/, long count, float* running_mean, float* running_var,
                    float momentum, float eps, float* save_mean, float* save_invstd, int C,
                    int64_t* num_batches_tracked /* += 1, or NULL */, sbl_stream_t stream);
int sbl_bn_eval_stats(const float* running_mean, const float* running_var, float eps, float* mean, float* invstd,
                      int C, sbl_stream_t stream);
/* y = [relu]( gamma*(x-mean)*invstd + beta [+ res] ), NHWC rows x C */
int sbl_bn_apply_fwd(const float* x, const float* res, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, float* y, long rows, int C, int relu, sbl_stream_t stream);
/* Training form: sbl_bn_finalize folded into the apply launch - mean / invstd are derived from the convolution epilogue's
 * (sum, sumsq) statistics inside the kernel (same double arithmetic), save_mean / save_invstd, the running statistics and
 * num_batches_tracked are written by the same launch.  C/4 must divide 256. */
int sbl_bn_apply_fwd_stats(const float* x, const float* res, const double* stats, long count, float* running_mean,
                           float* running_var, float momentum, float eps, const float* gamma, const float* beta, float* y,
                           float* save_mean, float* save_invstd, int64_t* num_batches_tracked, long rows, int C, int relu,
                           sbl_stream_t stream);
/* sums = double[2C] (sum g, sum g*xhat), g = dy * (y>0 if relu); overwritten by the call.
 * ws: NULL or the calling stream's sbl_gemm_f32 workspace (>= 16 KiB of int counters that are zero between launches,
 * then fp32 scratch): block partials + a last-arriver reduction replace 2C contended double atomics per block. */
int sbl_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                      double* sums, long rows, int C, int relu, void* ws, long ws_bytes, sbl_stream_t stream);
/* dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); dres = g (if non-null); dgamma, dbeta from sums
 * (accumulate != 0: += into the persistent gradient buffers) */
int sbl_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                     const float* gamma, const double* sums, float* dx, float* dres, float* dgamma, float* dbeta,
                     long rows, int C, int relu, int accumulate, sbl_stream_t stream);

/* ---------------------------------------------------------------- ResNet-18 trunk convolutions
 * conv3x3 / 1x1-stride-2, bias-free, NHWC implicit GEMM: SBL/transformer/video_frontend.py:10-12,69-70.
 * Weights are used in OHWI order [Cout][KH][KW][Cin]; pack/unpack convert from/to the
 * reference's OIHW parameter layout (state-dict shapes stay the reference's). */
int sbl_conv_weight_pack(const float* w_oihw, float* w_ohwi, float* w_dgrad /*[Cin][KH][KW][Cout] or NULL*/, int Cout,
                         int Cin, int KH, int KW, double* zero /* nzero doubles set to 0 by the same launch (the BN
                         statistics of the convolution that follows), or NULL */, int nzero, sbl_stream_t stream);
/* accumulate != 0: dw_oihw += (the persistent flat gradient buffer of a data-parallel replica) */
int sbl_conv_wgrad_unpack(const float* dw_ohwi, float* dw_oihw, int Cout, int Cin, int KH, int KW, int accumulate,
                          sbl_stream_t stream);
/* stats: NULL or double[2*Cout] (sum, sumsq of y) accumulated by the epilogue; zeroed by the call unless stats_zeroed.
 * ws / ws_bytes (may be NULL/0): the calling stream's sbl_gemm_f32 workspace.  With it, a launch whose tile count
 * is not a multiple of the 256 CUs runs its last partial round of tiles split along K (in-launch slab reduction). */
int sbl_conv2d_fwd(const float* x, const float* w_ohwi, float* y, double* stats, int stats_zeroed, int NIMG, int H, int W, int Cin,
                   int Cout, int KH, int KW, int stride, int pad, void* ws, long ws_bytes, sbl_stream_t stream);
int sbl_conv2d_dgrad(const float* dy, const float* w_dgrad, float* dx, int NIMG, int H, int W, int Cin, int Cout,
                     int KH, int KW, int stride, int pad, void* ws, long ws_bytes, sbl_stream_t stream);
/* The same, and in its epilogue the reduction pass of the BatchNorm backward that consumes dx: dx is the gradient of
 * act = relu(bn(pre)) (video_frontend.py:31-33: conv1 -> bn1 -> relu feeds conv2), so
 * sums[c] = sum_pixels g, sums[Cin + c] = sum_pixels g * (pre - mean[c]) * invstd[c] with g = dx * (act > 0) - what
 * sbl_bn_bwd_reduce(dx, act, pre, ...) would compute in a separate pass over the three tensors.  Stride 1 only. */
int sbl_conv2d_dgrad_bnstats(const float* dy, const float* w_dgrad, float* dx, int NIMG, int H, int W, int Cin, int Cout,
                             int KH, int KW, int stride, int pad, void* ws, long ws_bytes, const float* act,
                             const float* pre, const float* mean, const float* invstd, double* sums,
                             int sums_zeroed /* sums already hold zeros (a pooled memset): skip the call's own */,
                             sbl_stream_t stream);
/* The general form (any stride; every fused operand may be NULL) - what BasicBlock's backward needs so that no separate
 * pass touches dx (video_frontend.py:28-41):
 *   addend  the residual branch's gradient, added before the store and the sums.  Stride 1: laid out like dx (identity
 *           shortcut).  Stride 2: the COMPACT (NIMG, ceil(H/2), ceil(W/2), Cin) gradient of the 1x1 / stride-2 downsample
 *           branch (sbl_conv1x1s2_dgrad_compact), which lives on the even/even pixels only.
 *   act, pre, mean, invstd  as in sbl_conv2d_dgrad_bnstats: the BatchNorm whose output gradient dx (with the addend) is -
 *           the previous block's bn2; sums[0..2Cin).
 *   pre2, mean2, invstd2    a second BatchNorm fed through the same act (that block's downsample branch):
 *           sums[2Cin..4Cin) = (sum g, sum g * (pre2 - mean2) * invstd2).  sums: double[2*Cin] or double[4*Cin]. */
int sbl_conv2d_dgrad_fused(const float* dy, const float* w_dgrad, float* dx, int NIMG, int H, int W, int Cin, int Cout,
                           int KH, int KW, int stride, int pad, void* ws, long ws_bytes, const float* addend,
                           const float* act, const float* pre, const float* mean, const float* invstd, const float* pre2,
                           const float* mean2, const float* invstd2, double* sums, int sums_zeroed, sbl_stream_t stream);
/* Input gradient of the 1x1 / stride-2 downsample convolution on its own support: dx_compact (NIMG, ceil(H/2), ceil(W/2),
 * Cin) = dy (NIMG, ceil(H/2), ceil(W/2), Cout) * w; the other three quarters of the full-size gradient are zeros that
 * nobody needs to write (sbl_conv2d_dgrad_fused adds the compact form to conv1's gradient). */
int sbl_conv1x1s2_dgrad_compact(const float* dy, const float* w_dgrad, float* dx_compact, int NIMG, int H, int W, int Cin,
                                int Cout, void* ws, long ws_bytes, sbl_stream_t stream);
/* dw_ohwi zeroed by the call (unless dw_zeroed: the caller hands over zeros), then split-K float atomics */
int sbl_conv2d_wgrad(const float* x, const float* dy, float* dw_ohwi, int NIMG, int H, int W, int Cin, int Cout,
                     int KH, int KW, int stride, int pad, int dw_zeroed, sbl_stream_t stream);
/* AdaptiveAvgPool2d(1): (NIMG,HW,C) -> (NIMG,C): video_frontend.py:53,87-88 */
int sbl_avgpool_fwd(const float* x, float* y, int NIMG, int HW, int C, sbl_stream_t stream);
int sbl_avgpool_bwd(const float* dy, float* dx, int NIMG, int HW, int C, sbl_stream_t stream);

/* ---------------------------------------------------------------- dropout (fused Philox-style masks)
 * y = x * keep / (1-p); the mask is a function of (*seed, offset, element index) and is
 * regenerated in backward.  F.dropout(p=0.5) video_frontend.py:122; nn.Dropout(0.1) x80. */
int sbl_dropout(const float* x, float* y, long n, float p, const uint64_t* seed, uint64_t offset,
                sbl_stream_t stream);
int sbl_seed_bump(uint64_t* seed, sbl_stream_t stream);

/* ---------------------------------------------------------------- LayerNorm with fused residual
 * y = LN(x + res) * gamma + beta (eps 1e-5), rows of D=512: attention.py:58, module.py:51, encoder.py:54. */
/* With drop_p > 0 the sub-layer's dropout is fused: y = LN(dropout(x) + res) (attention.py:57-58,
 * module.py:50-51); the mask is a function of (*seed, offset, element index). */
int sbl_add_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                          float* mean, float* rstd, int M, int D, float eps, float drop_p, const uint64_t* seed,
                          uint64_t offset, sbl_stream_t stream);
/* sbl_add_layernorm2_fwd followed by the SBL cross-direction fusion (SBL/transformer/module.py:50-51 then decoder.py:127-143)
 * in one launch: xn0 = A + flip(B), xn1 = 2B + flip(A), A / B = the two directions' LayerNorm(dropout(x_d) + res_d) rows of a
 * ragged stage (B sequences per segment of length seg_L[s]; flip along each sequence's own prefix).  A / B are not stored. */
int sbl_add_layernorm2_fusion_fwd(const float* x0, const float* x1, const float* res0, const float* res1, const float* gamma0,
                                  const float* gamma1, const float* beta0, const float* beta1, float* xn0, float* xn1, float* mean0,
                                  float* mean1, float* rstd0, float* rstd1, int B, const int* seg_L, int nseg, int D, float eps,
                                  float drop_p, const uint64_t* seed, uint64_t offset0, uint64_t offset1, sbl_stream_t stream);
/* The same for two same-shape problems (the two decoder directions) in one launch. */
int sbl_add_layernorm2_fwd(const float* x0, const float* x1, const float* res0, const float* res1, const float* gamma0,
                           const float* gamma1, const float* beta0, const float* beta1, float* y0, float* y1, float* mean0,
                           float* mean1, float* rstd0, float* rstd1, int M, int D, float eps, float drop_p,
                           const uint64_t* seed, uint64_t offset0, uint64_t offset1, sbl_stream_t stream);
/* dz = gradient of the LayerNorm input (= dres); dx_drop (may be NULL) = gradient of the pre-dropout x;
 * dgamma/dbeta accumulated with float atomics (caller zeroes, or passes the .grad buffers to accumulate) */
int sbl_add_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma, const float* mean,
                          const float* rstd, float* dz, float* dx_drop, float* dgamma, float* dbeta, int M, int D,
                          float drop_p, const uint64_t* seed, uint64_t offset, sbl_stream_t stream);
/* y = x + pe[l] (l = row % L) : encoder.py:53-55 positional add */
int sbl_add_pe(const float* x, const float* pe, float* y, int B, int L, int D, sbl_stream_t stream);

/* y[m,:] = x[m,:] * s[m]: the `*= non_pad_mask` of encoder.py:86,89 / decoder.py:399,403,406 (ragged lengths) */
int sbl_rowscale(const float* x, const float* s, float* y, long M, int D, sbl_stream_t stream);

/* ---------------------------------------------------------------- scaled dot-product attention
 * One workgroup per (batch, head): S = Q K^T * scale, mask, softmax over keys, [dropout], O = P V.
 * q/k/v/o are (B, L, H*64) row-major views with row strides ldq/ldk/ldv/ldo (heads are the
 * contiguous 64-wide column blocks: attention.py:41-47); p_out is (H*B, Lq, Lk) head-major like
 * the reference's returned attn.  mask_kind: 0 none, 1 causal (key > query masked:
 * utils.py:116-124), 2 explicit uint8 (B,Lq,Lk), nonzero = masked.  Lq, Lk <= 64, d = 64.
 * Replaces attention.py:72-83. */
int sbl_attention_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* o, long ldo,
                      float* p_out, int mask_kind, const uint8_t* mask, int B, int H, int Lq, int Lk, float scale,
                      float drop_p, const uint64_t* seed, uint64_t offset, sbl_stream_t stream);
int sbl_attention_bwd(const float* dout, long lddo, const float* q, long ldq, const float* k, long ldk, const float* v,
                      long ldv, const float* p, float* dq, long lddq, float* dk, long lddk, float* dv, long lddv,
                      int B, int H, int Lq, int Lk, float scale, float drop_p, const uint64_t* seed, uint64_t offset,
                      sbl_stream_t stream);

/* Ragged ("segmented") forms.  A run of decoder steps whose input tokens are all known (teacher-forced) has no
 * step-to-step dependency (decoder.py:176-186 feeds back the argmax only when the coin says so), so the run is
 * processed as ONE batch: segment s = the step with prefix length seg_L[s]; its B*seg_L[s] rows follow segment
 * s-1's rows, in (b, l) order.  seg_L is a HOST array of nseg <= 16 lengths.  Lk_fixed == 0: self-attention inside
 * each segment; Lk_fixed > 0: all segments attend to the same (B, Lk_fixed) key/value rows (cross-attention); in
 * backward the segments' dk/dv contributions are summed inside the call (dq, dk, dv are always overwritten).
 * p_out holds the segments' (H*B, L, Lk) probability blocks back to back. */
int sbl_attention_seg_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* o,
                          long ldo, float* p_out, int mask_kind, const uint8_t* mask, int B, int H, const int* seg_L,
                          int nseg, int Lk_fixed, float scale, float drop_p, const uint64_t* seed, uint64_t offset,
                          sbl_stream_t stream);
/* sbl_attention_seg_fwd for two same-shape problems (the two decoder directions) in one launch; mask_kind 0 or 1. */
int sbl_attention_seg2_fwd(const float* q0, const float* q1, long ldq, const float* k0, const float* k1, long ldk,
                           const float* v0, const float* v1, long ldv, float* o0, float* o1, long ldo, float* p_out0,
                           float* p_out1, int mask_kind, int B, int H, const int* seg_L, int nseg, int Lk_fixed, float scale,
                           float drop_p, const uint64_t* seed, uint64_t offset0, uint64_t offset1, sbl_stream_t stream);
int sbl_attention_seg_bwd(const float* dout, long lddo, const float* q, long ldq, const float* k, long ldk, const float* v,
                          long ldv, const float* p, float* dq, long lddq, float* dk, long lddk, float* dv, long lddv,
                          int B, int H, const int* seg_L, int nseg, int Lk_fixed, float scale, float drop_p,
                          const uint64_t* seed, uint64_t offset, sbl_stream_t stream);
/* Stage head of the SBL decoder for BOTH directions in one launch: out_d = dropout(emb[tok_d] + pe[:L]) for every segment
 * of the stage (SBL/transformer/decoder.py:116-120); mask = f(seed, offset_d, element index inside the stage's rows), the
 * indexing of sbl_dropout, which regenerates it in backward.  drop_p == 0: plain embedding + PE. */
int sbl_embed_pe_drop2_fwd(const int64_t* tok0, const int64_t* tok1, long ldt, const float* emb, const float* pe, float* out0,
                           float* out1, int B, const int* seg_L, int nseg, int D, int V, float drop_p, const uint64_t* seed,
                           uint64_t offset0, uint64_t offset1, sbl_stream_t stream);
/* Stage tail of the SBL decoder in one launch (SBL/transformer/decoder.py:160-186): the last cross-direction fusion at the
 * last position of every sequence (A'[L-1] = A[L-1] + B[0], B'[L-1] = 2 B[L-1] + A[0]) -> last_d (nseg*B, 512), the two
 * bias-free Linear(512, V) heads -> pred_d (nseg*B, ldp), and - write_tok != 0 - ys_d[b, step + 1] = argmax of the stage's
 * final segment (first maximal index).  yf_d: the last layer's outputs of the stage's rows; D must be 512, V <= 64. */
int sbl_decoder_tail_fwd(const float* yf0, const float* yf1, const float* w0, const float* w1, float* last0, float* last1,
                         float* pred0, float* pred1, long ldp, int64_t* ys0, int64_t* ys1, long ldy, int step, int write_tok,
                         int B, const int* seg_L, int nseg, int D, int V, sbl_stream_t stream);
int sbl_embed_pe_seg_fwd(const int64_t* tok, long ldt, const float* emb, const float* pe, float* out, int B,
                         const int* seg_L, int nseg, int D, int V, sbl_stream_t stream);
int sbl_embed_seg_bwd(const int64_t* tok, long ldt, const float* dy, float* demb, int B, const int* seg_L, int nseg, int D,
                      int V, sbl_stream_t stream);
int sbl_fusion_seg_fwd(const float* a, const float* b, float* a2, float* b2, int B, const int* seg_L, int nseg, int D,
                       sbl_stream_t stream);
int sbl_fusion_seg_bwd(const float* da2, const float* db2, float* da, float* db, int B, const int* seg_L, int nseg, int D,
                       sbl_stream_t stream);
/* out[s*B + b, :] = x[last row of sequence (s, b)]: the rows the output heads read (decoder.py:166-167);
 * bwd zero-fills dx (all rows) and scatters the nseg*B gradient rows back */
int sbl_gather_last_fwd(const float* x, float* out, int B, const int* seg_L, int nseg, int D, sbl_stream_t stream);
int sbl_gather_last_bwd(const float* dy, float* dx, int B, const int* seg_L, int nseg, int D, sbl_stream_t stream);

/* ---------------------------------------------------------------- SBL decoder pieces
 * out[b,l,:] = emb[tok[b*ldt + l]] + pe[l]: decoder.py:116-120 */
int sbl_embed_pe_fwd(const int64_t* tok, long ldt, const float* emb, const float* pe, float* out, int B, int L, int D,
                     int V, sbl_stream_t stream);
/* demb[tok] += dy (float atomics) */
int sbl_embed_bwd(const int64_t* tok, long ldt, const float* dy, float* demb, int B, int L, int D, int V,
                  sbl_stream_t stream);
/* A' = A + flip_t(B), B' = 2B + flip_t(A): closed form of the aliased loops decoder.py:132-143,160-164 */
int sbl_fusion_fwd(const float* a, const float* b, float* a2, float* b2, int B, int L, int D, sbl_stream_t stream);
int sbl_fusion_bwd(const float* da2, const float* db2, float* da, float* db, int B, int L, int D,
                   sbl_stream_t stream);
/* Decoder.preprocess (decoder.py:62-77): strip IGNORE_ID, <sos> + ids (input form) / ids (label form), both padded with <eos>
   to maxlen; int64 (N,To) -> two int64 (N,maxlen).  padded1 != NULL: a second target set (the r2l direction) in the same launch. */
int sbl_decoder_preprocess(const int64_t* padded0, const int64_t* padded1, int64_t* ys_in0, int64_t* ys_out0, int64_t* ys_in1,
                           int64_t* ys_out1, int N, int To, int maxlen, int64_t sos, int64_t eos, int64_t ignore,
                           sbl_stream_t stream);
/* ys[b, step+1] = use_argmax ? argmax_c pred[b,c] : gold[b, step]: decoder.py:173-186.
 * use_argmax: host int, or if coins_dev != NULL coins_dev[step] (device int32, for graph replay). */
int sbl_argmax_select(const float* pred, long ldp, const int64_t* gold, long ldg, int64_t* ys, long ldy, int step,
                      int use_argmax, const int32_t* coins_dev, int B, int V, sbl_stream_t stream);

/* ---------------------------------------------------------------- label-smoothed cross entropy
 * SBL/transformer/loss.py:27-52.  out[0]=sum of row losses, out[1]=#valid rows, out[2]=#correct
 * (zeroed by the call).  bwd: dpred = gscale[0] / out[1] * (softmax - q) on valid rows. */
int sbl_smoothed_ce_fwd(const float* pred, const int64_t* gold, float* out3, int R, int C, float eps, int ignore_id,
                        sbl_stream_t stream);
int sbl_smoothed_ce_bwd(const float* pred, const int64_t* gold, const float* out3, const float* gscale, float* dpred,
                        int R, int C, float eps, int ignore_id, sbl_stream_t stream);

/* ---------------------------------------------------------------- device input pipeline
 * uint8 grayscale frames (N,Tin,Hin,Win) -> fp32 clips (N,Tout,Hc,Wc): out = lut256[in[n, src_frame[n,t], y1[n]+y,
 * x1[n] + (flip[n] ? Wc-1-x : x)]], zero where src_frame < 0.  lut256[v] = float32((v/255. - mean)/std) computed in
 * double by the caller: bit-identical to SBL/data_gen.py:122-125 + cvtransforms.py:44-48 (ColorNormalize), :22-33 /
 * :7-19 (Random/CenterCrop), :36-41 (HorizontalFlip), data_gen.py:104-108 (FrameRemoval as a source-frame map) and
 * :290-296 (zero padding to 30 frames).  y1/x1/flip: device int32[N]; src_frame: device int32[N*Tout]. */
int sbl_preprocess_clips(const uint8_t* in, float* out, const float* lut256, const int* y1, const int* x1, const int* flip,
                         const int* src_frame, int N, int Tin, int Hin, int Win, int Tout, int Hc, int Wc,
                         sbl_stream_t stream);

/* ---------------------------------------------------------------- fused Adam
 * torch.optim.Adam(betas=(0.9,0.98), eps=1e-9) over a flat fp32 buffer, grad pre-scaled by
 * grad_scale (1/world_size): SBL/train.py:75, SBL/transformer/optimizer.py:18-27. */
int sbl_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  int step, float grad_scale, sbl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
