/* curla_hip.h -- C ABI of libcurla_hip.so: the MI355X (gfx950) kernels behind the
 * CURL/SAC learner hot path of paulvantieghem/curla.
 *
 * The reference has no native layer (it is pure PyTorch); what each entry point
 * replaces is therefore the ATen work dispatched by the cited reference lines
 * (paths relative to the reference checkout).  The boundary is deliberately
 * plain C: device pointers, sizes, a hipStream_t passed as void*.  No torch
 * types, no exceptions across the boundary, no allocation, no host sync -- every
 * call only enqueues kernels on `stream`, so callers may capture them into a
 * hipGraph.  Return value: CURLA_OK or a negative CURLA_ERR_* code.
 *
 * Layouts:
 *   activations        float32 NHWC  [B][H][W][32]
 *   replay frames      uint8   NHWC  [capacity][H][W][C]   (C = 3 * frame_stack)
 *   conv weights       float32 OIHW  (the reference's Parameter layout)
 *   dense weights      float32 [out][in] row-major (nn.Linear layout); the encoder
 *                      fc weight has its input columns in (y,x,c) order
 *   twin-Q tensors     two equally shaped blocks separated by a `twin_stride`
 */
#ifndef CURLA_HIP_H
#define CURLA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CURLA_OK 0
#define CURLA_ERR_ARG (-1)         /* null / misaligned pointer, non-positive size */
#define CURLA_ERR_LAUNCH (-2)      /* HIP reported a launch/attribute error */
#define CURLA_ERR_UNSUPPORTED (-3) /* shape outside what the kernels are built for (cf. NotImplementedError, encoder.py:47) */

const char* curla_version(void);

/* The C ABI's number: bumped whenever an entry point's argument list changes (round 4 put dyn / dyn64 / rng_dev
 * pointers into the middle of the Adam and policy-head calls: 4 -> 5 names that).  A binding written for another
 * number must refuse the library -- curla_amd/_lib.py does -- instead of calling with shifted arguments. */
#define CURLA_ABI_VERSION 6
int curla_abi_version(void);

/* Run-time kernel-selection options (curla_amd/csrc/options.h).  Every option's default is the measured-best path;
 * the other values are fallbacks for shapes the default does not take or A/B partners for measurements, and every
 * value runs under the whole-update parity test (tests/test_gpu_switches.py).  The environment variable
 * CURLA_<NAME> sets the initial value (read once, at first use); afterwards only these calls change it.
 *   conv1_u8    auto | hybrid | band | rw | rwb   first layer from the uint8 ring (utils.py:151-166 + encoder.py:78-81); auto = rwb
 *                                      (LDS-free row walk on the bf16 matrix cores: a uint8 pixel is exact in bf16) where 3 C <= 32, else rw
 *   conv1_f32   rw | band              first layer and its weight gradient from a float NHWC minibatch
 *   s1_fwd      auto | f23 | f43 | b3  stride-1 forward / data gradient: Winograd F(2,3) or F(4,3) along x on the f32-input MFMA, or
 *                                      b3 (= auto): fp32 operands as three bf16 parts on the bf16 matrix cores behind F(2,3)
 *   bwd_split   auto | 0 | 1           stride-1 backward: 2 + 2 workgroups per CU, or 1 + 1 side by side
 *   gemm_tile   auto | 6464 | 6432 | 3232 | 12864   tile of the tiled GEMM (12864: the 128 x 64 bf16x3 tile wherever it applies)
 *   linear_bwd  pair | split           dW and dx of a linear layer in one launch or two
 *   gemm_mfma   auto | f32 | b3        arithmetic of the tiled GEMM (and of the encoder fc forward: bf16x3 unless f32): f32 = the
 *                                      f32-input MFMA; b3 = interior aligned tiles with fp32
 *                                      operands as three bf16 parts on the bf16 matrix cores, split once when a tile is staged;
 *                                      auto = b3 on 128 x 64 tiles where those give every CU a workgroup, f32 elsewhere
 *   s1_wgrad    auto | x | xy          stride-1 weight gradient: Winograd F(3,2) along x, or (auto) in both directions (2 x 2 gradient
 *                                      blocks: a third fewer f32 matrix instructions)
 *   wgrad1_u8   auto | f32 | b16       first-layer weight gradient from the uint8 ring: on the f32-input MFMA, or (auto, where
 *                                      3 C <= 32 and output rows hold >= 8 pixels) on the bf16 matrix cores -- a uint8 pixel is
 *                                      exact in one bf16, the gradient is split into three: nothing is dropped
 * curla_set_option returns CURLA_ERR_ARG for an unknown name or value; curla_get_option NULL for an unknown name. */
int curla_set_option(const char* name, const char* value);
const char* curla_get_option(const char* name);

/* ---- encoder convolutions (encoder.py:54-63 construction, :77-90 forward) ---- */

/* First layer: 3x3 stride 2, C -> 32, + bias + ReLU, with the minibatch assembly
 * fused into the load.  src_kind = 1: `src` is the uint8 replay ring
 * [N][Hs][Ws][C]; sample b reads frame idx[b] (NULL: b) cropped at
 * (h1[b], w1[b]) (NULL: 0) to Hc x Wc -- replaces utils.py:151-166 (gather,
 * RandomCrop.training_augmentation augmentations.py:47-75, .float()) and
 * encoder.py:78 (`obs / 255.`, pass scale = 1/255).  src_kind = 0: `src` is the
 * reference's float NCHW tensor [B][C][Hc][Wc] in [0,255]; src_kind = 2: a float
 * NHWC tensor [B][Hc][Wc][C] in [0,255] (what curla_color_jiggle / curla_noisy_cover
 * write).  C in {3, 6, 9, 12}.  The uint8 ring must be followed by >= 32 readable bytes (the row loader reads
 * whole aligned 16-byte runs) and start on a 4-byte boundary; frames may have any size. */
int curla_conv1_fwd(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                    const float* w, const float* bias, float* out, int B, int C, int Hs, int Ws, int Hc, int Wc,
                    int channels, float scale, void* stream);
/* The same for the first layer, both minibatches gathered from one uint8 ring (src_kind 1 of curla_conv1_fwd). */
int curla_conv1_fwd2(const uint8_t* ring, const int64_t* idx, const int32_t* h1, const int32_t* w1, const float* w,
                     const float* bias, float* out, int B, const int64_t* idx2, const int32_t* h1_2,
                     const int32_t* w1_2, const float* w2, const float* bias2, float* out2, int B2, int C, int Hs, int Ws,
                     int Hc, int Wc, int channels, float scale, void* stream);

/* Layers 2..L: 3x3 stride 1, 32 -> 32, + bias + ReLU (encoder.py:59-63,84-87). */
int curla_conv3x3_s1_fwd(const float* in, const float* w, const float* bias, float* out, int B, int Hi, int Wi,
                         int channels, void* stream);
/* Two forwards of the same geometry in ONE launch, each with its own weights (e.g. a minibatch through the online
 * encoder and another through the target encoder, curl_sac.py:350-358, 408-409): the second problem's items follow the
 * first's in the persistent grid and a workgroup re-builds its weight registers once.  A launch carries 4-9 us of
 * fixed cost at these sizes, so two 512-sample launches are slower than one of 1024. */
int curla_conv3x3_s1_fwd2(const float* in, const float* w, const float* bias, float* out, int B, const float* in2,
                          const float* w2, const float* bias2, float* out2, int B2, int Hi, int Wi, int channels,
                          void* stream);
/* The whole stack of stride-1 layers of one or two minibatches (B2 = 0: one) in ONE launch: layer l maps
 * [B][Hi-2l][Wi-2l][32] (in for l = 0, out[l-1] after) to out[l] with w[l], bias[l].  Every activation is written to HBM
 * as by the single-layer calls.  Needs B and B2 to be multiples of curla_conv3x3_s1_stack_granule() (a workgroup of the
 * persistent grid then owns its samples through all layers and only has to wait for itself); CURLA_ERR_UNSUPPORTED
 * otherwise -- fall back to one curla_conv3x3_s1_fwd2 per layer. */
int curla_conv3x3_s1_fwd_stack(int nlayers, const float* in, const float* const* w, const float* const* bias,
                               float* const* out, int B, const float* in2, const float* const* w2,
                               const float* const* bias2, float* const* out2, int B2, int Hi, int Wi, int channels,
                               void* stream);
/* Batch-size multiple curla_conv3x3_s1_fwd_stack needs: the size of its persistent grid on the current device (one
 * workgroup per CU in the row-walk form, two in the banded one). */
int curla_conv3x3_s1_stack_granule(void);

/* Autograd of the above (what critic_loss.backward() / loss.backward() run,
 * curl_sac.py:366,417).  `g` is the gradient w.r.t. the layer's pre-activation
 * (already ReLU-masked).  dgrad writes the pre-activation gradient of the layer
 * below: conv_transpose(g, w) zeroed where act_below <= 0. */
int curla_conv3x3_s1_dgrad(const float* g, const float* w, const float* act_below, float* gin, int B, int Ho, int Wo,
                           int channels, void* stream);
int curla_conv3x3_s1_wgrad(const float* in, const float* g, float* dw, float* db, float* workspace, int B, int Hi,
                           int Wi, int channels, void* stream);
int curla_conv1_wgrad(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                      const float* g, float* dw, float* db, float* workspace, int B, int C, int Hs, int Ws, int Hc,
                      int Wc, int channels, float scale, void* stream);
/* The weight-gradient kernels alone: every workgroup leaves its partial sums as one slab in `workspace` (*nslabs of
 * them, each 32*cin*9 + 32 floats) and nothing is reduced yet.  A backward pass runs one of these per conv layer, each
 * into its own workspace, and ONE curla_wgrad_reduce_multi at the end sums all layers' slabs (fixed order) into their
 * dW / db -- the per-layer reductions of curla_conv3x3_s1_wgrad / curla_conv1_wgrad would each be a launch of their
 * own.  njobs <= 8; nw[j] = channels*cin*9 of job j, nb[j] = its bias count = channels (NULL: 32 each).
 * FILTER COUNTS OTHER THAN 32 (the reference is generic in num_filters, encoder.py:54-63 / train.py:84): every conv
 * entry point above and below takes `channels` in 4 .. 256 (a multiple of 4); 32 runs the gfx950 row-walk kernels,
 * anything else plain direct convolutions on the vector ALU (csrc/conv_generic.h) -- the same results to fp32 rounding,
 * one to two orders of magnitude slower; their weight-gradient slabs are ONE slab of channels*cin*9 + channels floats.
 * curla_conv3x3_s1_fwd_stack stays 32-only (CURLA_ERR_UNSUPPORTED: one curla_conv3x3_s1_fwd / _fwd2 per layer). */
int curla_conv3x3_s1_wgrad_slabs(const float* in, const float* g, float* workspace, int B, int Hi, int Wi, int channels,
                                 int* nslabs, void* stream);
int curla_conv1_wgrad_slabs(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                            const float* g, float* workspace, int B, int C, int Hs, int Ws, int Hc, int Wc, int channels,
                            float scale, int* nslabs, void* stream);
/* Weight gradient (slabs, as curla_conv3x3_s1_wgrad_slabs) AND data gradient (as curla_conv3x3_s1_dgrad, ReLU mask =
 * the layer's input `in`) of one stride-1 layer in ONE launch: both only read the output gradient g. */
int curla_conv3x3_s1_bwd_slabs(const float* in, const float* g, const float* w, float* gin, float* workspace, int B, int Hi,
                               int Wi, int channels, int* nslabs, void* stream);
int curla_wgrad_reduce_multi(int njobs, const float* const* slabs, const int* nslabs, const int* nw, const int* nb,
                             float* const* dw, float* const* db, void* stream);
/* floats of `workspace` the two wgrad entry points need (per-workgroup partial slabs) */
size_t curla_conv_wgrad_workspace_floats(int cin);

/* ---- dense layers: encoder fc (encoder.py:66,98), actor trunk (curl_sac.py:70-74),
 * twin Q (curl_sac.py:129-139), CURL bilinear logits (curl_sac.py:219-220) ----
 * C[z] = epilogue(alpha * opA(A[z]) * opB(B[z])^T); a_kmajor/b_kmajor = operand is
 * stored [K][rows]; epilogue = +bias[n], ReLU, zero where mask <= 0.  ksplit > 1
 * writes partial products C + s*split_stride (no epilogue; reduce with
 * curla_splitk_reduce or curla_fc_ln_fwd). */
int curla_gemm(const float* A, int a_kmajor, int lda, long long strideA, const float* B, int b_kmajor, int ldb,
               long long strideB, float* C, int ldc, long long strideC, int M, int N, int K, int nbatch, int ksplit,
               long long split_stride, float alpha, const float* bias, long long strideBias, int relu,
               const float* mask, int ldmask, long long strideMask, void* stream);
/* Products with a small output and a long k (ceil(M/32) * ceil(N/32) * nbatch <= 128 tiles, K >= 256, K % 64 == 0, no
 * epilogue, no split) are computed by one workgroup per 16 x 16 tile whose waves split k; 1 if the shape qualifies. */
int curla_gemm_small_shape(int M, int N, int K, int nbatch);
/* Such a product plus colsum[z][m] = sum_k opA(A[z])[m][k] from the same launch -- a layer's weight gradient dy^T x
 * [N_out x N_in] with its bias gradient, the column sums of dy (curl_sac.py:70-74,129-133 backward): the workgroups
 * of the first column of tiles add up the A operands they load anyway.  Needs curla_gemm_small_shape(...);
 * CURLA_ERR_UNSUPPORTED otherwise. */
int curla_gemm_colsum(const float* A, int a_kmajor, int lda, long long strideA, const float* B, int b_kmajor, int ldb,
                      long long strideB, float* C, int ldc, long long strideC, int M, int N, int K, int nbatch,
                      float* colsum, long long strideColsum, void* stream);
/* Backward of a batched linear layer y = x W^T (curl_sac.py:70-74, 129-133: the hidden and first layers of the actor
 * trunk and of the twin Q functions) in ONE launch where the two products take the same kernel family, two otherwise:
 *   dW[z][n][k] = sum_b dy[z][b][n] x[z][b][k]          (db[z][n] = sum_b dy[z][b][n] when db != NULL: needs
 *                                                        curla_gemm_small_shape(N, K, B, nbatch), else UNSUPPORTED)
 *   dx[z][b][k] = sum_n dy[z][b][n] W[z][n][k], zeroed where mask[z][b][k] <= 0 (mask may be NULL)
 * dy [B][N], x / mask / dx [B][K], W / dW [N][K] row-major, batch strides in floats (0 = shared). */
int curla_linear_bwd(const float* dy, long long stride_dy, const float* x, long long stride_x, const float* W,
                     long long stride_W, const float* mask, long long stride_mask, float* dW, long long stride_dW,
                     float* db, long long stride_db, float* dx, long long stride_dx, int B, int N, int K, int nbatch,
                     void* stream);
/* curla_gemm with a two-level batch: item (outer, inner) at base + inner * stride + outer * stride2 (no split-K).  The
 * twin Q functions of the target critic and of the critic are one such batch of four: the twins a block apart inside a
 * flat parameter buffer, the two flat buffers wherever the allocator put them (curl_sac.py:353-358). */
int curla_gemm_nested(const float* A, int a_kmajor, int lda, long long strideA, long long strideA2, const float* B,
                      int b_kmajor, int ldb, long long strideB, long long strideB2, float* C, int ldc, long long strideC,
                      long long strideC2, int M, int N, int K, int nbatch, int nbatch2, float alpha, const float* bias,
                      long long strideBias, long long strideBias2, int relu, const float* mask, int ldmask,
                      long long strideMask, long long strideMask2, void* stream);
/* nprob <= 4 unrelated products of ONE shape in one launch, operands by pointer: C_i (+ split partials) =
 * A_i [M][K] * B_i [N][K]^T, no epilogue -- the fc layers of several encoders on several activation tensors (the
 * critic phase has three, curl_sac.py:350-358).  Each launch less is ~5 us. */
int curla_gemm_multi(int nprob, const float* const* A, const float* const* B, float* const* C, int lda, int ldb, int ldc,
                     int M, int N, int K, int ksplit, long long split_stride, void* stream);
/* The encoder fc layer's forward product as split-K partial sums, for up to four (x, W) pairs of one shape:
 * partial[i][s][b][f] = sum over the k of split s (32-wide slices, nsplit near-equal runs of them) of x[i][b][k] W[i][f][k],
 * split s at partial[i] + s * split_stride.  W [F][K] row-major; x [B][K] row-major (x_blocked = 0) or BLOCKED
 * [B / 16][K / 32][16][32] (x_blocked = 1: element (b, k) at ((b / 16) (K / 32) + k / 32) 512 + (b % 16) 32 + k % 32).
 * 50 <= F <= 64 even, B % 128 == 0, K % 32 == 0; other shapes: CURLA_ERR_UNSUPPORTED (curla_gemm_multi computes the same
 * sums from a row-major x in another order). */
int curla_fc_fwd_multi(int nprob, const float* const* x, const float* const* W, float* const* partial, int B, int F, int K,
                       int nsplit, long long split_stride, int x_blocked, void* stream);
int curla_splitk_reduce(const float* partial, int nsplit, long long split_stride, int M, int N, int ldp, float* C,
                        int ldc, const float* bias, int relu, void* stream);
/* Backward of the encoder fc layer z = fc(h) (encoder.py:98; autograd of the reference's nn.Linear), F <= 64 features,
 * K % 4 == 0 columns, 16-byte aligned matrices; CURLA_ERR_UNSUPPORTED otherwise (use curla_gemm):
 *   curla_fc_dx: dx[b][n] = (mask == NULL || mask[b][n] > 0) * sum_f dz[b][f] W[f][n]   (W [F][K] as stored by
 *                nn.Linear; mask = h, the ReLU output of the last conv layer: its backward fused in)
 *   curla_fc_dw: dW[f][n] = sum_b dz[b][f] x[b][n]
 * One workgroup per 64 columns, operands streamed from HBM/L2 straight into MFMA registers, 256-byte row pieces. */
int curla_fc_dx(const float* dz, const float* W, const float* mask, float* dx, int B, int F, int K, void* stream);
int curla_fc_dw(const float* dz, const float* x, float* dW, int B, int F, int K, void* stream);
/* Both at once (x doubles as the ReLU mask of dx: h = relu(conv) is the fc layer's input): one launch whose first
 * workgroups compute dx and whose rest compute dW. */
int curla_fc_bwd(const float* dz, const float* W, const float* x, float* dx, float* dW, int B, int F, int K,
                 void* stream);
/* curla_fc_bwd / curla_fc_dw that also finish the LayerNorm parameter gradients curla_ln_bwd_partial left as partial sums
 * (dgamma[f], dbeta[f], dbias_in[f] = sum over the nparts partials, in order; dbias_in may be NULL). */
int curla_fc_bwd_ln(const float* dz, const float* W, const float* x, float* dx, float* dW, int B, int F, int K,
                    const float* ln_partial, int nparts, float* dgamma, float* dbeta, float* dbias_in, void* stream);
int curla_fc_dw_ln(const float* dz, const float* x, float* dW, int B, int F, int K, const float* ln_partial, int nparts,
                   float* dgamma, float* dbeta, float* dbias_in, void* stream);
/* curla_fc_bwd_ln that also finishes a second set of partial sums in the same extra workgroup:
 * extra_out[i] = sum over p < extra_parts of extra_partial[p * extra_len + i] (the CURL head's dW partials). */
int curla_fc_bwd_ln2(const float* dz, const float* W, const float* x, float* dx, float* dW, int B, int F, int K,
                     const float* ln_partial, int nparts, float* dgamma, float* dbeta, float* dbias_in,
                     const float* extra_partial, int extra_parts, int extra_len, float* extra_out, void* stream);

/* Last layer of the actor trunk / the Q functions (curl_sac.py:73-74, 132-133): hidden -> N outputs, N <= 16
 * (Q: 1, actor: 2|A|), batched over `nbatch` identically laid-out networks `stride*` floats apart.
 *   fwd: out[z][m][n] = bias[z][n] + sum_k h[z][m][k] W[z][n][k]                       (K % 4 == 0)
 *   bwd: dh[z][m][k] = (h[z][m][k] > 0) * sum_n dy[z][m][n] W[z][n][k]   (ReLU mask of the layer below fused)
 *        dW[z][n][k] = sum_m dy[z][m][n] h[z][m][k]                      (dW may be NULL: data gradient only) */
int curla_mlp_out_fwd(const float* h, long long strideH, const float* W, long long strideW, const float* bias,
                      long long strideBias, float* out, long long strideOut, int M, int N, int K, int nbatch,
                      void* stream);
int curla_mlp_out_fwd_nested(const float* h, long long strideH, long long strideH2, const float* W, long long strideW,
                             long long strideW2, const float* bias, long long strideBias, long long strideBias2,
                             float* out, long long strideOut, long long strideOut2, int M, int N, int K, int nbatch,
                             int nbatch2, void* stream);
int curla_mlp_out_bwd(const float* dy, long long strideDy, const float* h, long long strideH, const float* W,
                      long long strideW, float* dh, long long strideDh, float* dW, long long strideDW, int M, int N,
                      int K, int nbatch, void* stream);
/* ... and (each optional) the bias gradient of this layer, db_out[z][n] = sum_m dy[z][m][n], and of the layer below,
 * db_hidden[z][k] = sum_m dh[z][m][k]; both with batch stride strideDb */
int curla_mlp_out_bwd_bias(const float* dy, long long strideDy, const float* h, long long strideH, const float* W,
                           long long strideW, float* dh, long long strideDh, float* dW, long long strideDW, int M, int N,
                           int K, int nbatch, float* db_out, float* db_hidden, long long strideDb, void* stream);

/* One problem of curla_fc_ln_fwd_multi (fields as curla_fc_ln_fwd's arguments; fc_out, xhat, rstd, xa, act optional;
 * xa without act: only the feature columns of the rows are written). */
typedef struct CurlaFcLnJob {
  const float *partial, *bias, *gamma, *beta;
  float *fc_out, *y, *xhat, *rstd, *xa;
  const float* act;
  int tanh_out;
} CurlaFcLnJob;
/* curla_mlp_out_bwd_bias for the twin Q functions' last layer (one output, two twins `twin_stride` floats apart in q /
 * target_q_twin / dq) with the output gradient computed in place of being read: the launch of the loss kernel that
 * would produce it goes away, its scalar results are written by the same launch.
 *   kind 1 (update_critic, curl_sac.py:350-362, = curla_critic_td_loss): target = reward + not_done * discount *
 *           (min(target_q_twin) - alpha * log_pi); dq = 2 (q - target) / B; scalars[0] = the critic loss; target_q [B]
 *   kind 2 (update_actor_and_alpha, curl_sac.py:378-399, = curla_actor_loss): dq = d(-min(Q1, Q2))/dQ / B;
 *           scalars[0..3] = actor loss, alpha loss, entropy, alpha; *dlog_alpha = d(alpha loss)/d(log_alpha)
 * dq [2][twin_stride] is also written out (bias-gradient fallbacks, tests). */
typedef struct CurlaLossArgs {
  int kind, A;
  long long twin_stride;
  const float *q, *target_q_twin, *log_pi, *reward, *not_done, *log_std;
  const double* log_alpha;
  double* dlog_alpha;
  float *target_q, *scalars, *dq;
  float discount, target_entropy;
} CurlaLossArgs;
int curla_mlp_out_bwd_loss(const CurlaLossArgs* loss, const float* h, long long strideH, const float* W,
                           long long strideW, float* dh, long long strideDh, float* dW, long long strideDW, int B, int K,
                           float* db_out, float* db_hidden, long long strideDb, void* stream);
/* fc split-K reduce + bias + LayerNorm(eps) [+ tanh] (encoder.py:98-107).  Saves
 * xhat / rstd for the backward when non-NULL.  F <= 256.  `xa` (optional, with `act` [B][A]): also writes the Q
 * functions' input rows xa[b] = [ y[b] | act[b] ], i.e. torch.cat([obs, action], dim=1) (curl_sac.py:138). */
int curla_fc_ln_fwd(const float* partial, int nsplit, long long split_stride, int ldp, const float* bias,
                    const float* gamma, const float* beta, int B, int F, float eps, float* fc_out, float* y,
                    float* xhat, float* rstd, int tanh_out, float* xa, const float* act, int A, void* stream);
/* up to 4 such problems of one shape (same B, F, nsplit, strides, eps, A) in one launch: the features of several
 * encoders over their own split-K partials (actor / target critic / critic, curl_sac.py:350-358) */
int curla_fc_ln_fwd_multi(int njobs, const CurlaFcLnJob* jobs, int nsplit, long long split_stride, int ldp, int B,
                          int F, float eps, int A, void* stream);
/* dx, and (when non-NULL) dgamma, dbeta; dbias_in (optional, needs dgamma/dbeta) = column sums of dx = the gradient of
 * the fc bias feeding the LayerNorm */
int curla_ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, int B, int F, float* dx,
                 float* dgamma, float* dbeta, float* dbias_in, void* stream);
/* The same with the incoming gradient the sum of two strided row blocks dy[b*ld_dy + f] + dy2[b*ld_dy + f] (dy2 may
 * be NULL): the two halves of the twin-Q input gradient [2][B][F + A], torch.cat's backward (curl_sac.py:138), read in
 * place. */
int curla_ln_bwd_twin(const float* dy, const float* dy2, int ld_dy, const float* xhat, const float* rstd,
                      const float* gamma, int B, int F, float* dx, float* dgamma, float* dbeta, float* dbias_in,
                      void* stream);
/* curla_ln_bwd_twin without the launch that finishes the parameter gradients: dx as above, and per-workgroup partial sums
 * of dgamma / dbeta / the fc bias gradient into `partial` ([*nparts][3][F] floats, *nparts = ceil(B / 4)).  The fc
 * backward that follows (curla_fc_bwd_ln / curla_fc_dw_ln) adds them up in one extra workgroup of its own launch. */
int curla_ln_bwd_partial(const float* dy, const float* dy2, int ld_dy, const float* xhat, const float* rstd,
                         const float* gamma, int B, int F, float* dx, float* partial, int* nparts, void* stream);
/* out[z][n] = sum_m X[z][m][n] (bias gradients) */
int curla_colsum(const float* X, int M, int N, int ldx, long long strideX, float* out, long long strideOut, int nbatch,
                 void* stream);

/* the three bias gradients of a (batched) 3-layer MLP backward in one launch: out_i[z][n] = sum_m X_i[z][m][n],
 * X_i dense [nbatch][M][N_i], out_i batches `strideOut` floats apart (curl_sac.py:70-74,129-133 backward) */
int curla_colsum3(const float* X0, int N0, const float* X1, int N1, const float* X2, int N2, int M, float* out0,
                  float* out1, float* out2, long long strideOut, int nbatch, void* stream);

/* ---- squashed-Gaussian policy head (curl_sac.py:20-35, 87-108) ----
 * trunk_out [B][2A] = [mu | raw log_std]; `noise` replaces torch.randn_like
 * (curl_sac.py:97); NULL noise = compute_pi False (select_action).  `pi_xa` (optional, with noise): pi is also
 * (or, with pi NULL, only) written at pi_xa[b * xa_ld + a] -- the action columns of the Q functions' input rows
 * [features | action] (curl_sac.py:138, 353). */
int curla_actor_head_fwd(const float* trunk_out, const float* noise, int B, int A, float log_std_min,
                         float log_std_max, float* mu, float* pi, float* log_pi, float* log_std, float* tanh_ls,
                         float* pi_xa, int xa_ld, void* stream);
/* The actor trunk's last layer and the policy head in one launch (curl_sac.py:79-108): trunk_out [B][2A] =
 * h [B][K] @ W [2A][K]^T + bias is written out and the head's outputs follow from it as in curla_actor_head_fwd
 * (2A <= 16, K % 4 == 0; CURLA_ERR_UNSUPPORTED otherwise). */
int curla_mlp_out_head_fwd(const float* h, const float* W, const float* bias, float* trunk_out, int B, int A, int K,
                           const float* noise, float log_std_min, float log_std_max, float* mu, float* pi,
                           float* log_pi, float* log_std, float* tanh_ls, float* pi_xa, int xa_ld, void* stream);
/* The two forms above with the standard-normal noise of `torch.randn_like(mu)` (curl_sac.py:97) drawn INSIDE the launch
 * instead of read: element i = b * A + a is Box-Muller on outputs (i % 4) & 2, + 1 of Philox4x32-10 with key `seed` and
 * counter `offset + i / 4` (cos branch for even i, sin for odd), and is also stored to noise_out [B][A] (the backward
 * pass reads it).  The caller advances `offset` by ceil(B A / 4) per call.  rng_dev != NULL (8-byte aligned device
 * memory): seed = rng_dev[0] and offset = rng_dev[1] are read when the kernel RUNS and the by-value pair is ignored -- a
 * captured hipGraph is replayed with new stream positions (CurlSacAgent.enable_update_graphs). */
int curla_actor_head_fwd_rng(const float* trunk_out, float* noise_out, unsigned long long seed,
                             unsigned long long offset, const unsigned long long* rng_dev, int B, int A,
                             float log_std_min, float log_std_max, float* mu, float* pi, float* log_pi, float* log_std,
                             float* tanh_ls, float* pi_xa, int xa_ld, void* stream);
int curla_mlp_out_head_fwd_rng(const float* h, const float* W, const float* bias, float* trunk_out, int B, int A, int K,
                               float* noise_out, unsigned long long seed, unsigned long long offset,
                               const unsigned long long* rng_dev, float log_std_min, float log_std_max, float* mu,
                               float* pi, float* log_pi, float* log_std, float* tanh_ls, float* pi_xa, int xa_ld,
                               void* stream);
/* gradient w.r.t. trunk_out of sum(gpi*pi) + glp*log_pi; glp = glp_rows[b] or glp_scale*exp(*log_alpha);
 * gpi[b][a] is read at gpi[b*gpi_ld + a] (+ gpi2[b*gpi_ld + a] when gpi2 is not NULL: the action columns of the twin-Q
 * input gradient summed over the twin in place) */
int curla_actor_head_bwd(const float* gpi, const float* gpi2, int gpi_ld, const float* glp_rows, const double* log_alpha,
                         float glp_scale, const float* noise, const float* pi, const float* log_std,
                         const float* tanh_ls, int B, int A, float log_std_min, float log_std_max, float* dtrunk_out,
                         void* stream);

/* torch.cat([z, action], 1) (curl_sac.py:138) and its backward summed over the twin */
int curla_concat(const float* z, const float* act, int B, int F, int A, float* xa, void* stream);
int curla_split_sum(const float* dxa, long long twin_stride, int B, int F, int A, float* dz, float* dact,
                    void* stream);

/* ---- SAC targets and losses ---- */
/* target_Q = r + not_done*discount*(min(tq1,tq2) - exp(log_alpha)*log_pi) (curl_sac.py:353-355) */
int curla_td_target(const float* tq, long long twin_stride, const float* log_pi, const float* reward,
                    const float* not_done, const double* log_alpha, float discount, int B, float* target_q,
                    void* stream);
/* loss = mse(q1,tQ)+mse(q2,tQ) and dq = dloss/dq (curl_sac.py:359) */
int curla_critic_loss(const float* q, long long twin_stride, const float* target_q, int B, float* loss, float* dq,
                      void* stream);
/* the two above in one launch: target_Q from the target critic's `tq`, then the loss and dq of the critic's `q` */
int curla_critic_td_loss(const float* q, const float* tq, long long twin_stride, const float* log_pi,
                         const float* reward, const float* not_done, const double* log_alpha, float discount, int B,
                         float* target_q, float* loss, float* dq, void* stream);
/* actor_loss, alpha_loss, entropy, alpha -> scalars4; dq = d actor_loss/dq; dlog_alpha (float64 like the
 * reference's log_alpha, curl_sac.py:292) (curl_sac.py:378-399) */
int curla_actor_loss(const float* q, long long twin_stride, const float* log_pi, const float* log_std, int A,
                     const double* log_alpha, float target_entropy, int B, float* scalars4, float* dq,
                     double* dlog_alpha, void* stream);
/* CrossEntropyLoss(logits, arange(B)) and d/dlogits (curl_sac.py:221,411-413); `loss` (the mean of row_loss) may be
 * NULL when the scalar is not going to be logged */
int curla_curl_ce(const float* logits, int B, int ld, float* row_loss, float* loss, float* dlogits, void* stream);
/* The CURL head in one launch (curl_sac.py:211-222,406-417 and their autograd), given the anchor features z_a, the
 * positives' z_pos and wz = (W z_pos^T)^T = z_pos W^T, all [B][F]: logits = z_a wz^T (optionally stored), row_loss[a] =
 * logsumexp(logits[a]) - logits[a][a] (their mean into `loss` when it is not NULL), dlogits = (softmax - I) / B
 * (optionally stored), dz = dlogits wz (optionally stored), dfc = the LayerNorm backward of dz with the anchor encoder's
 * (xhat, rstd, gamma), and per block of 16 rows the partial sums of dgamma / dbeta / fc-bias gradient (ln_partial
 * [*nparts][3][F], *nparts = B / 16) and of dW[i][k] = sum_a z_a[a][i] (dlogits z_pos)[a][k] (w_partial [*nparts][F * F]),
 * which curla_fc_bwd_ln2 adds up.  B % 128 == 0, B <= 1024, 49 <= F <= 52; else CURLA_ERR_UNSUPPORTED (the separate
 * launches compute the same quantities). */
int curla_curl_head(const float* z_a, const float* z_pos, const float* wz, const float* xhat, const float* rstd,
                    const float* gamma, int B, int F, float* row_loss, float* loss, float* dfc, float* ln_partial,
                    int* nparts, float* w_partial, float* logits, float* dlogits, float* dz, void* stream);
int curla_mean(const float* x, int n, float* out, void* stream);

/* target <- tau*param + (1-tau)*target over a flat parameter block (utils.py:37-41) */
int curla_soft_update(const float* param, float* target, size_t n, float tau, float one_minus_tau, void* stream);
/* one flat block with two rates: elements [0, split) use tau_a, [split, n) tau_b (encoder_tau | critic_tau,
 * curl_sac.py:442-445) */
int curla_soft_update2(const float* param, float* target, size_t n, size_t split, float tau_a, float one_minus_tau_a,
                       float tau_b, float one_minus_tau_b, void* stream);

/* One torch.optim.Adam step (weight_decay 0, amsgrad off; the reference's five optimizers, curl_sac.py:299-313) over a
 * flat run of n fp32 parameters with their gradient and moment runs: replaces torch's multi-tensor Adam launches
 * (~70 workgroups on a 256-CU chip).  `step` is the 1-based count of this step; the hyper-parameters are doubles as in the
 * optimizer's param_group (the bias corrections and lr/(1-beta1^step) are evaluated in double on the host, like
 * torch's single-tensor Adam, and rounded to float once).
 * dyn != NULL (device memory, 2 floats): the two step-dependent factors -- dyn[0] = (float)(lr / (1 - beta1^step)),
 * dyn[1] = (float)sqrt(1 - beta2^step) -- are read when the kernel RUNS instead of being evaluated from `step`: a captured
 * hipGraph is replayed with new step counts (the host writes the same two floats it would have passed by value). */
int curla_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                    double beta1, double beta2, double eps, long long step, const float* dyn, void* stream);
/* curla_adam_step plus, in the same launch, the soft update of the target copy of the same flat run with the freshly
 * stepped parameters (critic_optimizer.step() ... soft_update_params x3, curl_sac.py:367,442-445): target[i] <- tau p[i]
 * + (1 - tau) target[i], elements [0, split) with (tau_a, one_minus_tau_a), the rest with (tau_b, one_minus_tau_b);
 * the same arithmetic as curla_adam_step followed by curla_soft_update2. */
int curla_adam_step_lerp(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                         double beta1, double beta2, double eps, long long step, const float* dyn, float* target,
                         size_t split, float tau_a, float one_minus_tau_a, float tau_b, float one_minus_tau_b,
                         void* stream);
/* curla_adam_step plus, in the same launch, the Adam step of ONE float64 scalar parameter with its own optimizer state
 * and hyper-parameters (log_alpha, stepped right after the actor: curl_sac.py:393-404), in double.  dyn64 (device memory,
 * 2 doubles, or NULL): the scalar's two step-dependent factors, as `dyn` for the flat run. */
int curla_adam_step_scalar64(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                             double beta1, double beta2, double eps, long long step, const float* dyn, double* param64,
                             const double* grad64, double* exp_avg64, double* exp_avg_sq64, double lr64, double beta1_64,
                             double beta2_64, double eps64, long long step64, const double* dyn64, void* stream);
/* A float64 scalar riding in a float32 all-reduce bucket (data parallel: log_alpha's gradient inside the actor's bucket
 * instead of an 8-byte collective of its own; the reference has no counterpart -- its one float64
 * parameter is curl_sac.py:284-286, stepped at :404).  curla_f64_pack writes *value as CURLA_F64_WORDS float32 words:
 * signed 20-bit fixed-point digits, most significant first, digit j worth 2^(8 - 20 j); exact for |value| < 2^28 (larger
 * magnitudes, inf and nan travel in word 0 alone, inexactly).  Sums of the words over <= 16 ranks are exact in float32.
 * curla_f64_unpack: *value = RN(sum of (words[j] * n_mul) * 2^(8 - 20 j)) / n_div with ONE rounding of the exact sum --
 * n_mul undoes an average taken by the collective (ncclAvg: n_mul = world size), n_div = world size gives the mean. */
#define CURLA_F64_WORDS 8
int curla_f64_pack(const double* value, float* words, void* stream);
int curla_f64_unpack(const float* words, double n_mul, double n_div, double* value, void* stream);
/* Two such steps of two optimizers on the same parameters with the same gradient, one after the other, in one pass
 * (encoder_optimizer.step(); cpc_optimizer.step(), curl_sac.py:418-423).  exp_avg2 / exp_avg_sq2 cover n elements, the
 * first n_pre of which (CURL.W) take the second step only; exp_avg1 / exp_avg_sq1 cover the remaining n - n_pre.
 * dyn (device memory, 4 floats, or NULL): (lr1 / (1 - beta1_1^step1), sqrt(1 - beta2_1^step1)) then the same for the
 * second optimizer, as in curla_adam_step. */
int curla_adam_step2(float* param, const float* grad, float* exp_avg1, float* exp_avg_sq1, float* exp_avg2,
                     float* exp_avg_sq2, size_t n, size_t n_pre, double lr1, double beta1_1, double beta2_1, double eps1,
                     long long step1, double lr2, double beta1_2, double beta2_2, double eps2, long long step2,
                     const float* dyn, void* stream);

/* ---- augmentations that produce float observations (augmentations.py:78-205; kornia arithmetic is not
 * vendored by the reference: PARITY UNPINNED, the algorithm is this build's statement of kornia's documented
 * behaviour, see oracle/curla_oracle.py color_jiggle / noisy_cover).  Output: float NHWC [B][H][W][C] in [0,255].
 * color_jiggle: params[B*C/3][4] = (apply, contrast, saturation, hue_radians) per RGB frame, order[4] = permutation
 * of {0 brightness (identity), 1 contrast, 2 saturation, 3 hue}.  noisy_cover: rows [0,top) and [H-bottom,H) painted
 * with (c0,c1,c2) per RGB channel, + noise (float NHWC), clamped to [0,255].  gather_nhwc: identity (u8 -> float). */
int curla_color_jiggle(const uint8_t* frames, const int64_t* idx, const float* params, const int32_t* order, int B,
                       int C, int H, int W, float* out, void* stream);
int curla_noisy_cover(const uint8_t* frames, const int64_t* idx, const float* noise, float c0, float c1, float c2,
                      int top, int bottom, int B, int C, int H, int W, float* out, void* stream);
int curla_gather_nhwc(const uint8_t* frames, const int64_t* idx, int B, int C, int H, int W, float* out, void* stream);
/* The two augmentations on the reference's own tensor contract -- what ColorJiggle.training_augmentation(image_batch)
 * and NoisyCover.training_augmentation(image_batch) take and return (augmentations.py:105-136,170-205): float NCHW
 * [B][C][H][W] in [0,255], same arithmetic as the ring forms above; `out` may alias `in`. */
int curla_color_jiggle_nchw(const float* in, const float* params, const int32_t* order, int B, int C, int H, int W,
                            float* out, void* stream);
int curla_noisy_cover_nchw(const float* in, const float* noise, float c0, float c1, float c2, int top, int bottom,
                           int B, int C, int H, int W, float* out, void* stream);

/* ---- replay ring helpers ---- */
/* float/uint8 NCHW crops exactly as sample_cpc returns them (utils.py:151-166) */
int curla_crop_nchw(const uint8_t* frames, const int64_t* idx, const int32_t* h1, const int32_t* w1, int B, int C,
                    int Hs, int Ws, int Hc, int Wc, float* out_f32, uint8_t* out_u8, void* stream);
/* actions / rewards / not_dones of the sampled transitions (utils.py:159-166): rows idx[b] of the ring's
 * [capacity][A+2] scalar block (action | reward | not_done) into three dense outputs */
int curla_gather_transition_scalars(const float* scalars, const int64_t* idx, int B, int A, float* action, float* reward,
                                    float* not_done, void* stream);
/* The same, fed from pinned host memory: `host_block` (device-visible address of a pinned buffer, see
 * curla_host_device_pointer) holds the minibatch's index block -- B int64 ring rows first, then whatever else the
 * caller lays out (crop offsets, utils.py:151-156), nbytes in all, a multiple of 8.  The kernel reads it over PCIe,
 * writes it to `device_block` and gathers the scalar rows: the only host->device traffic of an update, without a
 * copy-engine transfer in the stream.  The host buffer must stay untouched until the launch has executed. */
int curla_sample_stage(const void* host_block, void* device_block, long long nbytes, const float* scalars, int B, int A,
                       float* action, float* reward, float* not_done, void* stream);
/* device-visible address of a pinned (hipHostMalloc'd / registered) host pointer; CURLA_ERR_ARG if it is not */
int curla_host_device_pointer(void* host, void** device);
/* ReplayBuffer.add: one CHW uint8 observation into ring slot `slot` (utils.py:120-128) */
int curla_store_frame(const uint8_t* chw, uint8_t* frames, long long slot, int C, int H, int W, void* stream);
/* De-duplicated frame store (next_obs[t] shares k-1 of its k frames with obs[t], and equals obs[t+1]
 * inside an episode, utils.py:238-268): `store` holds every RGB frame once, uint8 [F][H][W][3]; row idx[b] (NULL: b) of
 * `fid` (int32, `fid_stride` entries per row) lists the K frame ids of a stack.  Writes the minibatch of stacks
 * out[b][y][x][3f+c] = store[fid[idx[b]][f]][y][x][c] -- uint8 [B][H][W][3K], what curla_conv1_fwd reads (src_kind 1,
 * idx NULL); `out` needs the same 32 bytes of slack as a ring. */
int curla_gather_stacks(const uint8_t* store, const int32_t* fid, int fid_stride, const int64_t* idx, int B, int K,
                        int H, int W, uint8_t* out, void* stream);
int curla_nhwc_to_nchw(const float* in, float* out, int B, int H, int W, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CURLA_HIP_H */
