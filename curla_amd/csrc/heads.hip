// Small fused kernels around the GEMMs: LayerNorm (encoder.py:67,101),
// squashed-Gaussian policy head (curl_sac.py:20-35,87-108), SAC targets and
// losses (curl_sac.py:353-359,378-399), CURL cross-entropy (curl_sac.py:411-413),
// target soft update (utils.py:37-41), and the crop materialiser for callers
// that want the reference's float NCHW minibatch tensors (utils.py:151-166).
// All reductions are fixed-order (bitwise reproducible run to run).
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// block-wide fixed-order sum of one value per thread (256 threads); result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* sm) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// ---- fc split-K reduce + bias + LayerNorm (one wave per row; a lane holds features lane, lane+64, ...:
// NF = ceil(F / 64) rounded up to 1, 2, 3, 4, 8 or 16, i.e. F <= 1024: the reference's --encoder_feature_dim is free,
// train.py:80; 50 is what every configuration uses) ----
// (blockIdx.y = job: up to kMaxLnJobs encoders' features in one launch -- the actor's, the target critic's and the
// critic's at the top of update_critic, curl_sac.py:350-358)
constexpr int kMaxLnJobs = 4;
struct FcLnJobs {
  CurlaFcLnJob j[kMaxLnJobs];
};

template <int NF>
__global__ void fc_ln_fwd_kernel(FcLnJobs jobs, int nsplit, long long sSplit, int ldp, int B, int F, float eps, int A) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B) return;
  const CurlaFcLnJob& jb = jobs.j[blockIdx.y];
  const float *P = jb.partial, *bias = jb.bias, *gamma = jb.gamma, *beta = jb.beta, *act = jb.act;
  float *fc_out = jb.fc_out, *y = jb.y, *xhat = jb.xhat, *rstd = jb.rstd, *xa = jb.xa;
  const int tanh_out = jb.tanh_out;
  float v[NF];
  float tot = 0.f;
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int f = lane + 64 * j;
    v[j] = 0.f;
    if (f < F) {
      // same summation order as a plain loop, but 8 partials are fetched together (the loop is pure load latency)
      const float* p = P + (size_t)row * ldp + f;
      float a = 0.f;
      int s = 0;
      for (; s + 8 <= nsplit; s += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = p[(s + k) * sSplit];
#pragma unroll
        for (int k = 0; k < 8; ++k) a += t[k];
      }
      for (; s < nsplit; ++s) a += p[s * sSplit];
      v[j] = a + bias[f];
    }
    tot += v[j];
  }
  const float mean = wave_sum(tot) / F;
  float d[NF], sq = 0.f;
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    d[j] = lane + 64 * j < F ? v[j] - mean : 0.f;
    sq += d[j] * d[j];
  }
  const float var = wave_sum(sq) / F;
  const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int f = lane + 64 * j;
    if (f < F) {
      const float xh = d[j] * rs;
      float o = xh * gamma[f] + beta[f];
      if (tanh_out) o = tanhf(o);
      if (fc_out) fc_out[(size_t)row * F + f] = v[j];
      y[(size_t)row * F + f] = o;
      if (xhat) xhat[(size_t)row * F + f] = xh;
      if (xa) xa[(size_t)row * (F + A) + f] = o;  // torch.cat([z, action], 1) written in place (curl_sac.py:138)
    }
  }
  if (xa && act && lane < A) xa[(size_t)row * (F + A) + F + lane] = act[(size_t)row * A + lane];
  if (lane == 0 && rstd) rstd[row] = rs;
}

// dx = rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat)).  The incoming gradient is dy[b*ld + f]
// (+ dy2[b*ld + f] when dy2 is given: the two halves of the twin-Q input gradient, torch.cat's backward at
// curl_sac.py:138, summed here instead of in a pass of their own)
// `partial` (optional) [gridDim.x][3][F]: the workgroup's four rows' contributions to dgamma (dy xhat), dbeta (dy) and
// the fc bias gradient (dx), added in wave (= row) order -- the column sums over the whole batch are finished by one
// extra workgroup of the NEXT launch that walks these rows anyway, the fc backward (curla_fc_bwd_ln / curla_fc_dw_ln),
// instead of by a launch of their own (ln_param_grad_kernel).
template <int NF>
__global__ void ln_bwd_kernel(const float* dy, const float* dy2, int ld, const float* xhat, const float* rstd,
                              const float* gamma, int B, int F, float* dx, float* partial) {
  __shared__ float part[4][3][NF * 64];
  const int wave = threadIdx.x >> 6;
  const int row = blockIdx.x * (blockDim.x >> 6) + wave;
  const int lane = threadIdx.x & 63;
  const bool live = row < B;
  if (!live && !partial) return;
  float g[NF], xh[NF], d[NF], s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int f = lane + 64 * j;
    g[j] = 0.f, xh[j] = 0.f, d[j] = 0.f;
    if (live && f < F) {
      d[j] = dy[(size_t)row * ld + f];
      if (dy2) d[j] += dy2[(size_t)row * ld + f];
      g[j] = d[j] * gamma[f];
      xh[j] = xhat[(size_t)row * F + f];
    }
    s1 += g[j];
    s2 += g[j] * xh[j];
  }
  const float m1 = wave_sum(s1) / F;
  const float m2 = wave_sum(s2) / F;
  const float rs = live ? rstd[row] : 0.f;
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int f = lane + 64 * j;
    const float o = rs * (g[j] - m1 - xh[j] * m2);
    if (live && f < F) dx[(size_t)row * F + f] = o;
    if (partial) part[wave][0][f] = d[j] * xh[j], part[wave][1][f] = d[j], part[wave][2][f] = live ? o : 0.f;
  }
  if (partial) {
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * F; i += blockDim.x) {
      const int q = i / F, f = i - q * F;
      partial[(size_t)blockIdx.x * 3 * F + i] = ((part[0][q][f] + part[1][q][f]) + part[2][q][f]) + part[3][q][f];
    }
  }
}

// dgamma[f] = sum_b dy*xhat ; dbeta[f] = sum_b dy ; optionally dbias[f] = sum_b dx (the gradient of the fc bias
// that feeds the LayerNorm: the same walk over the rows).  One block of 1024 per 16 features: 64 row parts, so a
// thread's rows (8 at B = 512) are all in flight at once and F = 50 is four workgroups instead of one -- the kernel
// is pure load latency.  The parts are added in a fixed order: the 4 of a wave by lane exchange, the 16 waves in
// wave order.
__global__ __launch_bounds__(1024) void ln_param_grad_kernel(const float* dy, const float* dy2, int ld, const float* xhat,
                                                             const float* dx, int B, int F, float* dgamma,
                                                             float* dbeta, float* dbias) {
  __shared__ float sg[16][16], sb[16][16], sx[16][16];
  const int fl = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int f = blockIdx.x * 16 + fl;
  float ag = 0.f, ab = 0.f, ax = 0.f;
  if (f < F) {
    int b = part;
    for (; b + 7 * 64 < B; b += 8 * 64) {  // 8 rows in flight, accumulated in row order
      float d[8], x[8], e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        d[k] = dy[(size_t)(b + 64 * k) * ld + f], x[k] = xhat[(size_t)(b + 64 * k) * F + f];
        if (dy2) d[k] += dy2[(size_t)(b + 64 * k) * ld + f];
        e[k] = dbias ? dx[(size_t)(b + 64 * k) * F + f] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) ag += d[k] * x[k], ab += d[k], ax += e[k];
    }
    for (; b < B; b += 64) {
      float d = dy[(size_t)b * ld + f];
      if (dy2) d += dy2[(size_t)b * ld + f];
      ag += d * xhat[(size_t)b * F + f];
      ab += d;
      if (dbias) ax += dx[(size_t)b * F + f];
    }
  }
  // lanes l, l^16, l^32, l^48 hold the same feature: (p0 + p1) + (p2 + p3) in every lane
  ag += __shfl_xor(ag, 16), ab += __shfl_xor(ab, 16), ax += __shfl_xor(ax, 16);
  ag += __shfl_xor(ag, 32), ab += __shfl_xor(ab, 32), ax += __shfl_xor(ax, 32);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 16) sg[wave][fl] = ag, sb[wave][fl] = ab, sx[wave][fl] = ax;
  __syncthreads();
  if (threadIdx.x < 16 && f < F) {
    float g = sg[0][fl], bsum = sb[0][fl], xs = sx[0][fl];
#pragma unroll
    for (int k = 1; k < 16; ++k) g += sg[k][fl], bsum += sb[k][fl], xs += sx[k][fl];
    dgamma[f] = g;
    dbeta[f] = bsum;
    if (dbias) dbias[f] = xs;
  }
}

// out[z][n] = sum_m X[z][m][n]  (bias gradients); 32 columns x 32 row-parts per block, fixed order
__global__ __launch_bounds__(1024) void colsum_kernel(const float* X, int M, int N, int ldx, long long sX, float* out,
                                                      long long sOut) {
  __shared__ float sm[32][33];
  const int c = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  const float* x = X + blockIdx.y * sX;
  float a = 0.f;
  if (n < N) {
    int m = part;
    for (; m + 7 * 32 < M; m += 8 * 32) {  // 8 rows in flight, accumulated in row order
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = x[(size_t)(m + 32 * k) * ldx + n];
#pragma unroll
      for (int k = 0; k < 8; ++k) a += t[k];
    }
    for (; m < M; m += 32) a += x[(size_t)m * ldx + n];
  }
  sm[part][c] = a;
  __syncthreads();
  if (part == 0 && n < N) {
    float t = sm[0][c];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += sm[k][c];
    out[blockIdx.y * sOut + n] = t;
  }
}

// three column sums in one launch (the three bias gradients of an MLP backward): blockIdx.z picks the matrix
struct Colsum3Args {
  const float* X[3];
  float* out[3];
  int N[3];
  long long sX[3];
};
__global__ __launch_bounds__(1024) void colsum3_kernel(Colsum3Args a, int M, long long sOut) {
  __shared__ float sm[32][33];
  const int z = blockIdx.z;
  const int N = a.N[z];
  const int c = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  if (blockIdx.x * 32 >= N) return;  // block-uniform
  const float* x = a.X[z] + blockIdx.y * a.sX[z];
  float acc = 0.f;
  if (n < N) {
    int m = part;
    for (; m + 7 * 32 < M; m += 8 * 32) {  // 8 rows in flight, accumulated in row order
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = x[(size_t)(m + 32 * k) * N + n];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += t[k];
    }
    for (; m < M; m += 32) acc += x[(size_t)m * N + n];
  }
  sm[part][c] = acc;
  __syncthreads();
  if (part == 0 && n < N) {
    float t = sm[0][c];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += sm[k][c];
    a.out[z][blockIdx.y * sOut + n] = t;
  }
}

// ---- last layer of the actor / Q trunks: hidden -> N outputs with N tiny (Q: 1, actor: 2|A|) ----
// As GEMMs these are 16-32 workgroups walking K = 1024 behind two barriers per step (15-22 us each, pure latency);
// here the forward is one wave per row and the backward one pass over the activations that produces both the
// data gradient (ReLU mask of the layer below fused) and the weight gradient.
constexpr int kMaxOut = 16;

// ---- policy head (curl_sac.py:20-35, 87-108), one row: trunk output [mu | raw log_std] -> tanh(mu), pi, log_pi, ... ----
constexpr int kMaxA = 8;
constexpr float kHalfLog2Pi = 0.9189385332046727f;
struct HeadArgs {
  const float* noise;
  float *mu_t, *pi_t, *log_pi, *log_std, *tanh_ls, *pi_xa;  // pi_xa: pi also as the action columns of the Q input rows
  int A, xa_ld;
  float lo, hi;
  // noise_gen != nullptr (noise == nullptr then): the standard-normal draws of `torch.randn_like(mu)` (curl_sac.py:97)
  // are produced HERE -- element i from the Philox4x32-10 stream (rng_seed, counter rng_offset + i / 4, output i % 4),
  // Box-Muller -- and stored to noise_gen[i] for the backward pass, instead of by a launch of their own
  float* noise_gen;
  unsigned long long rng_seed, rng_offset;
  // rng_dev != nullptr: (seed, offset) are read from device memory when the kernel RUNS (rng_dev[0], rng_dev[1]) -- a
  // captured graph is replayed with new stream positions
  const unsigned long long* rng_dev;
};

__device__ __forceinline__ void philox4x32_10(unsigned long long seed, unsigned long long ctr, unsigned (&out)[4]) {
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1, c3 = (unsigned)p0, c0 = n0, c2 = n2;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

// standard normal number i of the stream: Box-Muller on two of the four 32-bit outputs of counter i / 4
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned long long offset, unsigned i) {
  unsigned r[4];
  philox4x32_10(seed, offset + (i >> 2), r);
  const unsigned a = r[(i & 2)], b = r[(i & 2) + 1];
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);          // [0, 1)
  const float rad = sqrtf(-2.0f * logf(u1)), ang = 6.283185307179586f * u2;
  return (i & 1) ? rad * sinf(ang) : rad * cosf(ang);
}

__device__ __forceinline__ void actor_head_one(float mu, float raw, int b, int a, const HeadArgs& hd, float& lp,
                                               float& corr) {
  const int A = hd.A;
  const float t = tanhf(raw);
  const float ls = hd.lo + 0.5f * (hd.hi - hd.lo) * (t + 1.f);
  if (hd.mu_t) hd.mu_t[(size_t)b * A + a] = tanhf(mu);
  if (hd.log_std) hd.log_std[(size_t)b * A + a] = ls;
  if (hd.tanh_ls) hd.tanh_ls[(size_t)b * A + a] = t;
  if (hd.noise || hd.noise_gen) {
    float n;
    if (hd.noise_gen) {
      const unsigned long long seed = hd.rng_dev ? hd.rng_dev[0] : hd.rng_seed;
      const unsigned long long offs = hd.rng_dev ? hd.rng_dev[1] : hd.rng_offset;
      n = philox_normal(seed, offs, (unsigned)(b * A + a));
      hd.noise_gen[(size_t)b * A + a] = n;
    } else {
      n = hd.noise[(size_t)b * A + a];
    }
    const float p = tanhf(mu + n * expf(ls));
    if (hd.pi_t) hd.pi_t[(size_t)b * A + a] = p;
    if (hd.pi_xa) hd.pi_xa[(size_t)b * hd.xa_ld + a] = p;
    lp += -0.5f * n * n - ls;
    corr += logf(fmaxf(1.f - p * p, 0.f) + 1e-6f);
  }
}

// one thread walks the row's actions
__device__ __forceinline__ void actor_head_row(const float* out2a_row, int b, const HeadArgs& hd) {
  float lp = 0.f, corr = 0.f;
  for (int a = 0; a < hd.A; ++a) actor_head_one(out2a_row[a], out2a_row[hd.A + a], b, a, hd, lp, corr);
  if ((hd.noise || hd.noise_gen) && hd.log_pi) hd.log_pi[b] = lp - kHalfLog2Pi * hd.A - corr;
}

// out[z][m][n] = bias[z][n] + sum_k h[z][m][k] * W[z][n][k]        (curl_sac.py:73-74,132-133 forward)
// (two-level batch: blockIdx.y = outer * nb_inner + inner; the outer level's strides end in 2)
// HEAD: the outputs are the actor trunk's [mu | raw log_std] and the wave that produced a row also runs the policy
// head on it (a launch of its own otherwise: a few hundred cycles of work per row behind a 5 us launch)
// R rows per wave: a weight float4 is loaded once for R rows; a row's sums do not depend on R (R = 1 is what runs).
// LDSW: the block first copies the N weight rows into LDS and its four waves read them from there -- with several
// rows (the actor's 8 x 1024) every wave otherwise pulls 32 KB through the L2 for itself, 16 MB per launch.
template <bool HEAD, int R, bool LDSW>
__global__ __launch_bounds__(256) void mlp_out_fwd_kernel(const float* h, long long sH, long long sH2, const float* W,
                                                          long long sW, long long sW2, const float* bias, long long sB,
                                                          long long sB2, float* out, long long sOut, long long sOut2,
                                                          int M, int N, int K, int nb_inner, HeadArgs hd) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const int zo = blockIdx.y / nb_inner, zi = blockIdx.y - zo * nb_inner;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R, lane = threadIdx.x & 63;
  const float* hz = h + zi * sH + zo * sH2;
  const float* Wz = W + zi * sW + zo * sW2;
  if (LDSW) {
    const int n4 = (N * K) >> 2;
    for (int i = threadIdx.x; i < n4; i += 256)
      reinterpret_cast<f32x4*>(wlds)[i] = reinterpret_cast<const f32x4*>(Wz)[i];
    __syncthreads();
    Wz = wlds;
  }
  if (row0 >= M) return;
  float acc[R][kMaxOut];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < kMaxOut; ++n) acc[r][n] = 0.f;
  for (int k = 4 * lane; k < K; k += 256) {  // (K % 4 == 0 checked by the host)
    f32x4 hv[R];
#pragma unroll
    for (int r = 0; r < R; ++r)  // (rows past the end re-read the last one; never stored)
      hv[r] = *reinterpret_cast<const f32x4*>(hz + (size_t)min(row0 + r, M - 1) * K + k);
#pragma unroll
    for (int n = 0; n < kMaxOut; ++n)
      if (n < N) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(Wz + (size_t)n * K + k);
#pragma unroll
        for (int r = 0; r < R; ++r)
          acc[r][n] += hv[r][0] * wv[0] + hv[r][1] * wv[1] + hv[r][2] * wv[2] + hv[r][3] * wv[3];
      }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = row0 + r;
    if (row >= M) break;  // wave-uniform
#pragma unroll
    for (int n = 0; n < kMaxOut; ++n)
      if (n < N) {
        const float s = wave_sum(acc[r][n]);
        acc[r][n] = s;
        if (lane == 0)
          out[zi * sOut + zo * sOut2 + (size_t)row * N + n] = s + (bias ? bias[zi * sB + zo * sB2 + n] : 0.f);
      }
    if (HEAD) {
      // every lane holds the row's outputs: lane a < A takes action a, the two sums over the actions are then added
      // in action order by lane 0 (the same order as the one-thread walk)
      float mu = 0.f, raw = 0.f;
#pragma unroll
      for (int n = 0; n < kMaxOut; ++n) {
        const float v = n < N ? acc[r][n] + (bias ? bias[n] : 0.f) : 0.f;
        if (n == lane) mu = v;
        if (n == lane + hd.A) raw = v;
      }
      float lp = 0.f, corr = 0.f;
      if (lane < hd.A) actor_head_one(mu, raw, row, lane, hd, lp, corr);
      float lps = 0.f, cs = 0.f;
      for (int a = 0; a < hd.A; ++a) lps += __shfl(lp, a), cs += __shfl(corr, a);
      if (lane == 0 && (hd.noise || hd.noise_gen) && hd.log_pi) hd.log_pi[row] = lps - kHalfLog2Pi * hd.A - cs;
    }
  }
}

// dh[z][m][k] = (h[z][m][k] > 0) * sum_n dy[z][m][n] * W[z][n][k];   dW[z][n][k] = sum_m dy[z][m][n] * h[z][m][k]
// One block = 16 k-columns x 64 row parts (K/16 x nbatch blocks: 128 for the twin Q functions); a thread's rows are
// all in flight at once; the 64 partial dW sums are added in part order (fixed order).
constexpr int kOutRows = 8;  // rows per thread per pass (64 parts x 8 = 512 rows per pass)
// db_out[z][n] = sum_m dy[z][m][n] and db_h[z][k] = sum_m dh[z][m][k] (optional): the bias gradients of this layer and
// of the one below -- the block already walks every row of its 16 columns.
// GEN: where dy comes from.  0: memory.  1 / 2: the twin Q functions' output gradient is a per-row formula of a few
// scalars -- the TD error of update_critic (curl_sac.py:350-362) or d(-min(Q1, Q2))/dQ of the actor loss
// (curl_sac.py:378-383) -- so every block evaluates it for its rows instead of waiting for a loss kernel's launch,
// and block (0, 0) also adds up that kernel's scalars (loss values, d(alpha loss)/d(log_alpha)).  N == 1, z = twin.
struct LossGen {
  const float *q, *tq, *log_pi, *reward, *not_done, *log_std;
  const double* log_alpha;
  double* dlog_alpha;
  float *target_q, *scalars, *dq;
  long long sTwin;
  float discount, target_entropy;
  int A;
};

// block-wide fixed-order sum of one value per thread (1024 threads); result valid in every thread
__device__ __forceinline__ float block_sum_1024(float v, float* sm16) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm16[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = sm16[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) t += sm16[w];
  return t;
}

template <int GEN>
__global__ __launch_bounds__(1024) void mlp_out_bwd_kernel(const float* dy, long long sDy, const float* h,
                                                           long long sH, const float* W, long long sW, float* dh,
                                                           long long sDh, float* dW, long long sDW, int M, int N,
                                                           int K, float* db_out, float* db_h, long long sDb, LossGen lg) {
  __shared__ float sm[64][17];
  const int z = blockIdx.y;
  const float alpha = GEN ? (float)exp(*lg.log_alpha) : 0.f;
  const float inv = GEN ? 1.f / M : 0.f;
  auto gen = [&](int m, float& target) -> float {  // dy[z][m] (and, GEN 1, the TD target of row m)
    if (GEN == 1) {
      const float v = fminf(lg.tq[m], lg.tq[lg.sTwin + m]) - alpha * lg.log_pi[m];
      target = lg.reward[m] + lg.not_done[m] * lg.discount * v;
      return 2.f * (lg.q[z * lg.sTwin + m] - target) * inv;
    }
    const float q1 = lg.q[m], q2 = lg.q[lg.sTwin + m];
    target = 0.f;
    // d(-min)/dq: the smaller one takes -1/B; an exact tie splits it (torch.min backward)
    if (z == 0) return q1 < q2 ? -inv : (q1 == q2 ? -0.5f * inv : 0.f);
    return q2 < q1 ? -inv : (q1 == q2 ? -0.5f * inv : 0.f);
  };
  const int kl = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + kl;
  const bool kv = k < K;
  const float* dyz = dy + z * sDy;
  const float* hz = h + z * sH;
  float* dhz = dh + z * sDh;
  float wreg[kMaxOut], acc[kMaxOut];
#pragma unroll
  for (int n = 0; n < kMaxOut; ++n) {
    wreg[n] = (n < N && kv) ? W[z * sW + (size_t)n * K + k] : 0.f;
    acc[n] = 0.f;
  }
  float bh = 0.f, bo = 0.f;  // column sums of dh (column k) and, in block 0, of dy (column kl)
  const bool do_bo = db_out && blockIdx.x == 0 && kl < N;
  for (int m0 = part; m0 < M; m0 += 64 * kOutRows) {
    float hv[kOutRows];
#pragma unroll
    for (int u = 0; u < kOutRows; ++u) {
      const int m = m0 + 64 * u;
      hv[u] = (m < M && kv) ? hz[(size_t)m * K + k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kOutRows; ++u) {
      const int m = m0 + 64 * u;
      if (m < M) {
        float s = 0.f;
        if (GEN) {
          float target;
          const float d = gen(m, target);
          s = d * wreg[0];
          acc[0] += d * hv[u];
          if (do_bo) bo += d;
          if (blockIdx.x == 0 && kl == 0) {  // one block per twin also leaves dy (and the target) in memory
            lg.dq[z * lg.sTwin + m] = d;
            if (GEN == 1 && z == 0) lg.target_q[m] = target;
          }
        } else {
#pragma unroll
          for (int n = 0; n < kMaxOut; ++n)
            if (n < N) {
              const float d = dyz[(size_t)m * N + n];
              s += d * wreg[n];
              acc[n] += d * hv[u];
            }
          if (do_bo) bo += dyz[(size_t)m * N + kl];
        }
        const float dv = hv[u] > 0.f ? s : 0.f;
        if (kv) dhz[(size_t)m * K + k] = dv;
        bh += dv;
      }
    }
  }
  // fixed-order sums over the 64 row parts: dW's first row and the two bias sums share one barrier pair (three thread
  // rows add one array each), dW's remaining rows follow one by one
  __shared__ float sm_h[64][17], sm_o[64][17];
  const bool want_o = db_out && blockIdx.x == 0;  // block-uniform
  __syncthreads();
  if (dW) sm[part][kl] = acc[0];
  if (db_h) sm_h[part][kl] = bh;
  if (want_o) sm_o[part][kl] = bo;
  __syncthreads();
  if (part < 3) {
    const float(*src)[17] = part == 0 ? sm : (part == 1 ? sm_h : sm_o);
    const bool on = part == 0 ? (dW != nullptr && kv) : (part == 1 ? (db_h != nullptr && kv) : (want_o && kl < N));
    if (on) {
      float t = src[0][kl];
#pragma unroll 8
      for (int p = 1; p < 64; ++p) t += src[p][kl];
      if (part == 0)
        dW[z * sDW + k] = t;
      else if (part == 1)
        db_h[z * sDb + k] = t;
      else
        db_out[z * sDb + kl] = t;
    }
  }
  if (GEN && blockIdx.x == 0 && z == 0) {  // the loss kernel's scalars (block-uniform branch)
    __shared__ float sm16[16];
    if (GEN == 1) {
      float a1 = 0.f, a2 = 0.f;
      for (int b = threadIdx.x; b < M; b += 1024) {
        const float v = fminf(lg.tq[b], lg.tq[lg.sTwin + b]) - alpha * lg.log_pi[b];
        const float t = lg.reward[b] + lg.not_done[b] * lg.discount * v;
        const float d1 = lg.q[b] - t, d2 = lg.q[lg.sTwin + b] - t;
        a1 += d1 * d1, a2 += d2 * d2;
      }
      const float s1 = block_sum_1024(a1, sm16);
      const float s2 = block_sum_1024(a2, sm16);
      if (threadIdx.x == 0) lg.scalars[0] = s1 * inv + s2 * inv;
    } else {
      float al = 0.f, hl = 0.f, en = 0.f;
      for (int b = threadIdx.x; b < M; b += 1024) {
        al += alpha * lg.log_pi[b] - fminf(lg.q[b], lg.q[lg.sTwin + b]);
        hl += -lg.log_pi[b] - lg.target_entropy;
        float e = 0.f;
        for (int a = 0; a < lg.A; ++a) e += lg.log_std[(size_t)b * lg.A + a];
        en += 0.5f * lg.A * (1.0f + 1.8378770664093453f) + e;
      }
      const float s_al = block_sum_1024(al, sm16);
      const float s_hl = block_sum_1024(hl, sm16);
      const float s_en = block_sum_1024(en, sm16);
      if (threadIdx.x == 0) {
        lg.scalars[0] = s_al * inv;
        lg.scalars[1] = alpha * (s_hl * inv);
        lg.scalars[2] = s_en * inv;
        lg.scalars[3] = alpha;
        if (lg.dlog_alpha) *lg.dlog_alpha = (double)(alpha * (s_hl * inv));  // d/dlog_alpha exp(log_alpha)*c = alpha*c
      }
    }
  }
  if (dW == nullptr) return;  // block-uniform
#pragma unroll
  for (int n = 1; n < kMaxOut; ++n)
    if (n < N) {  // block-uniform
      __syncthreads();
      sm[part][kl] = acc[n];
      __syncthreads();
      if (part == 0 && kv) {
        float t = sm[0][kl];
#pragma unroll 8
        for (int p = 1; p < 64; ++p) t += sm[p][kl];
        dW[z * sDW + (size_t)n * K + k] = t;
      }
    }
}

// ---- policy head, a launch of its own (acting path; training fuses it into mlp_out_fwd_kernel<true>) ----
__global__ void actor_head_fwd_kernel(const float* out2a, int B, HeadArgs hd) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  actor_head_row(out2a + (size_t)b * 2 * hd.A, b, hd);
}

// gradient of (sum_a gpi[a]*pi[a] + glp*log_pi) wrt the trunk output [mu | raw_log_std]
// gpi[b][a] = gpi[b*gpi_ld + a] (+ gpi2[b*gpi_ld + a]): the action columns of the twin-Q input gradient can be read in
// place, summed over the twin (curl_sac.py:138), instead of through a `split_sum` pass
__global__ void actor_head_bwd_kernel(const float* gpi, const float* gpi2, int gpi_ld, const float* glp_scalar,
                                      const double* log_alpha, float glp_scale, const float* noise, const float* pi_t,
                                      const float* log_std, const float* tanh_ls, int B, int A, float lo, float hi,
                                      float* dout2a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  // d loss / d log_pi is the same for every row: alpha/B (actor loss) unless given explicitly
  const float glp = glp_scalar ? glp_scalar[b] : glp_scale * (float)exp(*log_alpha);
  for (int a = 0; a < A; ++a) {
    const float p = pi_t[(size_t)b * A + a];
    const float om = 1.f - p * p;
    float gp = gpi[(size_t)b * gpi_ld + a];
    if (gpi2) gp += gpi2[(size_t)b * gpi_ld + a];
    if (om > 0.f) gp += glp * (2.f * p / (om + 1e-6f));
    const float gu = gp * om;
    const float ls = log_std[(size_t)b * A + a];
    const float t = tanh_ls[(size_t)b * A + a];
    const float gls = gu * noise[(size_t)b * A + a] * expf(ls) - glp;
    dout2a[(size_t)b * 2 * A + a] = gu;
    dout2a[(size_t)b * 2 * A + A + a] = gls * 0.5f * (hi - lo) * (1.f - t * t);
  }
}

// xa[z] = [ zfeat | act ] for the twin-Q input
__global__ void concat_kernel(const float* zf, const float* act, int B, int F, int A, float* xa) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * (F + A)) return;
  const int b = i / (F + A), c = i - b * (F + A);
  xa[i] = c < F ? zf[(size_t)b * F + c] : act[(size_t)b * A + (c - F)];
}

// dxa[2][B][F+A] -> dz[B][F] (sum over the twin) and/or dact[B][A]
__global__ void split_sum_kernel(const float* dxa, long long sTwin, int B, int F, int A, float* dz, float* dact) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * (F + A)) return;
  const int b = i / (F + A), c = i - b * (F + A);
  const float v = dxa[i] + dxa[sTwin + i];
  if (c < F) {
    if (dz) dz[(size_t)b * F + c] = v;
  } else if (dact) {
    dact[(size_t)b * A + (c - F)] = v;
  }
}

// target_Q = r + not_done * gamma * (min(tq1,tq2) - alpha*log_pi)      (curl_sac.py:353-355)
__global__ void td_target_kernel(const float* tq, long long sTwin, const float* log_pi, const float* reward,
                                 const float* not_done, const double* log_alpha, float discount, int B,
                                 float* target_q) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float alpha = (float)exp(*log_alpha);
  const float v = fminf(tq[b], tq[sTwin + b]) - alpha * log_pi[b];
  target_q[b] = reward[b] + not_done[b] * discount * v;
}

// loss = mse(q1,tQ) + mse(q2,tQ); dq = 2(q - tQ)/B                    (curl_sac.py:359)
__global__ void critic_loss_kernel(const float* q, long long sTwin, const float* target_q, int B, float* loss,
                                   float* dq) {
  __shared__ float sm[4];
  float a1 = 0.f, a2 = 0.f;
  const float inv = 1.f / B;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float d1 = q[b] - target_q[b], d2 = q[sTwin + b] - target_q[b];
    a1 += d1 * d1;
    a2 += d2 * d2;
    dq[b] = 2.f * d1 * inv;
    dq[sTwin + b] = 2.f * d2 * inv;
  }
  const float s1 = block_sum_256(a1, sm);
  const float s2 = block_sum_256(a2, sm);
  if (threadIdx.x == 0) loss[0] = s1 * inv + s2 * inv;
}

// the two kernels above in one launch (the TD target only feeds the loss): target_q is still written out
__global__ void critic_td_loss_kernel(const float* q, const float* tq, long long sTwin, const float* log_pi,
                                      const float* reward, const float* not_done, const double* log_alpha,
                                      float discount, int B, float* target_q, float* loss, float* dq) {
  __shared__ float sm[4];
  float a1 = 0.f, a2 = 0.f;
  const float inv = 1.f / B;
  const float alpha = (float)exp(*log_alpha);
  for (int b = threadIdx.x; b < B; b += 256) {
    const float v = fminf(tq[b], tq[sTwin + b]) - alpha * log_pi[b];
    const float t = reward[b] + not_done[b] * discount * v;
    target_q[b] = t;
    const float d1 = q[b] - t, d2 = q[sTwin + b] - t;
    a1 += d1 * d1;
    a2 += d2 * d2;
    dq[b] = 2.f * d1 * inv;
    dq[sTwin + b] = 2.f * d2 * inv;
  }
  const float s1 = block_sum_256(a1, sm);
  const float s2 = block_sum_256(a2, sm);
  if (threadIdx.x == 0) loss[0] = s1 * inv + s2 * inv;
}

// actor/alpha losses and their seeds                                  (curl_sac.py:378-399)
// scalars: [0] actor_loss [1] alpha_loss [2] entropy mean [3] alpha
__global__ void actor_loss_kernel(const float* q, long long sTwin, const float* log_pi, const float* log_std, int A,
                                  const double* log_alpha, float target_entropy, int B, float* scalars, float* dq,
                                  double* dlog_alpha) {
  __shared__ float sm[4];
  const float alpha = (float)exp(*log_alpha);
  const float inv = 1.f / B;
  float al = 0.f, hl = 0.f, en = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float q1 = q[b], q2 = q[sTwin + b];
    al += alpha * log_pi[b] - fminf(q1, q2);
    hl += -log_pi[b] - target_entropy;
    float e = 0.f;
    for (int a = 0; a < A; ++a) e += log_std[(size_t)b * A + a];
    en += 0.5f * A * (1.0f + 1.8378770664093453f) + e;
    // d(-min)/dq: the smaller one takes -1/B; an exact tie splits it (torch.min backward)
    dq[b] = q1 < q2 ? -inv : (q1 == q2 ? -0.5f * inv : 0.f);
    dq[sTwin + b] = q2 < q1 ? -inv : (q1 == q2 ? -0.5f * inv : 0.f);
  }
  const float s_al = block_sum_256(al, sm);
  const float s_hl = block_sum_256(hl, sm);
  const float s_en = block_sum_256(en, sm);
  if (threadIdx.x == 0) {
    scalars[0] = s_al * inv;
    scalars[1] = alpha * (s_hl * inv);
    scalars[2] = s_en * inv;
    scalars[3] = alpha;
    if (dlog_alpha) *dlog_alpha = (double)(alpha * (s_hl * inv));  // d/dlog_alpha exp(log_alpha)*c = alpha*c
  }
}

// CURL InfoNCE: loss = mean_a( logsumexp_b(l[a][:]) - l[a][a] ); dl = (softmax - I)/B
__global__ void curl_ce_kernel(const float* logits, int B, int ld, float* row_loss, float* dlogits) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B) return;
  const float* l = logits + (size_t)row * ld;
  float mx = -INFINITY;
  for (int j = lane; j < B; j += 64) mx = fmaxf(mx, l[j]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int j = lane; j < B; j += 64) se += expf(l[j] - mx);
  se = wave_sum(se);
  const float lse = logf(se) + mx;
  if (lane == 0) row_loss[row] = lse - l[row];
  if (dlogits) {
    const float inv = 1.f / B;
    for (int j = lane; j < B; j += 64) {
      const float p = expf(l[j] - lse);
      dlogits[(size_t)row * ld + j] = (p - (j == row ? 1.f : 0.f)) * inv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The CURL head in ONE launch (curl_sac.py:211-222, 406-417 and their autograd): from the anchor features z_a, the
// positives' z_pos and (W z_pos^T)^T to the row losses, d(loss)/d(fc output) of the anchor encoder (LayerNorm backward
// included) and the partial sums of dW and of the LayerNorm / fc-bias gradients -- what used to be seven launches of
// ~5 us each (logits, cross-entropy, three small products, LayerNorm backward), one of which the host could not enqueue
// in time.  TWO 512-thread workgroups per 16 anchor rows (both do A; one does dz and C1, the other T and C2):
//   A  logits[16][B] = z_a rows . wz^T on the matrix pipe (a wave takes every eighth 16-column tile; k walked as 4
//      contiguous runs so that both operands are plain row reads), row max / sum of exponentials across lanes, tiles
//      and waves, dlogits = (softmax - I) / B into LDS;
//   B  dz = dlogits . wz or T = dlogits . z_pos (16 x F, k = B): a wave takes an eighth of k for all four feature tiles,
//      the A operand from LDS, the B operand one 16-byte load per lane and step from L2;
//   C  LayerNorm backward of the 16 dz rows (a wave per row), their contributions to dgamma / dbeta / fc-bias gradient,
//      and the 16-row partial of dW[i][k] = sum_a z_a[a][i] T[a][k].
// The partial sums are finished by the extra workgroup of the fc backward that follows (curla_fc_bwd_ln).
// ---------------------------------------------------------------------------------------------------------------------
struct CurlHeadArgs {
  const float *za, *zp, *wz;         // [B][F]
  const float *xhat, *rstd, *gamma;  // LayerNorm state of the anchor encoder: [B][F], [B], [F]
  float* row_loss;                   // [B]
  float* dfc;                        // [B][F]
  float* ln_partial;                 // [B / 16][3][F]
  float* w_partial;                  // [B / 16][F * F]
  float *logits, *dlogits, *dz;      // optional: [B][B], [B][B], [B][F]
  int B, F;
};

constexpr int kCurlMaxKQ = 13;  // k per lane group of phase A: F <= 52

// NTILE = B / 128: the 16-column logits tiles a wave owns in phase A (compile-time, so that all of a phase's operand
// loads are issued together: the kernel is a chain of memory latencies, not of arithmetic)
template <int NTILE>
__global__ __launch_bounds__(512) void curl_head_kernel(CurlHeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int B = 128 * NTILE, QB = B / 4;  // QB: k (= logits column) per lane group in phase B
  constexpr int LD = B + 4;                   // dlogits rows in LDS (16-byte aligned rows)
  // phase B's operand (one float per lane and k-step) is requested before phase A when it fits the register file
  // next to phase A's tiles; else in runs of 32 in front of their MFMAs
  constexpr bool PRELOAD = NTILE <= 4;  // (B / 32 <= 16 operand registers x 4)
  const int F = a.F;
  float* dl = lds;                      // [16][LD]
  float* zs = dl + 16 * LD;             // [16][64]   z_a rows of this block
  float* t0 = zs + 16 * 64;             // [16][64] the product's tile (dz or T); (second half unused)
  float* red = t0 + 2 * 16 * 64;        // [8][16] row maxima, then [8][16] row sums
  float* dg = red + 2 * 8 * 16;         // [16] diagonal logits
  float* pr = dg + 16;                  // [8][3][64] LayerNorm partials per wave
  float* pq = pr + 8 * 3 * 64;          // [8][16][64] phase B: the waves' k-eighth partial tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int rblk = blockIdx.x >> 1, mat = blockIdx.x & 1;  // two workgroups per 16 rows: matrix 0 (wz -> dz), 1 (z_pos -> T)
  const int r0 = rblk * 16;
  constexpr int KQ = kCurlMaxKQ;        // k per lane group of phase A: 4 KQ - 3 <= F <= 4 KQ (checked by the host)

  // ---- operand loads of phases A and B, all in flight together.  The texture addresser is shared by the CU's eight
  // waves and spends its time per ADDRESS, not per byte: a lane's k-run of phase A (KQ consecutive floats of a row) is
  // read as 16-byte pieces (4-byte aligned) + single floats, phase B's operand as one 16-byte load per k-step (below)
  float af[KQ], bf[NTILE][KQ];
  const int k0 = min(KQ * kq, F - KQ);  // (the last lane group's run is moved back into the row; see phase A)
  {
    auto load_run = [&](float (&dst)[KQ], const float* p) {
#pragma unroll
      for (int s4 = 0; s4 + 4 <= KQ; s4 += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + s4);
        dst[s4] = v[0], dst[s4 + 1] = v[1], dst[s4 + 2] = v[2], dst[s4 + 3] = v[3];
      }
#pragma unroll
      for (int s1 = KQ & ~3; s1 < KQ; ++s1) dst[s1] = p[s1];
    };
    load_run(af, a.za + (size_t)(r0 + li) * F + k0);
#pragma unroll
    for (int i = 0; i < NTILE; ++i) load_run(bf[i], a.wz + (size_t)(16 * (wave + 8 * i) + li) * F + k0);
  }
  // phase B: wave -> an eighth of k, all four feature tiles; tile t's column li is feature 4 li + t, so ONE
  // 16-byte load per lane and k-step feeds four MFMAs and a lane group's 16 lanes read 256 contiguous bytes
  constexpr int KW = B / 32;                      // k per lane group and wave
  const int cbase = wave * (B / 8) + KW * kq;     // first logits column (= row of wz / z_pos) of this lane group
  const int nfull = F >> 2, nrest = F & 3;        // lanes li < nfull: four features; lane nfull: the remaining nrest
  // every lane issues the same 16-byte load: lanes li < nfull at their four features, the others at the row's LAST four
  // floats (in range), from which lane nfull picks its nrest features; lanes beyond hold zeros
  const float* mrow = (mat ? a.zp : a.wz) + (size_t)cbase * F + (li < nfull ? 4 * li : F - 4);
  f32x4 bv[KW <= 16 ? KW : 8];
  auto load_bv = [&](f32x4& v, int c) { v = *reinterpret_cast<const f32x4*>(mrow + (size_t)c * F); };
  auto fix_bv = [&](f32x4& v) {  // (features 4 nfull + t sit at position t + 4 - nrest of the row's last four floats)
    if (li >= nfull) {
      const f32x4 w = v;
      const bool mine = li == nfull;
      v[0] = (mine && nrest > 0) ? (nrest == 1 ? w[3] : nrest == 2 ? w[2] : w[1]) : 0.f;
      v[1] = (mine && nrest > 1) ? (nrest == 2 ? w[3] : w[2]) : 0.f;
      v[2] = (mine && nrest > 2) ? w[3] : 0.f;
      v[3] = 0.f;
    }
  };
  if (PRELOAD) {
#pragma unroll
    for (int c = 0; c < KW; ++c) load_bv(bv[c], c);
  }
  for (int i = tid; i < 16 * 64; i += 512) {
    const int r = i >> 6, f = i & 63;
    zs[i] = f < F ? a.za[(size_t)(r0 + r) * F + f] : 0.f;
  }
  // ---- A: logits of rows r0 .. r0 + 15.  A lane group's run starts at k0 = min(KQ kq, F - KQ): the k below KQ kq of the
  // moved run belong to the previous group and are dropped (a zero on the z_a side is enough)
#pragma unroll
  for (int s = 0; s < KQ; ++s) af[s] = k0 + s >= KQ * kq ? af[s] : 0.f;
  f32x4 lg[NTILE];
#pragma unroll
  for (int i = 0; i < NTILE; ++i) {
    lg[i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < kCurlMaxKQ; ++s) lg[i] = mfma16(af[s], bf[i][s], lg[i]);
  }
  // lane (li, kq) holds logits[r0 + 4 kq + r][16 (wave + 8 i) + li]
  float mx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float m = lg[0][r];
#pragma unroll
    for (int i = 1; i < NTILE; ++i) m = fmaxf(m, lg[i][r]);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (li == 0) red[wave * 16 + 4 * kq + r] = m;
  }
#pragma unroll
  for (int i = 0; i < NTILE; ++i) {
    const int col = 16 * (wave + 8 * i) + li;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col == r0 + 4 * kq + r) dg[4 * kq + r] = lg[i][r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float m = red[4 * kq + r];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w * 16 + 4 * kq + r]);
    mx[r] = m;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NTILE; ++i) sum += expf(lg[i][r] - m);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o);
    if (li == 0) red[128 + wave * 16 + 4 * kq + r] = sum;
  }
  __syncthreads();
  const float inv = 1.f / B;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float sum = red[128 + 4 * kq + r];
#pragma unroll
    for (int w = 1; w < 8; ++w) sum += red[128 + w * 16 + 4 * kq + r];
    const float lse = logf(sum) + mx[r];
    const int row = r0 + 4 * kq + r;
    if (mat == 0 && wave == 0 && li == 0) a.row_loss[row] = lse - dg[4 * kq + r];
#pragma unroll
    for (int i = 0; i < NTILE; ++i) {
      const int col = 16 * (wave + 8 * i) + li;
      const float d = (expf(lg[i][r] - lse) - (col == row ? 1.f : 0.f)) * inv;
      dl[(4 * kq + r) * LD + col] = d;
      if (mat == 0 && a.logits) a.logits[(size_t)row * B + col] = lg[i][r];
      if (mat == 0 && a.dlogits) a.dlogits[(size_t)row * B + col] = d;
    }
  }
  __syncthreads();
  // ---- B: this workgroup's product -- dz = dl . wz (matrix 0) or T = dl . z_pos (matrix 1): a wave multiplies its
  // eighth of k into all four feature tiles (four independent accumulator chains), the eighths are then added in
  // order through LDS
  {
    const float* arow = dl + li * LD + cbase;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0, 0, 0, 0};
    if (PRELOAD) {
#pragma unroll
      for (int c = 0; c < KW; ++c) fix_bv(bv[c]);
#pragma unroll
      for (int c0 = 0; c0 < KW; c0 += 4) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(arow + c0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = mfma16(av[u], bv[c0 + u][t], acc[t]);
      }
    } else {
      // runs of CH k-steps: 8 where KW = 4 NTILE divides by 8, else 4 (NTILE 5, 7: a run of 8 would walk into the next
      // lane group's columns and, in the last wave, past the operand's last row)
      constexpr int CH = (KW % 8 == 0) ? 8 : 4;
      static_assert(KW % CH == 0, "phase B walks whole runs");
      for (int c0 = 0; c0 < KW; c0 += CH) {
#pragma unroll
        for (int u = 0; u < CH; ++u) load_bv(bv[u], c0 + u);
#pragma unroll
        for (int u = 0; u < CH; ++u) fix_bv(bv[u]);
#pragma unroll
        for (int u4 = 0; u4 < CH; u4 += 4) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(arow + c0 + u4);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma16(av[u], bv[u4 + u][t], acc[t]);
        }
      }
    }
    // lane (li, kq) of tile t: rows 4 kq + r, feature 4 li + t.  Partial tiles -> pq[wave][row][feature]
    float* qp = pq + (wave * 16) * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) qp[(4 * kq + r) * 64 + 4 * li + t] = acc[t][r];
  }
  __syncthreads();
  for (int i = tid; i < 16 * 64; i += 512) {  // the product's 16 x 64 tile: the eight k ranges in order
    float v = pq[i];
#pragma unroll
    for (int w = 1; w < 8; ++w) v += pq[w * 16 * 64 + i];
    t0[i] = v;
  }
  __syncthreads();
  if (mat == 1) {
    // ---- C2: this block's 16 rows of dW[i][k] = sum_a z_a[a][i] T[a][k]
    float* wp = a.w_partial + (size_t)rblk * F * F;
    for (int idx = tid; idx < F * F; idx += 512) {
      const int i = idx / F, kk = idx - i * F;
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += zs[r * 64 + i] * t0[r * 64 + kk];
      wp[idx] = acc;
    }
    return;
  }
  // ---- C1: LayerNorm backward of rows 2 wave, 2 wave + 1 (lane = feature); partial column sums in row order
  {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    const bool fv = lane < F;
    const float gm = fv ? a.gamma[lane] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wave + rr, grow = r0 + row;
      const float d = fv ? t0[row * 64 + lane] : 0.f;
      const float xh = fv ? a.xhat[(size_t)grow * F + lane] : 0.f;
      const float g = d * gm;
      const float m1 = wave_sum(g) / F, m2 = wave_sum(g * xh) / F;
      const float o = a.rstd[grow] * (g - m1 - xh * m2);
      if (fv) {
        a.dfc[(size_t)grow * F + lane] = o;
        if (a.dz) a.dz[(size_t)grow * F + lane] = d;
      }
      p0 += d * xh, p1 += d, p2 += fv ? o : 0.f;
    }
    pr[(wave * 3 + 0) * 64 + lane] = p0, pr[(wave * 3 + 1) * 64 + lane] = p1, pr[(wave * 3 + 2) * 64 + lane] = p2;
  }
  __syncthreads();
  for (int i = tid; i < 3 * F; i += 512) {
    const int q = i / F, f = i - q * F;
    float v = pr[(0 * 3 + q) * 64 + f];
#pragma unroll
    for (int w = 1; w < 8; ++w) v += pr[(w * 3 + q) * 64 + f];
    a.ln_partial[(size_t)rblk * 3 * F + i] = v;
  }
}

__global__ void mean_kernel(const float* x, int n, float* out) {
  __shared__ float sm[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += x[i];
  const float s = block_sum_256(a, sm);
  if (threadIdx.x == 0) out[0] = s / n;
}

// target <- tau*p + (1-tau)*target                                     (utils.py:37-41)
__global__ void soft_update_kernel(const float* p, float* tgt, size_t n, float tau, float omt) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // two roundings of the products and one of the sum, as `tau * p + (1 - tau) * t` evaluates in torch (no fused
  // multiply-add), so the targets match the reference's bit for bit given the same inputs
  for (; i < n; i += stride) {
#pragma clang fp contract(off)
    tgt[i] = tau * p[i] + omt * tgt[i];
  }
}

// the same over one flat block whose first `split` elements take (tau_a, 1-tau_a) and the rest (tau_b, 1-tau_b):
// the encoder and the Q functions of the critic are adjacent in the flat layout and have their own rates
__global__ void soft_update2_kernel(const float* p, float* tgt, size_t n, size_t split, float tau_a, float omt_a,
                                    float tau_b, float omt_b) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
#pragma clang fp contract(off)
    const bool a = i < split;
    tgt[i] = (a ? tau_a : tau_b) * p[i] + (a ? omt_a : omt_b) * tgt[i];
  }
}

// One Adam step over a flat run of parameters (torch.optim.Adam, weight_decay 0, no amsgrad; curl_sac.py:299-313):
//   m += (1-b1)(g-m);  v = b2 v + (1-b2) g g;  p -= step_size * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// with step_size = lr/(1-b1^t) and the two bias corrections evaluated on the host in double, as torch's
// single-tensor Adam does.  Seven fp32 streams (4 read, 3 written), 16-byte accesses when the run is 16-byte
// aligned: HBM-bound (28 B per parameter).
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float w1, float b2, float w2,
                                         float step_size, float bc2_sqrt, float eps) {
  // (no fused multiply-adds: every product and sum rounds on its own, as torch's element-wise Adam does, and the
  // one-step and two-step kernels then agree bit for bit whatever the compiler would have contracted in each)
#pragma clang fp contract(off)
  m = (w1 < 0.5f) ? m + w1 * (g - m) : g - (g - m) * (1.0f - w1);
  v = v * b2 + (w2 * g) * g;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - (step_size * m) / denom;
}

// A float64 scalar parameter stepped by the same launch (log_alpha beside the actor's parameters: the reference steps
// actor_optimizer and log_alpha_optimizer back to back, curl_sac.py:393-404); p == nullptr: none.
struct AdamScalar64 {
  double *p, *m, *v;
  const double* g;
  double w1, b2, w2, step_size, bc2_sqrt, eps;
  const double* dyn;  // != nullptr: step_size = dyn[0], bc2_sqrt = dyn[1], read when the kernel runs
};

// The target network's soft update in the same pass: target <- tau p + (1 - tau) target with the parameter this launch
// has just stepped (utils.py:37-41 right after critic_optimizer.step(), curl_sac.py:367,442-445); elements [0, split)
// take (tau_a, omt_a), the rest (tau_b, omt_b).  tgt == nullptr: none.
struct AdamLerp {
  float* tgt;
  size_t split;
  float tau_a, omt_a, tau_b, omt_b;
};
__device__ __forceinline__ float lerp_one(float p, float t, float tau, float omt) {
#pragma clang fp contract(off)
  return tau * p + omt * t;  // two roundings of the products and one of the sum, as soft_update_kernel
}

__global__ void __launch_bounds__(256) adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, size_t n,
                                                        int vec, float w1, float b2, float w2, float step_size,
                                                        float bc2_sqrt, float eps, AdamScalar64 sc, AdamLerp lp,
                                                        const float* __restrict__ dyn) {
  // dyn != nullptr: the two step-dependent factors lr / (1 - b1^t) and sqrt(1 - b2^t) come from device memory, written
  // there (by the host, through the update's pinned control block) before the kernel runs: a captured graph is
  // replayed with new step counts.  The same two floats the host would have passed by value.
  if (dyn) step_size = dyn[0], bc2_sqrt = dyn[1];
  if (sc.p && blockIdx.x == 0 && threadIdx.x == 0) {  // torch's single-tensor Adam, in double
#pragma clang fp contract(off)
    if (sc.dyn) sc.step_size = sc.dyn[0], sc.bc2_sqrt = sc.dyn[1];
    const double gs = *sc.g;
    double ms = *sc.m, vs = *sc.v;
    ms = (sc.w1 < 0.5) ? ms + sc.w1 * (gs - ms) : gs - (gs - ms) * (1.0 - sc.w1);
    vs = vs * sc.b2 + (sc.w2 * gs) * gs;
    *sc.m = ms, *sc.v = vs;
    *sc.p = *sc.p - sc.step_size * (ms / (sqrt(vs) / sc.bc2_sqrt + sc.eps));
  }
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  size_t done = 0;
  if (vec) {
    const size_t n4 = n >> 2;
    for (size_t i = tid; i < n4; i += stride) {
      f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
      const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pp[e], me = mm[e], ve = vv[e];
        adam_one(pe, gg[e], me, ve, w1, b2, w2, step_size, bc2_sqrt, eps);
        pp[e] = pe, mm[e] = me, vv[e] = ve;
      }
      reinterpret_cast<f32x4*>(p)[i] = pp;
      reinterpret_cast<f32x4*>(m)[i] = mm;
      reinterpret_cast<f32x4*>(v)[i] = vv;
      if (lp.tgt) {  // (vec implies target and split are 16-byte granular too: a float4 lies on one side of split)
        f32x4 tt = reinterpret_cast<f32x4*>(lp.tgt)[i];
        const bool a = 4 * i < lp.split;
#pragma unroll
        for (int e = 0; e < 4; ++e) tt[e] = lerp_one(pp[e], tt[e], a ? lp.tau_a : lp.tau_b, a ? lp.omt_a : lp.omt_b);
        reinterpret_cast<f32x4*>(lp.tgt)[i] = tt;
      }
    }
    done = n4 << 2;
  }
  for (size_t i = done + tid; i < n; i += stride) {
    adam_one(p[i], g[i], m[i], v[i], w1, b2, w2, step_size, bc2_sqrt, eps);
    if (lp.tgt) lp.tgt[i] = lerp_one(p[i], lp.tgt[i], i < lp.split ? lp.tau_a : lp.tau_b, i < lp.split ? lp.omt_a : lp.omt_b);
  }
}

// Two Adam steps of two optimizers on the SAME parameters with the same gradient, back to back, in one pass (the
// reference steps the encoder with encoder_optimizer and then again with cpc_optimizer, curl_sac.py:418-423): elements
// [0, n_pre) belong to the second optimizer only (CURL.W in front of the encoder in the flat buffer), the rest take
// step 1 then step 2 exactly as the two launches would (the same per-element operation order).
struct AdamHyper {
  float w1, b2, w2, step_size, bc2_sqrt, eps;
};

__global__ void __launch_bounds__(256) adam_step2_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m1, float* __restrict__ v1,
                                                         float* __restrict__ m2, float* __restrict__ v2, size_t n,
                                                         size_t n_pre, int vec, AdamHyper h1, AdamHyper h2,
                                                         const float* __restrict__ dyn) {
  if (dyn) h1.step_size = dyn[0], h1.bc2_sqrt = dyn[1], h2.step_size = dyn[2], h2.bc2_sqrt = dyn[3];  // (adam_step_kernel)
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  size_t done = 0;
  if (vec) {  // everything 16-byte aligned and n_pre a multiple of 4: a float4 lies on one side of n_pre
    const size_t n4 = n >> 2, pre4 = n_pre >> 2;
    for (size_t i = tid; i < n4; i += stride) {
      f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
      const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
      if (i >= pre4) {
        f32x4 mm = reinterpret_cast<f32x4*>(m1)[i - pre4], vv = reinterpret_cast<f32x4*>(v1)[i - pre4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pe = pp[e], me = mm[e], ve = vv[e];
          adam_one(pe, gg[e], me, ve, h1.w1, h1.b2, h1.w2, h1.step_size, h1.bc2_sqrt, h1.eps);
          pp[e] = pe, mm[e] = me, vv[e] = ve;
        }
        reinterpret_cast<f32x4*>(m1)[i - pre4] = mm;
        reinterpret_cast<f32x4*>(v1)[i - pre4] = vv;
      }
      f32x4 mm = reinterpret_cast<f32x4*>(m2)[i], vv = reinterpret_cast<f32x4*>(v2)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pp[e], me = mm[e], ve = vv[e];
        adam_one(pe, gg[e], me, ve, h2.w1, h2.b2, h2.w2, h2.step_size, h2.bc2_sqrt, h2.eps);
        pp[e] = pe, mm[e] = me, vv[e] = ve;
      }
      reinterpret_cast<f32x4*>(m2)[i] = mm;
      reinterpret_cast<f32x4*>(v2)[i] = vv;
      reinterpret_cast<f32x4*>(p)[i] = pp;
    }
    done = n4 << 2;
  }
  for (size_t i = done + tid; i < n; i += stride) {
    float pi = p[i];
    const float gi = g[i];
    if (i >= n_pre) {
      float m = m1[i - n_pre], v = v1[i - n_pre];
      adam_one(pi, gi, m, v, h1.w1, h1.b2, h1.w2, h1.step_size, h1.bc2_sqrt, h1.eps);
      m1[i - n_pre] = m, v1[i - n_pre] = v;
    }
    float m = m2[i], v = v2[i];
    adam_one(pi, gi, m, v, h2.w1, h2.b2, h2.w2, h2.step_size, h2.bc2_sqrt, h2.eps);
    m2[i] = m, v2[i] = v;
    p[i] = pi;
  }
}

// out_act[b][:] = sc[idx[b]][0:A], out_rew[b] = sc[idx[b]][A], out_nd[b] = sc[idx[b]][A+1]: the action / reward /
// not_done of the sampled transitions (utils.py:159-166) from the ring's [capacity][A+2] scalar rows, one launch
__global__ void gather_transition_scalars_kernel(const float* sc, const int64_t* idx, int B, int A, float* act,
                                                 float* rew, float* nd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * (A + 2)) return;
  const int b = i / (A + 2), c = i - b * (A + 2);
  const float v = sc[(size_t)idx[b] * (A + 2) + c];
  if (c < A)
    act[(size_t)b * A + c] = v;
  else if (c == A)
    rew[b] = v;
  else
    nd[b] = v;
}

// The same gather with the minibatch's index block taken straight from pinned host memory: the block (indices, crop
// offsets; ~20 KB) is read over PCIe with system-scope loads and written to its device slot by this kernel, the
// scalar rows are gathered with the indices as they arrive -- no copy-engine transfer (and its queue hand-over, ~15 us
// of idle stream) in front of every update.  host / dev: the block as 8-byte words, idx = its first B words.
__global__ void sample_stage_kernel(const unsigned long long* host, unsigned long long* dev, int nwords, const float* sc,
                                    int B, int A, float* act, float* rew, float* nd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nwords) dev[i] = __hip_atomic_load(host + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (i >= B * (A + 2)) return;
  const int b = i / (A + 2), c = i - b * (A + 2);
  const long long row = (long long)__hip_atomic_load(host + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const float v = sc[(size_t)row * (A + 2) + c];
  if (c < A)
    act[(size_t)b * A + c] = v;
  else if (c == A)
    rew[b] = v;
  else
    nd[b] = v;
}

// out[b][c][i][j] = (float) frames[idx[b]][h1[b]+i][w1[b]+j][c]          (augmentations.py:47-75 + utils.py:161)
__global__ void crop_nchw_kernel(const uint8_t* frames, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                                 int B, int C, int Hs, int Ws, int Hc, int Wc, float* out_f32, uint8_t* out_u8) {
  const size_t n = (size_t)B * C * Hc * Wc;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int x = i % Wc;
    size_t t = i / Wc;
    const int y = t % Hc;
    t /= Hc;
    const int c = t % C;
    const int b = t / C;
    const int64_t fi = idx ? idx[b] : b;
    const int oh = h1 ? h1[b] : 0, ow = w1 ? w1[b] : 0;
    const uint8_t v = frames[(((size_t)fi * Hs + oh + y) * Ws + ow + x) * C + c];
    if (out_f32) out_f32[i] = (float)v;
    if (out_u8) out_u8[i] = v;
  }
}

// frames[slot][y][x][c] = chw[c][y][x]   (replay add: the reference stores CHW, utils.py:120-128)
__global__ void store_frame_kernel(const uint8_t* chw, uint8_t* frames, int64_t slot, int C, int H, int W) {
  const int n = C * H * W;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = i % C, p = i / C;
    frames[(size_t)slot * n + i] = chw[(size_t)c * H * W + p];
  }
}

// De-duplicated replay storage: every RGB frame is stored once ([F][H][W][3] uint8) and a transition keeps the k frame
// ids of its observation stack.  out[b][y][x][3f+c] = store[fid[idx[b]][f]][y][x][c] rebuilds the [B][H][W][3k] stacks
// of a minibatch (the layout the first conv's loader reads).  One thread moves 4 pixels: 3 aligned dwords from each
// of the k frames, 3k aligned dwords out (FAST); any other geometry goes byte by byte.
template <bool FAST>
__global__ void gather_stacks_kernel(const uint8_t* store, const int32_t* fid, int fid_stride, const int64_t* idx, int B,
                                     int K, int HW, uint8_t* out) {
  const int groups = (HW + 3) >> 2;
  const size_t n = (size_t)B * groups;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int C = 3 * K;
  for (; i < n; i += stride) {
    const int b = i / groups, g = i - (size_t)b * groups;
    const int32_t* ids = fid + (size_t)(idx ? idx[b] : b) * fid_stride;
    uint8_t* o = out + ((size_t)b * HW + 4 * (size_t)g) * C;
    if (FAST) {
      uint32_t ob[12];  // up to k = 4 frames: 4 pixels x 12 bytes
#pragma unroll
      for (int w = 0; w < 12; ++w) ob[w] = 0;
      for (int f = 0; f < K; ++f) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(store + ((size_t)ids[f] * HW + 4 * (size_t)g) * 3);
        const uint32_t s0 = src[0], s1 = src[1], s2 = src[2];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int sb = 3 * p + c;  // byte of the 12 source bytes
            const uint32_t v = ((sb < 4 ? s0 : sb < 8 ? s1 : s2) >> (8 * (sb & 3))) & 0xffu;
            const int db = p * C + 3 * f + c;  // byte of the 4*C output bytes
            ob[db >> 2] |= v << (8 * (db & 3));
          }
      }
      uint32_t* o4 = reinterpret_cast<uint32_t*>(o);
      for (int w = 0; w < C; ++w) o4[w] = ob[w];  // 4 pixels x C bytes = C dwords
    } else {
      const int np = min(4, HW - 4 * g);
      for (int f = 0; f < K; ++f) {
        const uint8_t* src = store + ((size_t)ids[f] * HW + 4 * (size_t)g) * 3;
        for (int p = 0; p < np; ++p)
          for (int c = 0; c < 3; ++c) o[p * C + 3 * f + c] = src[3 * p + c];
      }
    }
  }
}

// NHWC float activation -> NCHW (for callers that read encoder.outputs)
__global__ void nhwc_to_nchw_kernel(const float* in, float* out, int B, int H, int W, int C) {
  const size_t n = (size_t)B * H * W * C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int x = i % W;
    size_t t = i / W;
    const int y = t % H;
    t /= H;
    const int c = t % C;
    const int b = t / C;
    out[i] = in[(((size_t)b * H + y) * W + x) * C + c];
  }
}

inline int nblocks(size_t n, int bs, int cap = 4096) {
  size_t b = (n + bs - 1) / bs;
  return (int)(b < (size_t)cap ? b : cap);
}

}  // namespace

extern "C" {

int curla_fc_ln_fwd_multi(int njobs, const CurlaFcLnJob* jobs, int nsplit, long long split_stride, int ldp, int B,
                          int F, float eps, int A, void* stream) {
  CURLA_REQUIRE(jobs && njobs > 0 && njobs <= kMaxLnJobs && B > 0 && F > 0 && nsplit > 0 && A >= 0 && A <= 64);
  FcLnJobs js;
  for (int i = 0; i < njobs; ++i) {
    const CurlaFcLnJob& j = jobs[i];
    CURLA_REQUIRE(j.partial && j.bias && j.gamma && j.beta && j.y);
    CURLA_REQUIRE(!j.xa || A > 0);
    js.j[i] = j;
  }
  if (F > 1024) return CURLA_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define CURLA_FC_LN(NF)                                                                                          \
  hipLaunchKernelGGL(fc_ln_fwd_kernel<NF>, dim3((B + 3) / 4, njobs), dim3(256), 0, st, js, nsplit, split_stride, ldp, \
                     B, F, eps, A)
  switch ((F + 63) / 64) {
    case 1: CURLA_FC_LN(1); break;
    case 2: CURLA_FC_LN(2); break;
    case 3: CURLA_FC_LN(3); break;
    case 4: CURLA_FC_LN(4); break;
    case 5: case 6: case 7: case 8: CURLA_FC_LN(8); break;
    default: CURLA_FC_LN(16); break;
  }
#undef CURLA_FC_LN
  return curla_launch_status();
}

int curla_fc_ln_fwd(const float* partial, int nsplit, long long split_stride, int ldp, const float* bias,
                    const float* gamma, const float* beta, int B, int F, float eps, float* fc_out, float* y,
                    float* xhat, float* rstd, int tanh_out, float* xa, const float* act, int A, void* stream) {
  CURLA_REQUIRE(!xa || (act && A > 0 && A <= 64));
  CurlaFcLnJob j;
  j.partial = partial, j.bias = bias, j.gamma = gamma, j.beta = beta, j.fc_out = fc_out, j.y = y, j.xhat = xhat;
  j.rstd = rstd, j.xa = xa, j.act = act, j.tanh_out = tanh_out;
  return curla_fc_ln_fwd_multi(1, &j, nsplit, split_stride, ldp, B, F, eps, xa ? A : 0, stream);
}

int curla_ln_bwd_twin(const float* dy, const float* dy2, int ld_dy, const float* xhat, const float* rstd,
                      const float* gamma, int B, int F, float* dx, float* dgamma, float* dbeta, float* dbias_in,
                      void* stream) {
  CURLA_REQUIRE(dy && xhat && rstd && gamma && dx && B > 0 && F > 0 && ld_dy >= F);
  if (F > 1024) return CURLA_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CURLA_REQUIRE(!dbias_in || (dgamma && dbeta));
  float* const no_partial = nullptr;
#define CURLA_LN_BWD(NF, PART)                                                                                          \
  hipLaunchKernelGGL(ln_bwd_kernel<NF>, dim3((B + 3) / 4), dim3(256), 0, st, dy, dy2, ld_dy, xhat, rstd, gamma, B, F, dx, \
                     PART)
  switch ((F + 63) / 64) {
    case 1: CURLA_LN_BWD(1, no_partial); break;
    case 2: CURLA_LN_BWD(2, no_partial); break;
    case 3: CURLA_LN_BWD(3, no_partial); break;
    case 4: CURLA_LN_BWD(4, no_partial); break;
    case 5: case 6: case 7: case 8: CURLA_LN_BWD(8, no_partial); break;
    default: CURLA_LN_BWD(16, no_partial); break;
  }
  if (dgamma && dbeta)
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3((F + 15) / 16), dim3(1024), 0, st, dy, dy2, ld_dy, xhat, dx, B, F,
                       dgamma, dbeta, dbias_in);
  return curla_launch_status();
}

int curla_ln_bwd_partial(const float* dy, const float* dy2, int ld_dy, const float* xhat, const float* rstd,
                         const float* gamma, int B, int F, float* dx, float* partial, int* nparts, void* stream) {
  CURLA_REQUIRE(dy && xhat && rstd && gamma && dx && partial && nparts && B > 0 && F > 0 && ld_dy >= F);
  if (F > 1024) return CURLA_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch ((F + 63) / 64) {
    case 1: CURLA_LN_BWD(1, partial); break;
    case 2: CURLA_LN_BWD(2, partial); break;
    case 3: CURLA_LN_BWD(3, partial); break;
    case 4: CURLA_LN_BWD(4, partial); break;
    case 5: case 6: case 7: case 8: CURLA_LN_BWD(8, partial); break;
    default: CURLA_LN_BWD(16, partial); break;
  }
#undef CURLA_LN_BWD
  *nparts = (B + 3) / 4;
  return curla_launch_status();
}

int curla_ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, int B, int F, float* dx,
                 float* dgamma, float* dbeta, float* dbias_in, void* stream) {
  return curla_ln_bwd_twin(dy, nullptr, F, xhat, rstd, gamma, B, F, dx, dgamma, dbeta, dbias_in, stream);
}

int curla_colsum(const float* X, int M, int N, int ldx, long long strideX, float* out, long long strideOut, int nbatch,
                 void* stream) {
  CURLA_REQUIRE(X && out && M > 0 && N > 0 && nbatch > 0);
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 31) / 32, nbatch), dim3(1024), 0, static_cast<hipStream_t>(stream), X, M,
                     N, ldx, strideX, out, strideOut);
  return curla_launch_status();
}

int curla_colsum3(const float* X0, int N0, const float* X1, int N1, const float* X2, int N2, int M, float* out0,
                  float* out1, float* out2, long long strideOut, int nbatch, void* stream) {
  CURLA_REQUIRE(X0 && X1 && X2 && out0 && out1 && out2 && M > 0 && N0 > 0 && N1 > 0 && N2 > 0 && nbatch > 0);
  Colsum3Args a;
  a.X[0] = X0, a.X[1] = X1, a.X[2] = X2, a.out[0] = out0, a.out[1] = out1, a.out[2] = out2;
  a.N[0] = N0, a.N[1] = N1, a.N[2] = N2;
  a.sX[0] = (long long)M * N0, a.sX[1] = (long long)M * N1, a.sX[2] = (long long)M * N2;
  const int nmax = N0 > N1 ? (N0 > N2 ? N0 : N2) : (N1 > N2 ? N1 : N2);
  hipLaunchKernelGGL(colsum3_kernel, dim3((nmax + 31) / 32, nbatch, 3), dim3(1024), 0, static_cast<hipStream_t>(stream), a, M,
                     strideOut);
  return curla_launch_status();
}

static int mlp_out_fwd_launch(const float* h, long long sH, long long sH2, const float* W, long long sW, long long sW2,
                              const float* bias, long long sB, long long sB2, float* out, long long sOut, long long sOut2,
                              int M, int N, int K, int nb, int nb2, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  // (R = 4 rows per wave to share the weight loads was measured for the actor's 8 outputs: 25 us against 14 -- the
  // layer is load latency, and 128 waves hide less of it than 512)
  const size_t wbytes = (size_t)N * K * sizeof(float);
  if (N >= 2 && wbytes <= 48 * 1024)  // (one row: a wave reads 4 KB once, nothing to share)
    hipLaunchKernelGGL((mlp_out_fwd_kernel<false, 1, true>), dim3((M + 3) / 4, nb * nb2), dim3(256), wbytes, st, h, sH,
                       sH2, W, sW, sW2, bias, sB, sB2, out, sOut, sOut2, M, N, K, nb, HeadArgs{});
  else
    hipLaunchKernelGGL((mlp_out_fwd_kernel<false, 1, false>), dim3((M + 3) / 4, nb * nb2), dim3(256), 0, st, h, sH, sH2,
                       W, sW, sW2, bias, sB, sB2, out, sOut, sOut2, M, N, K, nb, HeadArgs{});
  return curla_launch_status();
}

int curla_mlp_out_fwd(const float* h, long long strideH, const float* W, long long strideW, const float* bias,
                      long long strideBias, float* out, long long strideOut, int M, int N, int K, int nbatch,
                      void* stream) {
  CURLA_REQUIRE(h && W && out && M > 0 && N > 0 && K > 0 && nbatch > 0);
  if (N > kMaxOut || K % 4 != 0 || strideH % 4 != 0 || strideW % 4 != 0) return CURLA_ERR_UNSUPPORTED;
  CURLA_REQUIRE(aligned16(h) && aligned16(W));
  return mlp_out_fwd_launch(h, strideH, 0LL, W, strideW, 0LL, bias, strideBias, 0LL, out, strideOut, 0LL, M, N, K, nbatch,
                            1, stream);
}

int curla_mlp_out_fwd_nested(const float* h, long long strideH, long long strideH2, const float* W, long long strideW,
                             long long strideW2, const float* bias, long long strideBias, long long strideBias2,
                             float* out, long long strideOut, long long strideOut2, int M, int N, int K, int nbatch,
                             int nbatch2, void* stream) {
  CURLA_REQUIRE(h && W && out && M > 0 && N > 0 && K > 0 && nbatch > 0 && nbatch2 > 0);
  if (N > kMaxOut || K % 4 != 0 || strideH % 4 != 0 || strideW % 4 != 0 || strideH2 % 4 != 0 || strideW2 % 4 != 0)
    return CURLA_ERR_UNSUPPORTED;
  CURLA_REQUIRE(aligned16(h) && aligned16(W));
  return mlp_out_fwd_launch(h, strideH, strideH2, W, strideW, strideW2, bias, strideBias, strideBias2, out, strideOut,
                            strideOut2, M, N, K, nbatch, nbatch2, stream);
}

int curla_mlp_out_bwd_bias(const float* dy, long long strideDy, const float* h, long long strideH, const float* W,
                           long long strideW, float* dh, long long strideDh, float* dW, long long strideDW, int M, int N,
                           int K, int nbatch, float* db_out, float* db_hidden, long long strideDb, void* stream) {
  CURLA_REQUIRE(dy && h && W && dh && M > 0 && N > 0 && K > 0 && nbatch > 0);
  if (N > kMaxOut) return CURLA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(mlp_out_bwd_kernel<0>, dim3((K + 15) / 16, nbatch), dim3(1024), 0, static_cast<hipStream_t>(stream),
                     dy, strideDy, h, strideH, W, strideW, dh, strideDh, dW, strideDW, M, N, K, db_out, db_hidden,
                     strideDb, LossGen{});
  return curla_launch_status();
}

int curla_mlp_out_bwd_loss(const CurlaLossArgs* loss, const float* h, long long strideH, const float* W,
                           long long strideW, float* dh, long long strideDh, float* dW, long long strideDW, int B, int K,
                           float* db_out, float* db_hidden, long long strideDb, void* stream) {
  CURLA_REQUIRE(loss && h && W && dh && B > 0 && K > 0);
  CURLA_REQUIRE(loss->q && loss->log_pi && loss->log_alpha && loss->scalars && loss->dq && loss->twin_stride >= B);
  LossGen lg;
  lg.q = loss->q, lg.tq = loss->target_q_twin, lg.log_pi = loss->log_pi, lg.reward = loss->reward;
  lg.not_done = loss->not_done, lg.log_std = loss->log_std, lg.log_alpha = loss->log_alpha;
  lg.dlog_alpha = loss->dlog_alpha, lg.target_q = loss->target_q, lg.scalars = loss->scalars, lg.dq = loss->dq;
  lg.sTwin = loss->twin_stride, lg.discount = loss->discount, lg.target_entropy = loss->target_entropy, lg.A = loss->A;
  const dim3 grid((K + 15) / 16, 2);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (loss->kind == 1) {
    CURLA_REQUIRE(loss->target_q_twin && loss->reward && loss->not_done && loss->target_q);
    hipLaunchKernelGGL(mlp_out_bwd_kernel<1>, grid, dim3(1024), 0, st, nullptr, 0LL, h, strideH, W, strideW, dh, strideDh,
                       dW, strideDW, B, 1, K, db_out, db_hidden, strideDb, lg);
  } else if (loss->kind == 2) {
    CURLA_REQUIRE(loss->log_std && loss->A > 0);
    hipLaunchKernelGGL(mlp_out_bwd_kernel<2>, grid, dim3(1024), 0, st, nullptr, 0LL, h, strideH, W, strideW, dh, strideDh,
                       dW, strideDW, B, 1, K, db_out, db_hidden, strideDb, lg);
  } else {
    return CURLA_ERR_ARG;
  }
  return curla_launch_status();
}

int curla_mlp_out_bwd(const float* dy, long long strideDy, const float* h, long long strideH, const float* W,
                      long long strideW, float* dh, long long strideDh, float* dW, long long strideDW, int M, int N,
                      int K, int nbatch, void* stream) {
  return curla_mlp_out_bwd_bias(dy, strideDy, h, strideH, W, strideW, dh, strideDh, dW, strideDW, M, N, K, nbatch,
                                nullptr, nullptr, 0, stream);
}

static int head_args(const float* noise, int B, int A, float log_std_min, float log_std_max, float* mu, float* pi,
                     float* log_pi, float* log_std, float* tanh_ls, float* pi_xa, int xa_ld, HeadArgs* hd) {
  CURLA_REQUIRE(B > 0 && A > 0 && A <= kMaxA);
  CURLA_REQUIRE(!noise || pi || pi_xa);
  CURLA_REQUIRE(!pi_xa || (noise && xa_ld >= A));
  hd->noise = noise, hd->mu_t = mu, hd->pi_t = pi, hd->log_pi = log_pi, hd->log_std = log_std, hd->tanh_ls = tanh_ls;
  hd->pi_xa = pi_xa, hd->A = A, hd->xa_ld = xa_ld, hd->lo = log_std_min, hd->hi = log_std_max;
  hd->noise_gen = nullptr, hd->rng_seed = 0, hd->rng_offset = 0, hd->rng_dev = nullptr;
  return CURLA_OK;
}

// the same with the noise drawn inside the kernel (written to noise_out [B][A])
static int head_args_rng(float* noise_out, unsigned long long seed, unsigned long long offset,
                         const unsigned long long* rng_dev, int B, int A, float log_std_min, float log_std_max, float* mu,
                         float* pi, float* log_pi, float* log_std, float* tanh_ls, float* pi_xa, int xa_ld, HeadArgs* hd) {
  CURLA_REQUIRE(noise_out && (reinterpret_cast<uintptr_t>(rng_dev) & 7) == 0);
  const int rc = head_args(noise_out, B, A, log_std_min, log_std_max, mu, pi, log_pi, log_std, tanh_ls, pi_xa, xa_ld, hd);
  if (rc != CURLA_OK) return rc;
  hd->noise = nullptr, hd->noise_gen = noise_out, hd->rng_seed = seed, hd->rng_offset = offset, hd->rng_dev = rng_dev;
  return CURLA_OK;
}

int curla_actor_head_fwd(const float* trunk_out, const float* noise, int B, int A, float log_std_min,
                         float log_std_max, float* mu, float* pi, float* log_pi, float* log_std, float* tanh_ls,
                         float* pi_xa, int xa_ld, void* stream) {
  CURLA_REQUIRE(trunk_out);
  HeadArgs hd;
  const int rc = head_args(noise, B, A, log_std_min, log_std_max, mu, pi, log_pi, log_std, tanh_ls, pi_xa, xa_ld, &hd);
  if (rc != CURLA_OK) return rc;
  hipLaunchKernelGGL(actor_head_fwd_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     trunk_out, B, hd);
  return curla_launch_status();
}

int curla_actor_head_fwd_rng(const float* trunk_out, float* noise_out, unsigned long long seed,
                             unsigned long long offset, const unsigned long long* rng_dev, int B, int A,
                             float log_std_min, float log_std_max, float* mu, float* pi, float* log_pi, float* log_std,
                             float* tanh_ls, float* pi_xa, int xa_ld, void* stream) {
  CURLA_REQUIRE(trunk_out);
  HeadArgs hd;
  const int rc = head_args_rng(noise_out, seed, offset, rng_dev, B, A, log_std_min, log_std_max, mu, pi, log_pi, log_std, tanh_ls,
                               pi_xa, xa_ld, &hd);
  if (rc != CURLA_OK) return rc;
  hipLaunchKernelGGL(actor_head_fwd_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     trunk_out, B, hd);
  return curla_launch_status();
}

static int mlp_out_head_launch(const float* h, const float* W, const float* bias, float* trunk_out, int B, int A, int K,
                               const HeadArgs& hd, void* stream);

int curla_mlp_out_head_fwd_rng(const float* h, const float* W, const float* bias, float* trunk_out, int B, int A, int K,
                               float* noise_out, unsigned long long seed, unsigned long long offset,
                               const unsigned long long* rng_dev, float log_std_min, float log_std_max, float* mu,
                               float* pi, float* log_pi, float* log_std, float* tanh_ls, float* pi_xa, int xa_ld,
                               void* stream) {
  CURLA_REQUIRE(h && W && trunk_out && K > 0);
  HeadArgs hd;
  const int rc = head_args_rng(noise_out, seed, offset, rng_dev, B, A, log_std_min, log_std_max, mu, pi, log_pi, log_std, tanh_ls,
                               pi_xa, xa_ld, &hd);
  if (rc != CURLA_OK) return rc;
  return mlp_out_head_launch(h, W, bias, trunk_out, B, A, K, hd, stream);
}

int curla_mlp_out_head_fwd(const float* h, const float* W, const float* bias, float* trunk_out, int B, int A, int K,
                           const float* noise, float log_std_min, float log_std_max, float* mu, float* pi,
                           float* log_pi, float* log_std, float* tanh_ls, float* pi_xa, int xa_ld, void* stream) {
  CURLA_REQUIRE(h && W && trunk_out && K > 0);
  HeadArgs hd;
  const int rc = head_args(noise, B, A, log_std_min, log_std_max, mu, pi, log_pi, log_std, tanh_ls, pi_xa, xa_ld, &hd);
  if (rc != CURLA_OK) return rc;
  return mlp_out_head_launch(h, W, bias, trunk_out, B, A, K, hd, stream);
}

static int mlp_out_head_launch(const float* h, const float* W, const float* bias, float* trunk_out, int B, int A, int K,
                               const HeadArgs& hd, void* stream) {
  if (2 * A > kMaxOut || K % 4 != 0) return CURLA_ERR_UNSUPPORTED;
  CURLA_REQUIRE(aligned16(h) && aligned16(W));
  const size_t wbytes = (size_t)2 * A * K * sizeof(float);
  if (wbytes <= 48 * 1024)
    hipLaunchKernelGGL((mlp_out_fwd_kernel<true, 1, true>), dim3((B + 3) / 4, 1), dim3(256), wbytes,
                       static_cast<hipStream_t>(stream), h, 0LL, 0LL, W, 0LL, 0LL, bias, 0LL, 0LL, trunk_out, 0LL, 0LL,
                       B, 2 * A, K, 1, hd);
  else
    hipLaunchKernelGGL((mlp_out_fwd_kernel<true, 1, false>), dim3((B + 3) / 4, 1), dim3(256), 0,
                       static_cast<hipStream_t>(stream), h, 0LL, 0LL, W, 0LL, 0LL, bias, 0LL, 0LL, trunk_out, 0LL, 0LL,
                       B, 2 * A, K, 1, hd);
  return curla_launch_status();
}

int curla_actor_head_bwd(const float* gpi, const float* gpi2, int gpi_ld, const float* glp_rows, const double* log_alpha,
                         float glp_scale, const float* noise, const float* pi, const float* log_std,
                         const float* tanh_ls, int B, int A, float log_std_min, float log_std_max, float* dtrunk_out,
                         void* stream) {
  CURLA_REQUIRE(gpi && noise && pi && log_std && tanh_ls && dtrunk_out && B > 0 && A > 0 && A <= kMaxA && gpi_ld >= A);
  CURLA_REQUIRE(glp_rows || log_alpha);
  hipLaunchKernelGGL(actor_head_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), gpi,
                     gpi2, gpi_ld, glp_rows, log_alpha, glp_scale, noise, pi, log_std, tanh_ls, B, A, log_std_min,
                     log_std_max, dtrunk_out);
  return curla_launch_status();
}

int curla_concat(const float* z, const float* act, int B, int F, int A, float* xa, void* stream) {
  CURLA_REQUIRE(z && act && xa && B > 0 && F > 0 && A > 0);
  hipLaunchKernelGGL(concat_kernel, dim3((B * (F + A) + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), z,
                     act, B, F, A, xa);
  return curla_launch_status();
}

int curla_split_sum(const float* dxa, long long twin_stride, int B, int F, int A, float* dz, float* dact,
                    void* stream) {
  CURLA_REQUIRE(dxa && B > 0 && F > 0 && A > 0 && (dz || dact));
  hipLaunchKernelGGL(split_sum_kernel, dim3((B * (F + A) + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dxa, twin_stride, B, F, A, dz, dact);
  return curla_launch_status();
}

int curla_td_target(const float* tq, long long twin_stride, const float* log_pi, const float* reward,
                    const float* not_done, const double* log_alpha, float discount, int B, float* target_q,
                    void* stream) {
  CURLA_REQUIRE(tq && log_pi && reward && not_done && log_alpha && target_q && B > 0);
  hipLaunchKernelGGL(td_target_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), tq,
                     twin_stride, log_pi, reward, not_done, log_alpha, discount, B, target_q);
  return curla_launch_status();
}

int curla_critic_loss(const float* q, long long twin_stride, const float* target_q, int B, float* loss, float* dq,
                      void* stream) {
  CURLA_REQUIRE(q && target_q && loss && dq && B > 0);
  hipLaunchKernelGGL(critic_loss_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), q, twin_stride,
                     target_q, B, loss, dq);
  return curla_launch_status();
}

int curla_critic_td_loss(const float* q, const float* tq, long long twin_stride, const float* log_pi,
                         const float* reward, const float* not_done, const double* log_alpha, float discount, int B,
                         float* target_q, float* loss, float* dq, void* stream) {
  CURLA_REQUIRE(q && tq && log_pi && reward && not_done && log_alpha && target_q && loss && dq && B > 0);
  hipLaunchKernelGGL(critic_td_loss_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), q, tq, twin_stride,
                     log_pi, reward, not_done, log_alpha, discount, B, target_q, loss, dq);
  return curla_launch_status();
}

int curla_actor_loss(const float* q, long long twin_stride, const float* log_pi, const float* log_std, int A,
                     const double* log_alpha, float target_entropy, int B, float* scalars4, float* dq,
                     double* dlog_alpha, void* stream) {
  CURLA_REQUIRE(q && log_pi && log_std && log_alpha && scalars4 && dq && B > 0 && A > 0);
  hipLaunchKernelGGL(actor_loss_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), q, twin_stride, log_pi,
                     log_std, A, log_alpha, target_entropy, B, scalars4, dq, dlog_alpha);
  return curla_launch_status();
}

int curla_curl_ce(const float* logits, int B, int ld, float* row_loss, float* loss, float* dlogits, void* stream) {
  CURLA_REQUIRE(logits && row_loss && B > 0 && ld >= B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(curl_ce_kernel, dim3((B + 3) / 4), dim3(256), 0, st, logits, B, ld, row_loss, dlogits);
  if (loss) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_loss, B, loss);  // only when it is logged
  return curla_launch_status();
}

int curla_curl_head(const float* z_a, const float* z_pos, const float* wz, const float* xhat, const float* rstd,
                    const float* gamma, int B, int F, float* row_loss, float* loss, float* dfc, float* ln_partial,
                    int* nparts, float* w_partial, float* logits, float* dlogits, float* dz, void* stream) {
  CURLA_REQUIRE(z_a && z_pos && wz && xhat && rstd && gamma && row_loss && dfc && ln_partial && nparts && w_partial);
  // 128-row multiples (every wave of a block takes whole column tiles), at most 1024 rows (accumulator tiles per wave),
  // 49..52 features (phase A's k per lane group is compiled in: 13)
  if (B <= 0 || B % 128 != 0 || B > 1024 || F <= 4 * kCurlMaxKQ - 4 || F > 4 * kCurlMaxKQ) return CURLA_ERR_UNSUPPORTED;
  CurlHeadArgs a;
  a.za = z_a, a.zp = z_pos, a.wz = wz, a.xhat = xhat, a.rstd = rstd, a.gamma = gamma, a.row_loss = row_loss, a.dfc = dfc;
  a.ln_partial = ln_partial, a.w_partial = w_partial, a.logits = logits, a.dlogits = dlogits, a.dz = dz, a.B = B, a.F = F;
  const size_t lds = (size_t)(16 * (B + 4) + 16 * 64 + 2 * 16 * 64 + 2 * 8 * 16 + 16 + 8 * 3 * 64 + 2 * 4 * 16 * 64) * sizeof(float);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define CURLA_CURL_HEAD(NTILE)                                                                                  \
  case NTILE:                                                                                                    \
    if (curla_set_dyn_lds(reinterpret_cast<const void*>(curl_head_kernel<NTILE>), lds) != CURLA_OK) return CURLA_ERR_LAUNCH; \
    hipLaunchKernelGGL(curl_head_kernel<NTILE>, dim3(2 * (B / 16)), dim3(512), lds, st, a);                    \
    break
  switch (B / 128) {
    CURLA_CURL_HEAD(1);
    CURLA_CURL_HEAD(2);
    CURLA_CURL_HEAD(3);
    CURLA_CURL_HEAD(4);
    CURLA_CURL_HEAD(5);
    CURLA_CURL_HEAD(6);
    CURLA_CURL_HEAD(7);
    default: CURLA_CURL_HEAD(8);
  }
#undef CURLA_CURL_HEAD
  if (loss) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_loss, B, loss);  // only when it is logged
  *nparts = B / 16;
  return curla_launch_status();
}

// ---- a float64 scalar riding in a float32 all-reduce bucket ("pack into the actor bucket") ----------
// log_alpha's gradient is one float64 (curl_sac.py:397-404); as a collective of its own it is an 8-byte all-reduce per
// even update.  It travels instead as kF64Words float32 words at the end of the actor's gradient bucket: signed
// fixed-point digits of 20 bits, digit j with the weight 2^(kF64TopExp - 20 j).  Every float64 below 2^28 in magnitude
// whose last bit is worth at least 2^-132 is the EXACT sum of its digits, digit sums over up to 16 ranks stay below 2^24
// (exact in float32), and ncclAvg's division by a power-of-two world size is exact too -- so the receiver recovers the
// exact sum of the ranks' float64 gradients and rounds ONCE: for two ranks that is bit for bit today's float64
// all-reduce, (a + b) / 2.  (A world size that is not a power of two rounds each averaged digit to 24 bits: 2^-24 of
// the value -- the reference's own gradient is a float32 sum widened to float64.)
constexpr int kF64Words = 8;
constexpr int kF64TopExp = 8;

__global__ void f64_pack_kernel(const double* __restrict__ v, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  double r = *v;
  if (!(fabs(r) < 0x1p28)) {  // out of the fixed-point range, inf or nan: the top word alone carries it, inexactly
    out[0] = (float)(r * 0x1p-8);
    for (int j = 1; j < kF64Words; ++j) out[j] = 0.f;
    return;
  }
  for (int j = 0; j < kF64Words; ++j) {  // (every operation here is exact in double)
    const double p = trunc(ldexp(r, 20 * j - kF64TopExp));
    r -= ldexp(p, kF64TopExp - 20 * j);
    out[j] = (float)p;
  }
}

struct U192 {
  unsigned long long w2, w1, w0;
};
__device__ __forceinline__ bool u192_bit(const U192& a, int i) {
  return ((i >= 128 ? a.w2 : i >= 64 ? a.w1 : a.w0) >> (i & 63)) & 1ull;
}
__device__ __forceinline__ bool u192_any_below(const U192& a, int i) {  // any set bit at a position < i
  if (i <= 0) return false;
  if (i >= 192) return (a.w0 | a.w1 | a.w2) != 0;
  const unsigned long long mask = (i & 63) ? ((1ull << (i & 63)) - 1ull) : 0ull;
  if (i >= 128) return (a.w0 | a.w1 | (a.w2 & mask)) != 0;
  if (i >= 64) return (a.w0 | (a.w1 & mask)) != 0;
  return (a.w0 & mask) != 0;
}
__device__ __forceinline__ unsigned long long u192_shr_low(const U192& a, int s) {  // low 64 bits of a >> s
  const int q = s >> 6, r = s & 63;
  const unsigned long long lo = q == 0 ? a.w0 : q == 1 ? a.w1 : q == 2 ? a.w2 : 0ull;
  const unsigned long long hi = q == 0 ? a.w1 : q == 1 ? a.w2 : 0ull;
  return r ? (lo >> r) | (hi << (64 - r)) : lo;
}

// words: the (averaged or summed) digits; n_mul undoes an average (digit sums are integers again), n_div turns the
// sum into the mean: out = RN(sum over ranks) / n_div.
__global__ void f64_unpack_kernel(const float* __restrict__ words, double n_mul, double n_div, double* __restrict__ out) {
  if (threadIdx.x != 0) return;
  long long d[kF64Words];
  bool ok = true;
  for (int j = 0; j < kF64Words; ++j) {
    const double x = (double)words[j] * n_mul;
    if (!(fabs(x) < 0x1p40)) ok = false;
    d[j] = ok ? llrint(x) : 0;
  }
  if (!ok) {  // an out-of-range / non-finite gradient somewhere: plain double arithmetic, no exactness claimed
    double s = 0.;
    for (int j = kF64Words - 1; j >= 0; --j) s += ldexp((double)words[j] * n_mul, kF64TopExp - 20 * j);
    *out = s / n_div;
    return;
  }
  // carries up the digits: d[1..] in [0, 2^20), d[0] keeps the sign; a negative value is negated and carried again
  for (int pass = 0; pass < 2; ++pass) {
    for (int j = kF64Words - 1; j >= 1; --j) {
      const long long c = d[j] >> 20;  // (arithmetic shift: floor)
      d[j] -= c << 20;
      d[j - 1] += c;
    }
    if (pass == 1 || d[0] >= 0) break;
    for (int j = 0; j < kF64Words; ++j) d[j] = -d[j];
    ok = false;  // (re-used: "the value is negative")
  }
  const bool neg = !ok;
  U192 m = {0ull, 0ull, 0ull};  // the magnitude in units of 2^(kF64TopExp - 20 (kF64Words - 1))
  for (int j = 0; j < kF64Words; ++j) {
    m.w2 = (m.w2 << 20) | (m.w1 >> 44);
    m.w1 = (m.w1 << 20) | (m.w0 >> 44);
    m.w0 = (m.w0 << 20) | (unsigned long long)d[j];  // (d[0] < 2^41 goes in first: nothing overlaps)
  }
  const int msb = m.w2 ? 191 - __clzll(m.w2) : m.w1 ? 127 - __clzll(m.w1) : m.w0 ? 63 - __clzll(m.w0) : -1;
  double r = 0.;
  if (msb >= 0) {
    const int unit = kF64TopExp - 20 * (kF64Words - 1);
    if (msb <= 52) {
      r = ldexp((double)m.w0, unit);
    } else {  // 53 bits, round to nearest even on everything below them
      const int s = msb - 52;
      unsigned long long mant = u192_shr_low(m, s) & ((1ull << 53) - 1ull);
      const bool half = u192_bit(m, s - 1), sticky = u192_any_below(m, s - 1);
      if (half && (sticky || (mant & 1ull))) mant += 1ull;
      r = ldexp((double)mant, unit + s);
    }
  }
  *out = (neg ? -r : r) / n_div;
}

int curla_f64_pack(const double* value, float* words, void* stream) {
  CURLA_REQUIRE(value && words && (reinterpret_cast<uintptr_t>(value) & 7) == 0 && (reinterpret_cast<uintptr_t>(words) & 3) == 0);
  hipLaunchKernelGGL(f64_pack_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), value, words);
  return curla_launch_status();
}

int curla_f64_unpack(const float* words, double n_mul, double n_div, double* value, void* stream) {
  CURLA_REQUIRE(value && words && n_mul >= 1. && n_div >= 1. && (reinterpret_cast<uintptr_t>(value) & 7) == 0 &&
                (reinterpret_cast<uintptr_t>(words) & 3) == 0);
  hipLaunchKernelGGL(f64_unpack_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), words, n_mul, n_div, value);
  return curla_launch_status();
}

int curla_mean(const float* x, int n, float* out, void* stream) {
  CURLA_REQUIRE(x && out && n > 0);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, out);
  return curla_launch_status();
}

int curla_soft_update(const float* param, float* target, size_t n, float tau, float one_minus_tau, void* stream) {
  CURLA_REQUIRE(param && target && n > 0);
  hipLaunchKernelGGL(soft_update_kernel, dim3(nblocks(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), param,
                     target, n, tau, one_minus_tau);
  return curla_launch_status();
}

int curla_soft_update2(const float* param, float* target, size_t n, size_t split, float tau_a, float one_minus_tau_a,
                       float tau_b, float one_minus_tau_b, void* stream) {
  CURLA_REQUIRE(param && target && n > 0 && split <= n);
  hipLaunchKernelGGL(soft_update2_kernel, dim3(nblocks(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), param,
                     target, n, split, tau_a, one_minus_tau_a, tau_b, one_minus_tau_b);
  return curla_launch_status();
}

static int adam_step_launch(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                            double beta1, double beta2, double eps, long long step, const float* dyn,
                            const AdamScalar64& sc, void* stream, const AdamLerp& lp = AdamLerp{}) {
  CURLA_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1 && beta1 >= 0. && beta1 < 1. &&
                beta2 >= 0. && beta2 < 1. && (reinterpret_cast<uintptr_t>(dyn) & 3) == 0);
  const double b1 = beta1, b2 = beta2;
  const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  const int vec = aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq) &&
                  (!lp.tgt || (aligned16(lp.tgt) && lp.split % 4 == 0));
  hipLaunchKernelGGL(adam_step_kernel, dim3(nblocks((n + 3) / 4, 256, 8192)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), param, grad, exp_avg, exp_avg_sq, n, vec, (float)(1.0 - b1),
                     (float)b2, (float)(1.0 - b2), (float)(lr / bc1), (float)sqrt(bc2), (float)eps, sc, lp, dyn);
  return curla_launch_status();
}

int curla_adam_step_lerp(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                         double beta1, double beta2, double eps, long long step, const float* dyn, float* target,
                         size_t split, float tau_a, float one_minus_tau_a, float tau_b, float one_minus_tau_b,
                         void* stream) {
  CURLA_REQUIRE(target && split <= n);
  AdamScalar64 none = {};
  AdamLerp lp{target, split, tau_a, one_minus_tau_a, tau_b, one_minus_tau_b};
  return adam_step_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, dyn, none, stream, lp);
}

int curla_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                    double beta1, double beta2, double eps, long long step, const float* dyn, void* stream) {
  AdamScalar64 none = {};
  return adam_step_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, dyn, none, stream);
}

int curla_adam_step_scalar64(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                             double beta1, double beta2, double eps, long long step, const float* dyn, double* param64,
                             const double* grad64, double* exp_avg64, double* exp_avg_sq64, double lr64, double beta1_64,
                             double beta2_64, double eps64, long long step64, const double* dyn64, void* stream) {
  CURLA_REQUIRE(param64 && grad64 && exp_avg64 && exp_avg_sq64 && step64 >= 1 && beta1_64 >= 0. && beta1_64 < 1. &&
                beta2_64 >= 0. && beta2_64 < 1. && (reinterpret_cast<uintptr_t>(dyn64) & 7) == 0);
  AdamScalar64 sc;
  sc.p = param64, sc.g = grad64, sc.m = exp_avg64, sc.v = exp_avg_sq64;
  sc.w1 = 1.0 - beta1_64, sc.b2 = beta2_64, sc.w2 = 1.0 - beta2_64;
  sc.step_size = lr64 / (1.0 - pow(beta1_64, (double)step64));
  sc.bc2_sqrt = sqrt(1.0 - pow(beta2_64, (double)step64)), sc.eps = eps64, sc.dyn = dyn64;
  return adam_step_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, dyn, sc, stream);
}

static AdamHyper adam_hyper(double lr, double beta1, double beta2, double eps, long long step) {
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  AdamHyper h;
  h.w1 = (float)(1.0 - beta1), h.b2 = (float)beta2, h.w2 = (float)(1.0 - beta2);
  h.step_size = (float)(lr / bc1), h.bc2_sqrt = (float)sqrt(bc2), h.eps = (float)eps;
  return h;
}

int curla_adam_step2(float* param, const float* grad, float* exp_avg1, float* exp_avg_sq1, float* exp_avg2,
                     float* exp_avg_sq2, size_t n, size_t n_pre, double lr1, double beta1_1, double beta2_1, double eps1,
                     long long step1, double lr2, double beta1_2, double beta2_2, double eps2, long long step2,
                     const float* dyn, void* stream) {
  CURLA_REQUIRE(param && grad && exp_avg1 && exp_avg_sq1 && exp_avg2 && exp_avg_sq2 && n > 0 && n_pre <= n &&
                step1 >= 1 && step2 >= 1 && (reinterpret_cast<uintptr_t>(dyn) & 3) == 0);
  CURLA_REQUIRE(beta1_1 >= 0. && beta1_1 < 1. && beta2_1 >= 0. && beta2_1 < 1. && beta1_2 >= 0. && beta1_2 < 1. &&
                beta2_2 >= 0. && beta2_2 < 1.);
  const int vec = (n_pre % 4 == 0) && aligned16(param) && aligned16(grad) && aligned16(exp_avg1) &&
                  aligned16(exp_avg_sq1) && aligned16(exp_avg2) && aligned16(exp_avg_sq2);
  hipLaunchKernelGGL(adam_step2_kernel, dim3(nblocks((n + 3) / 4, 256, 8192)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), param, grad, exp_avg1, exp_avg_sq1, exp_avg2, exp_avg_sq2, n, n_pre, vec,
                     adam_hyper(lr1, beta1_1, beta2_1, eps1, step1), adam_hyper(lr2, beta1_2, beta2_2, eps2, step2), dyn);
  return curla_launch_status();
}

int curla_gather_transition_scalars(const float* scalars, const int64_t* idx, int B, int A, float* action, float* reward,
                                    float* not_done, void* stream) {
  CURLA_REQUIRE(scalars && idx && action && reward && not_done && B > 0 && A > 0);
  hipLaunchKernelGGL(gather_transition_scalars_kernel, dim3((B * (A + 2) + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), scalars, idx, B, A, action, reward, not_done);
  return curla_launch_status();
}

int curla_host_device_pointer(void* host, void** device) {
  CURLA_REQUIRE(host && device);
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess || !d) {
    (void)hipGetLastError();
    return CURLA_ERR_ARG;
  }
  *device = d;
  return CURLA_OK;
}

int curla_sample_stage(const void* host_block, void* device_block, long long nbytes, const float* scalars, int B, int A,
                       float* action, float* reward, float* not_done, void* stream) {
  CURLA_REQUIRE(host_block && device_block && scalars && action && reward && not_done && B > 0 && A > 0);
  CURLA_REQUIRE(nbytes >= (long long)B * 8 && nbytes % 8 == 0 && nbytes < (1LL << 30));
  CURLA_REQUIRE(((uintptr_t)host_block | (uintptr_t)device_block) % 8 == 0);
  const int nwords = (int)(nbytes / 8);
  const int n = nwords > B * (A + 2) ? nwords : B * (A + 2);
  hipLaunchKernelGGL(sample_stage_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned long long*>(host_block), static_cast<unsigned long long*>(device_block),
                     nwords, scalars, B, A, action, reward, not_done);
  return curla_launch_status();
}

int curla_crop_nchw(const uint8_t* frames, const int64_t* idx, const int32_t* h1, const int32_t* w1, int B, int C,
                    int Hs, int Ws, int Hc, int Wc, float* out_f32, uint8_t* out_u8, void* stream) {
  CURLA_REQUIRE(frames && (out_f32 || out_u8) && B > 0 && C > 0 && Hc > 0 && Wc > 0 && Hs >= Hc && Ws >= Wc);
  hipLaunchKernelGGL(crop_nchw_kernel, dim3(nblocks((size_t)B * C * Hc * Wc, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), frames, idx, h1, w1, B, C, Hs, Ws, Hc, Wc, out_f32, out_u8);
  return curla_launch_status();
}

int curla_store_frame(const uint8_t* chw, uint8_t* frames, long long slot, int C, int H, int W, void* stream) {
  CURLA_REQUIRE(chw && frames && slot >= 0 && C > 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(store_frame_kernel, dim3(nblocks((size_t)C * H * W, 256, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), chw, frames, (int64_t)slot, C, H, W);
  return curla_launch_status();
}

int curla_gather_stacks(const uint8_t* store, const int32_t* fid, int fid_stride, const int64_t* idx, int B, int K,
                        int H, int W, uint8_t* out, void* stream) {
  CURLA_REQUIRE(store && fid && out && B > 0 && K > 0 && H > 0 && W > 0 && fid_stride >= K);
  const int HW = H * W;
  const size_t n = (size_t)B * ((HW + 3) / 4);
  const bool fast = K <= 4 && HW % 4 == 0 && (reinterpret_cast<uintptr_t>(store) & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 3) == 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (fast)
    hipLaunchKernelGGL(gather_stacks_kernel<true>, dim3(nblocks(n, 256, 8192)), dim3(256), 0, st, store, fid, fid_stride,
                       idx, B, K, HW, out);
  else
    hipLaunchKernelGGL(gather_stacks_kernel<false>, dim3(nblocks(n, 256, 8192)), dim3(256), 0, st, store, fid,
                       fid_stride, idx, B, K, HW, out);
  return curla_launch_status();
}

int curla_nhwc_to_nchw(const float* in, float* out, int B, int H, int W, int C, void* stream) {
  CURLA_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblocks((size_t)B * H * W * C, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out, B, H, W, C);
  return curla_launch_status();
}

const char* curla_version(void) { return "curla_hip 0.6 (gfx950, abi 6)"; }

int curla_abi_version(void) { return CURLA_ABI_VERSION; }

}  // extern "C"
