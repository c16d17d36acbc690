// Batched fp32 GEMM on the exact-f32 matrix pipe (v_mfma_f32_16x16x4_f32) with
// fused epilogues.  Serves every dense layer of the path: encoder fc
// (encoder.py:66,98) forward/backward, the actor trunk and twin-Q MLPs
// (curl_sac.py:70-74,129-133) and the CURL bilinear logits
// z_a (W z_pos^T) (curl_sac.py:219-220) with their backward products.
//
//   C[z][m][n] = epilogue( alpha * sum_k opA(A)[m][k] * opB(B)[n][k] )
//   opA: row-major A[m*lda + k]   or k-major A[k*lda + m]   (transposed operand)
//   opB: row-major B[n*ldb + k]   or k-major B[k*ldb + n]
//   epilogue: + bias[n], ReLU, zero where mask[m][n] <= 0 (ReLU backward).
// grid.z enumerates batch x split-K; a split writes its partial product to
// C + split * split_stride and the caller reduces (deterministic order).
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "options.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int RSTR = BK + 8;   // LDS stride, row-major tile [64][BK]: 40 floats -> 16-B aligned rows, ds_read_b128 of 16 rows x 4 k-quarters conflict-free
#ifndef CURLA_GEMM_KSTR
#define CURLA_GEMM_KSTR (BM + 4)
#endif
// LDS stride, k-major tile [BK][64].  A fragment read is k = 16j + 4kq + e of row li: bank = (k * KSTR + li) mod 32, and
// with KSTR = 68 (= 4 mod 32) that is (16 kq + 4 e + li) mod 32 -- the two lane groups of a half-wave (kq, kq + 1) take
// disjoint halves of the banks.  (80, the stride of rounds 1-2, put both on the same 16 banks: SQ_LDS_BANK_CONFLICT 0.3-0.4
// of the LDS cycles of the kernels with a k-major operand.)
constexpr int KSTR = CURLA_GEMM_KSTR;
// bf16x3 form (round 5, gemm_tile<..., B3 = true>): an operand tile is THREE bf16 images [part][row][32 k], rows of 64
// bytes = four 16-byte chunks (a lane's fragment is one chunk, a staging thread's four k half a chunk), chunk c of row r at
// slot c ^ b3_swz(r).  With that XOR the four 16-lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS: {0-3, 12-15,
// 20-27}, ...) each read 16 different slots of the 256-byte bank row, and the 8-byte stores of a row-major operand (16
// lanes = 2 rows x 8 pieces) cover all 32 banks once; a k-major operand's stores (16 rows, one piece) are 2-way.  (The
// first two versions padded rows to 80 and 96 bytes: fragment reads 2-way at 80, stores 2- and 4-way at 96 -- PMC
// SQ_LDS_BANK_CONFLICT 0.30 / 0.50 of the LDS cycles of the 128 x 64 kernels.)
constexpr int kB3RowBytes = 64;
__device__ __forceinline__ int b3_swz(int row) { return (-((row >> 2) & 3)) & 3; }
constexpr int b3_part_bytes(int rows) { return rows * kB3RowBytes; }
constexpr int b3_tile_floats(int rows) { return 3 * b3_part_bytes(rows) / 4; }
constexpr int kGemmTileFloatsF32 = (BK * KSTR > BM * RSTR) ? BK * KSTR : BM * RSTR;
// floats of one operand tile buffer with `rows` rows (the f32 forms exist for up to 64 rows only)
constexpr int gemm_tile_floats(int rows) {
  return rows > 64 ? b3_tile_floats(rows) : (b3_tile_floats(64) > kGemmTileFloatsF32 ? b3_tile_floats(64) : kGemmTileFloatsF32);
}

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* mask;
  int M, N, K, lda, ldb, ldc, ldmask;
  long long sA, sB, sC, sBias, sMask, sSplit;
  int nbatch, ksplit, kchunk;
  float alpha;
  int relu, vecA, vecB;
  int stream_c;  // output far larger than the L2s (fc data gradient): streaming stores
  // two-level batch: item = outer * nb_inner + inner at base + inner * s + outer * s2 (e.g. the twin Q functions of
  // the target critic and of the critic: twins a block apart inside a flat buffer, the two flat buffers wherever
  // the allocator put them -- one launch for all four)
  int nb_inner;
  long long sA2, sB2, sC2, sBias2, sMask2;
  // colsum (small-output kernel only): also colsum[z][m] = alpha * sum_k opA[m][k] (a weight gradient dy^T x with the
  // bias gradient, the column sums of dy, from the operands it loads anyway)
  float* colsum;
  long long sColsum, sColsum2;
  // nptr > 0: the batch items are unrelated problems of one shape, given by pointer (A = Ap[batch] ...) instead of by
  // stride -- e.g. the fc products of three encoders on three activation tensors in one launch
  int nptr;
  const float* Ap[4];
  const float* Bp[4];
  float* Cp[4];
};

// load one BK x ROWS operand tile into registers (ROWS/32 float4 per thread)
template <bool KMAJOR, int ROWS>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int ld, int rows_total, int r0, int k0, int kend,
                                          int vec, int tid, f32x4 (&reg)[ROWS / 32]) {
#pragma unroll
  for (int u = 0; u < ROWS / 32; ++u) {
    f32x4 v = {0, 0, 0, 0};
    if (!KMAJOR) {
      const int row = (tid >> 3) + 32 * u, k4 = (tid & 7) * 4;
      const int gr = r0 + row, gk = k0 + k4;
      if (gr < rows_total) {
        const float* p = P + (size_t)gr * ld + gk;
        if (vec && gk + 3 < kend) {
          v = *reinterpret_cast<const f32x4*>(p);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (gk + i < kend) v[i] = p[i];
        }
      }
    } else {
      constexpr int TPR = ROWS / 4;  // threads per k-row
      const int k = tid / TPR + (256 / TPR) * u, r4 = (tid % TPR) * 4;
      const int gk = k0 + k, gr = r0 + r4;
      if (gk < kend) {
        const float* p = P + (size_t)gk * ld + gr;
        if (vec && gr + 3 < rows_total) {
          v = *reinterpret_cast<const f32x4*>(p);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (gr + i < rows_total) v[i] = p[i];
        }
      }
    }
    reg[u] = v;
  }
}

// interior, aligned tiles: no bounds or alignment tests in the k loop
template <bool KMAJOR, int ROWS>
__device__ __forceinline__ void tile_load_fast(const float* __restrict__ P, int ld, int rows_total, int r0, int k0,
                                               int tid, f32x4 (&reg)[ROWS / 32]) {
#pragma unroll
  for (int u = 0; u < ROWS / 32; ++u) {
    if (!KMAJOR) {
      // rows past the matrix edge re-read the last row (their products are discarded by the epilogue)
      const int row = min(r0 + (tid >> 3) + 32 * u, rows_total - 1), k4 = (tid & 7) * 4;
      reg[u] = *reinterpret_cast<const f32x4*>(P + (size_t)row * ld + k0 + k4);
    } else {
      constexpr int TPR = ROWS / 4;
      const int k = tid / TPR + (256 / TPR) * u, r4 = (tid % TPR) * 4;
      reg[u] = *reinterpret_cast<const f32x4*>(P + (size_t)(k0 + k) * ld + r0 + r4);
    }
  }
}

template <bool KMAJOR, int ROWS>
__device__ __forceinline__ void tile_store(float* __restrict__ S, int tid, const f32x4 (&reg)[ROWS / 32]) {
#pragma unroll
  for (int u = 0; u < ROWS / 32; ++u) {
    if (!KMAJOR) {
      const int row = (tid >> 3) + 32 * u, k4 = (tid & 7) * 4;
      *reinterpret_cast<f32x4*>(S + row * RSTR + k4) = reg[u];
    } else {
      constexpr int TPR = ROWS / 4;
      const int k = tid / TPR + (256 / TPR) * u, r4 = (tid % TPR) * 4;
      *reinterpret_cast<f32x4*>(S + k * KSTR + r4) = reg[u];
    }
  }
}

// MFMA operand values of one lane for the 8 k-steps of a BK=32 tile.  k-step (j, e) multiplies
// k = 16j + 4kq + e on lane group kq (any k order is fine as long as A and B agree): a row-major
// tile delivers them with two ds_read_b128, a k-major tile with eight ds_read_b32.
template <bool KMAJOR>
__device__ __forceinline__ void frags(const float* __restrict__ S, int row, int kq, float (&v)[8]) {
  if (!KMAJOR) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(S + row * RSTR + 4 * kq);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(S + row * RSTR + 16 + 4 * kq);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = lo[e], v[4 + e] = hi[e];
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * j + e] = S[(16 * j + 4 * kq + e) * KSTR + row];
  }
}

// ---- bf16x3 operands (the heads' 512 x 1024 x 1024 products ran at 0.6-0.7 of the f32-input MFMA: a barrier per 32-deep
// k tile with 32 matrix instructions of 32 cycles between two of them).  Every fp32 operand element is split ONCE, when
// its tile is staged into LDS, into three bf16 parts (x = xh + xm + xl exactly, conv_rwb.h); a k tile is then one k-step
// of v_mfma_f32_16x16x32_bf16 per term, six terms per fp32 product: 24 matrix instructions of 16 cycles per wave and
// tile instead of 32 of 32, and the split costs 5.5 VALU instructions per element against the 2 x 16 .. 4 x 16
// products the element takes part in.  Both operand kinds end in the same image: a row-major operand is loaded as
// float4 along k, a k-major one as four dword loads along k per thread (a wave reads 64 consecutive rows of one k: 256
// contiguous bytes), so a thread always holds four consecutive k of one row.
typedef short gbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned gu32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gbf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned g_cvt_pk_bf16(float x0, float x1) {
  const f32x2 v = {x0, x1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gbf16x2));
}

template <bool KMAJOR, int ROWS, int NT = 256>
__device__ __forceinline__ void tile_load_b3(const float* __restrict__ P, int ld, int rows_total, int r0, int k0, int tid,
                                             f32x4 (&reg)[ROWS * 8 / NT]) {
#pragma unroll
  for (int u = 0; u < ROWS * 8 / NT; ++u) {
    if (!KMAJOR) {
      const int row = min(r0 + (tid >> 3) + (NT / 8) * u, rows_total - 1), k4 = (tid & 7) * 4;
      reg[u] = *reinterpret_cast<const f32x4*>(P + (size_t)row * ld + k0 + k4);
    } else {
      const int row = tid % ROWS, k4 = 4 * (tid / ROWS + (NT / ROWS) * u);
      const float* p = P + (size_t)(k0 + k4) * ld + r0 + row;
#pragma unroll
      for (int e = 0; e < 4; ++e) reg[u][e] = p[(size_t)e * ld];
    }
  }
}

template <bool KMAJOR, int ROWS, int NT = 256>
__device__ __forceinline__ void tile_store_b3(float* __restrict__ S, int tid, const f32x4 (&reg)[ROWS * 8 / NT]) {
  char* base = reinterpret_cast<char*>(S);
#pragma unroll
  for (int u = 0; u < ROWS * 8 / NT; ++u) {
    const int row = KMAJOR ? tid % ROWS : (tid >> 3) + (NT / 8) * u;
    const int k4 = KMAJOR ? 4 * (tid / ROWS + (NT / ROWS) * u) : (tid & 7) * 4;
    gu32x2 h, m, l;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float x0 = reg[u][2 * q], x1 = reg[u][2 * q + 1];
      const unsigned hh = g_cvt_pk_bf16(x0, x1);
      const float r0 = x0 - __builtin_bit_cast(float, hh << 16), r1 = x1 - __builtin_bit_cast(float, hh & 0xFFFF0000u);
      const unsigned mm = g_cvt_pk_bf16(r0, r1);
      const float s0 = r0 - __builtin_bit_cast(float, mm << 16), s1 = r1 - __builtin_bit_cast(float, mm & 0xFFFF0000u);
      h[q] = hh, m[q] = mm, l[q] = g_cvt_pk_bf16(s0, s1);
    }
    char* p = base + row * kB3RowBytes + (((k4 >> 3) ^ b3_swz(row)) << 4) + ((k4 & 4) << 1);
    *reinterpret_cast<gu32x2*>(p) = h;
    *reinterpret_cast<gu32x2*>(p + b3_part_bytes(ROWS)) = m;
    *reinterpret_cast<gu32x2*>(p + 2 * b3_part_bytes(ROWS)) = l;
  }
}

struct Frag3 {
  gu32x4 h, m, l;
};

template <int ROWS>
__device__ __forceinline__ Frag3 frags_b3(const float* __restrict__ S, int row, int kq) {
  const char* p = reinterpret_cast<const char*>(S) + row * kB3RowBytes + ((kq ^ b3_swz(row)) << 4);
  Frag3 f;
  f.h = *reinterpret_cast<const gu32x4*>(p);
  f.m = *reinterpret_cast<const gu32x4*>(p + b3_part_bytes(ROWS));
  f.l = *reinterpret_cast<const gu32x4*>(p + 2 * b3_part_bytes(ROWS));
  return f;
}

// eight fp32 values (a lane's 8 consecutive k of one row) -> the three bf16 operands (x = h + m + l exactly)
__device__ __forceinline__ Frag3 g_split8(const f32x4 lo, const f32x4 hi) {
  Frag3 o;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = p < 2 ? lo[2 * p] : hi[2 * p - 4], x1 = p < 2 ? lo[2 * p + 1] : hi[2 * p - 3];
    const unsigned hh = g_cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, hh << 16), r1 = x1 - __builtin_bit_cast(float, hh & 0xFFFF0000u);
    const unsigned mm = g_cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, mm << 16), s1 = r1 - __builtin_bit_cast(float, mm & 0xFFFF0000u);
    o.h[p] = hh, o.m[p] = mm, o.l[p] = g_cvt_pk_bf16(s0, s1);
  }
  return o;
}

__device__ __forceinline__ f32x4 g_mfma_bf16(const gu32x4 a, const gu32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gbf16x8, a), __builtin_bit_cast(gbf16x8, b), c, 0, 0, 0);
}

// NT = 256 threads = 2 x 2 waves, a wave computes (TBM/2) x (TBN/2) of the TBM x TBN tile; NT = 512 (the 128 x 64 bf16x3
// tile): 4 x 2 waves of 32 x 32 -- two waves per SIMD (with one wave per SIMD, 64 x 32 each, the k tile's phases ran
// back to back: 37 against 32 us for the twin critics' hidden layer).
// TBN = 32 doubles the workgroup count for the mid-sized GEMMs of the heads
// (512 x 1024 x 1024 is only 128 tiles of 64 x 64 on a 256-CU chip).
// The k loop is double-buffered in LDS (two operand tile pairs, 40 KB): while the waves multiply tile t out of one
// buffer, tile t+1 goes from registers into the other and tile t+2 is in flight from HBM/L2 -- ONE barrier per k tile,
// and the LDS write -> barrier -> fragment read latency of the next tile sits under the current tile's MFMAs.
template <int ROWS>
using GemmLds = float[2][gemm_tile_floats(ROWS)];

// one TBM x TBN output tile (tile column bx, tile row by, batch x split item z) by the 256 threads of a workgroup
template <bool AK, bool BKM, int TBM, int TBN, bool FAST, bool B3 = false, int NT = 256>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int bx, const int by, const int z, float (*As)[gemm_tile_floats(TBM)],
                                          float (*Bs)[gemm_tile_floats(TBN)]) {
  static_assert(!B3 || FAST, "the bf16x3 form takes interior, aligned tiles only");
  static_assert(TBM <= 64 || B3, "tiles of more than 64 rows exist in the bf16x3 form only");
  static_assert(NT == 256 || B3, "the f32 forms are written for 256 threads");
  constexpr int WR = NT / 128;        // rows of waves (two columns of waves always)
  constexpr int WM = TBM / WR;        // rows per wave
  constexpr int MI = WM / 16;         // 16-row fragments per wave
  constexpr int RA = TBM * 8 / NT, RB = TBN * 8 / NT;  // staging registers (float4) per thread and operand
  constexpr int NJ = TBN / 32;
  constexpr int WN = TBN / 2;  // columns per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int batch = z / g.ksplit, ks = z - batch * g.ksplit;
  const int m0 = by * TBM, n0 = bx * TBN;
  const int bo = batch / g.nb_inner, bi = batch - bo * g.nb_inner;
  const float* A = g.nptr ? g.Ap[batch] : g.A + bi * g.sA + bo * g.sA2;
  const float* B = g.nptr ? g.Bp[batch] : g.B + bi * g.sB + bo * g.sB2;
  float* C = (g.nptr ? g.Cp[batch] : g.C + bi * g.sC + bo * g.sC2) + ks * g.sSplit;
  const int kbeg = ks * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);

  const bool epi = g.ksplit == 1;
  const float* bias = (epi && g.bias) ? g.bias + bi * g.sBias + bo * g.sBias2 : nullptr;
  const float* mask = (epi && g.mask) ? g.mask + bi * g.sMask + bo * g.sMask2 : nullptr;
  // The MFMA is issued with the N-side operand as its row operand, so a lane holds 4 CONSECUTIVE n
  // (rows 4kq..4kq+3 of the 16x16 tile) of ONE m (column li): C, bias and mask move as float4.
  const bool vec_c = (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                     (!mask || ((g.ldmask % 4 == 0) && ((reinterpret_cast<uintptr_t>(mask) & 15) == 0)));
  // the ReLU mask of a data-gradient product (as large as C itself for the fc layer) is requested before the k loop
  // and arrives under it, instead of being waited for between the last MFMA and the stores
  f32x4 mk[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      mk[i][j] = f32x4{1.f, 1.f, 1.f, 1.f};
      const int m = m0 + wm * WM + i * 16 + li;
      const int n = n0 + wn * WN + j * 16 + 4 * kq;
      if (mask && vec_c && m < g.M && n + 3 < g.N) mk[i][j] = *reinterpret_cast<const f32x4*>(mask + (size_t)m * g.ldmask + n);
    }

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  auto load = [&](int k0, f32x4 (&ra)[RA], f32x4 (&rb)[RB]) {
    if constexpr (B3) {
      tile_load_b3<AK, TBM, NT>(A, g.lda, g.M, m0, k0, tid, ra);
      tile_load_b3<BKM, TBN, NT>(B, g.ldb, g.N, n0, k0, tid, rb);
    } else if constexpr (FAST) {
      tile_load_fast<AK, TBM>(A, g.lda, g.M, m0, k0, tid, ra);
      tile_load_fast<BKM, TBN>(B, g.ldb, g.N, n0, k0, tid, rb);
    } else {
      tile_load<AK, TBM>(A, g.lda, g.M, m0, k0, kend, g.vecA, tid, ra);
      tile_load<BKM, TBN>(B, g.ldb, g.N, n0, k0, kend, g.vecB, tid, rb);
    }
  };
  auto stage = [&](int buf, const f32x4 (&ra)[RA], const f32x4 (&rb)[RB]) {
    if constexpr (B3) {
      tile_store_b3<AK, TBM, NT>(As[buf], tid, ra);
      tile_store_b3<BKM, TBN, NT>(Bs[buf], tid, rb);
    } else {
      tile_store<AK, TBM>(As[buf], tid, ra);
      tile_store<BKM, TBN>(Bs[buf], tid, rb);
    }
  };
  auto multiply = [&](int buf) {
    if constexpr (B3) {
      Frag3 xa[MI], xb[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) xa[i] = frags_b3<TBM>(As[buf], wm * WM + i * 16 + li, kq);
#pragma unroll
      for (int j = 0; j < NJ; ++j) xb[j] = frags_b3<TBN>(Bs[buf], wn * WN + j * 16 + li, kq);
      // six terms per product, smallest first; the (i, j) accumulators interleaved term by term
#pragma unroll
      for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const gu32x4& pb = term == 0 ? xb[j].l : (term == 2 || term == 3) ? xb[j].m : xb[j].h;
            const gu32x4& pa = (term == 0 || term == 3 || term == 5) ? xa[i].h : (term == 1) ? xa[i].l : xa[i].m;
            acc[i][j] = g_mfma_bf16(pb, pa, acc[i][j]);
          }
    } else {
    float fa[MI][8], fb[NJ][8];
#pragma unroll
    for (int i = 0; i < MI; ++i) frags<AK>(As[buf], wm * WM + i * 16 + li, kq, fa[i]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) frags<BKM>(Bs[buf], wn * WN + j * 16 + li, kq, fb[j]);
#pragma unroll
    for (int s = 0; s < BK / 4; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16(fb[j][s], fa[i][s], acc[i][j]);
    }
  };
  const int ntiles = (kend - kbeg + BK - 1) / BK;
  if (ntiles > 0) {
    f32x4 ra0[RA], rb0[RB], ra1[RA], rb1[RB];
    load(kbeg, ra0, rb0);
    load(ntiles > 1 ? kbeg + BK : kbeg, ra1, rb1);  // (a single-tile product re-reads its tile: keeps this branch-free)
    stage(0, ra0, rb0);
    __syncthreads();
    int k0 = kbeg, t = 0;
    // steady state: tile t in buffer 0, tile t+1 in the second register set; both refills unconditional so that the
    // compiler's vmcnt bookkeeping keeps the younger tile's loads in flight across each LDS store
    for (; t + 3 < ntiles; t += 2, k0 += 2 * BK) {
      load(k0 + 2 * BK, ra0, rb0);
      multiply(0);
      stage(1, ra1, rb1);
      __syncthreads();
      load(k0 + 3 * BK, ra1, rb1);
      multiply(1);
      stage(0, ra0, rb0);
      __syncthreads();
    }
    const int rem = ntiles - t;  // 1..3 tiles left: tile t in buffer 0, tile t+1 (if any) in the second register set
    if (rem >= 3) load(k0 + 2 * BK, ra0, rb0);
    multiply(0);
    if (rem >= 2) {
      stage(1, ra1, rb1);
      __syncthreads();
      multiply(1);
    }
    if (rem >= 3) {
      stage(0, ra0, rb0);  // (buffer 0 was last read before the barrier above)
      __syncthreads();
      multiply(0);
    }
  }

#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int m = m0 + wm * WM + i * 16 + li;
      const int n = n0 + wn * WN + j * 16 + 4 * kq;
      if (m >= g.M || n >= g.N) continue;
      f32x4 v = acc[i][j] * g.alpha;
      if (vec_c && n + 3 < g.N) {
        if (bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bias[n + r];
        }
        if (epi && g.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (mask) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = mk[i][j][r] > 0.f ? v[r] : 0.f;
        }
        if (g.stream_c)
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(C + (size_t)m * g.ldc + n));
        else
          *reinterpret_cast<f32x4*>(C + (size_t)m * g.ldc + n) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= g.N) continue;
          float x = v[r] + (bias ? bias[n + r] : 0.f);
          if (epi && g.relu) x = fmaxf(x, 0.f);
          if (mask) x = mask[(size_t)m * g.ldmask + n + r] > 0.f ? x : 0.f;
          C[(size_t)m * g.ldc + n + r] = x;
        }
      }
    }
}

#ifndef CURLA_GEMM_XCD
#define CURLA_GEMM_XCD 1
#endif
// Workgroup b runs on XCD b mod 8 (each XCD has its own L2).  xcd_order gives XCD x the x-th eighth of the tile list, so
// that with the tiles listed row tile fastest, then column tile, then problem, an XCD's workgroups share a few column
// tiles of B and the rows of A of ONE problem instead of touching every problem's A.
__device__ __forceinline__ int xcd_order(int b, int total) {
  return (CURLA_GEMM_XCD && (total & 7) == 0) ? (b & 7) * (total >> 3) + (b >> 3) : b;
}

template <bool AK, bool BKM, int TBM, int TBN, bool FAST, bool B3 = false, int NT = 256>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) GemmLds<TBM> As;
  __shared__ __attribute__((aligned(16))) GemmLds<TBN> Bs;
  // one-dimensional grid: gx * gy * (nbatch * ksplit) workgroups in XCD order
  const int gx = (g.N + TBN - 1) / TBN, gy = (g.M + TBM - 1) / TBM;
  const int v = xcd_order(blockIdx.x, gx * gy * g.nbatch * g.ksplit);
  const int r = v / gy;
  gemm_tile<AK, BKM, TBM, TBN, FAST, B3, NT>(g, r % gx, v - r * gy, r / gx, As, Bs);
}

// Backward of a linear layer as ONE launch: the weight gradient dW = dy^T x (both operands k-major, k = the batch rows)
// and the data gradient dx = (dy W) masked (W k-major) are independent products over the same dy -- the first `n1`
// workgroups take the tiles of the first, the rest the tiles of the second (grids gx x gy x batch each).  Saves a
// launch's ramp and tail per layer and lets the second product's first workgroups fill the first one's last round.
template <int TM1, int TN1, int TM2, int TN2, bool B3 = false, bool SECOND_FIRST = false, int NT = 256>
__global__ __launch_bounds__(NT) void gemm_pair_kernel(GemmArgs g1, GemmArgs g2, int n1, int gx1, int gy1, int gx2,
                                                        int gy2, int n2) {
  constexpr int RA = TM1 > TM2 ? TM1 : TM2, RB = TN1 > TN2 ? TN1 : TN2;
  __shared__ __attribute__((aligned(16))) GemmLds<RA> As;
  __shared__ __attribute__((aligned(16))) GemmLds<RB> Bs;
  int bid = blockIdx.x;
  // XCD order inside each product (xcd_order), the second product's share of an XCD first when SECOND_FIRST (its tiles
  // are the longer ones: k = N against k = the batch rows -- the short ones then fill the last round)
  if (CURLA_GEMM_XCD && ((n1 | n2) & 7) == 0) {
    const int x = bid & 7, i = bid >> 3, s1 = n1 >> 3, s2 = n2 >> 3;
    const int fst = SECOND_FIRST ? s2 : s1;
    const bool in_first = i < fst;
    const bool second = SECOND_FIRST ? in_first : !in_first;
    const int j = in_first ? i : i - fst;
    bid = second ? n1 + x * s2 + j : x * s1 + j;
  } else if (SECOND_FIRST) {
    bid = bid < n2 ? bid + n1 : bid - n2;
  }
  if (bid < n1) {
    const int r = bid / gy1;
    gemm_tile<true, true, TM1, TN1, true, B3, NT>(g1, r % gx1, bid - r * gy1, r / gx1,
                                                  reinterpret_cast<float(*)[gemm_tile_floats(TM1)]>(&As[0][0]),
                                                  reinterpret_cast<float(*)[gemm_tile_floats(TN1)]>(&Bs[0][0]));
  } else {
    bid -= n1;
    const int r = bid / gy2;
    gemm_tile<false, true, TM2, TN2, true, B3, NT>(g2, r % gx2, bid - r * gy2, r / gx2,
                                                   reinterpret_cast<float(*)[gemm_tile_floats(TM2)]>(&As[0][0]),
                                                   reinterpret_cast<float(*)[gemm_tile_floats(TN2)]>(&Bs[0][0]));
  }
}

// ---------------------------------------------------------------------------
// Backward of the encoder fc layer (encoder.py:98, z = fc(h), h = the NHWC-flattened conv output, K = 30752 columns at
// 84x84): both products have ONE tiny dimension (F <= 64 features) and stream a [B][K] matrix of tens of MB once.
// The generic kernel cuts them into 64 x 64 tiles of 2 k-tiles each -- thousands of short workgroups whose time is
// prologue and epilogue.  Here a workgroup owns 64 columns for its whole life and nothing is staged in LDS: the big
// operand is read with one float4 per lane whose four components are the column operands of four MFMA tiles
// (column j of tile c is n0 + 4j + c), so that the four accumulators of a lane hold 4 CONSECUTIVE columns and every
// wave store / mask load is 256 contiguous bytes per matrix row.
//   fc_dx:  G[b][n] = (mask[b][n] > 0) * sum_f dz[b][f] W[f][n]     (ReLU mask of the last conv layer fused)
//   fc_dw:  dW[f][n] = sum_b dz[b][f] x[b][n]
// ---------------------------------------------------------------------------
__device__ float g_zero16[16];  // where lanes past a reduction range point their operand load (see conv.hip g_zero_px)

constexpr int kFcMaxSteps = 16;  // F <= 64

struct FcBwdArgs {
  const float* dz;    // [B][F]
  const float* W;     // fc_dx: weight [F][K];  fc_dw: x [B][K]
  const float* mask;  // fc_dx: [B][K] (may be NULL)
  float* out;         // fc_dx: G [B][K];  fc_dw: dW [F][K]
  int B, F, K;
};

template <int KS>  // KS = ceil(F / 4) k-steps
__device__ __forceinline__ void fc_dx_body(const FcBwdArgs& g, const int bid) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int n0 = bid * 64;
  const int ncol = n0 + 4 * li;            // this lane's 4 columns (float4)
  const bool cvalid = ncol + 3 < g.K;      // K % 4 == 0: a float4 is inside or outside as a whole
  const int nc = cvalid ? ncol : n0;       // (columns past the edge re-read the first ones; their results are not stored)
  // column operand: W[f = 4s + kq][nc .. nc+3] for every k-step, kept for the workgroup's whole life
  f32x4 wv[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int f = 4 * s + kq;
    wv[s] = *reinterpret_cast<const f32x4*>(g.W + (size_t)min(f, g.F - 1) * g.K + nc);
  }
  const int ntiles = (g.B + 15) >> 4;
  // row operand of b-tile t: dz[16t + li][4s + kq]; features past F multiply a zero (loaded from the zero page)
  auto load_dz = [&](int t, float (&dv)[KS]) {
    const int b = min(16 * t + li, g.B - 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int f = 4 * s + kq;
      dv[s] = *(f < g.F ? g.dz + (size_t)b * g.F + f : g_zero16);
    }
  };
  auto load_mask = [&](int t, f32x4 (&mk)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = min(16 * t + 4 * kq + r, g.B - 1);
      mk[r] = g.mask ? *reinterpret_cast<const f32x4*>(g.mask + (size_t)b * g.K + nc) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
  };
  auto compute = [&](int t, const float (&dv)[KS], const f32x4 (&mk)[4]) {
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = mfma16(dv[s], wv[s][c], acc[c]);
    // lane (li, kq): rows b = 16t + 4kq + r, columns nc + c
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = 16 * t + 4 * kq + r;
      f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = mk[r][c] > 0.f ? v[c] : 0.f;
      if (cvalid && b < g.B) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(g.out + (size_t)b * g.K + ncol));
    }
  };
  // this wave's b-tiles: wave, wave + 4, ...; operands (and the mask) of the next tile are requested before the current
  // one's MFMAs (a third register set in flight measured no faster: 39.4 against 37.5 us at B = 512, K = 30752)
  float dA[KS], dB[KS];
  f32x4 mA[4], mB[4];
  int t = wave;
  if (t < ntiles) {
    load_dz(t, dA);
    load_mask(t, mA);
  }
  for (; t < ntiles; t += 8) {
    const int t1 = min(t + 4, ntiles - 1), t2 = min(t + 8, ntiles - 1);  // (past the end: re-read the last tile)
    load_dz(t1, dB);
    load_mask(t1, mB);
    __builtin_amdgcn_sched_barrier(0);
    compute(t, dA, mA);
    __builtin_amdgcn_sched_barrier(0);
    load_dz(t2, dA);
    load_mask(t2, mA);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 4 < ntiles) compute(t + 4, dB, mB);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int KS>
__global__ __launch_bounds__(256, 2) void fc_dx_kernel(FcBwdArgs g) {
  fc_dx_body<KS>(g, blockIdx.x);
}

template <int NT>  // NT = ceil(F / 16) feature tiles
__device__ __forceinline__ void fc_dw_body(const FcBwdArgs& g, const int bid) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][NT][4][64 lanes] float4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int n0 = bid * 64;
  const int ncol = n0 + 4 * li;
  const bool cvalid = ncol + 3 < g.K;
  const int nc = cvalid ? ncol : n0;
  const int nsteps = (g.B + 3) >> 2;  // k-steps of 4 rows b
  const int s0 = (int)((long long)wave * nsteps / 4), s1 = (int)((long long)(wave + 1) * nsteps / 4);
  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0, 0, 0, 0};
  // step s: rows b = 4s + kq.  Column operand: x[b][nc..nc+3]; row operand of feature tile t: dz[b][16t + li]
  // (features past F: re-read feature F-1, those output rows are not stored; rows past B: zeros from the zero page)
  auto fetch = [&](int s, f32x4& xv, float (&dv)[NT]) {
    const int b = 4 * min(s, s1 - 1) + kq;
    const bool bv = b < g.B;
    xv = *reinterpret_cast<const f32x4*>(g.W + (size_t)min(b, g.B - 1) * g.K + nc);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      dv[t] = *(bv ? g.dz + (size_t)b * g.F + min(16 * t + li, g.F - 1) : g_zero16);
  };
  auto multiply = [&](const f32x4& xv, const float (&dv)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[t][c] = mfma16(dv[t], xv[c], acc[t][c]);
  };
  if (s1 > s0) {
    // eight register sets: a k-step is 16 MFMAs (~0.4 us with the SIMD shared), an HBM round trip several times that
    constexpr int D = 8;
    f32x4 xr[D];
    float dr[D][NT];
#pragma unroll
    for (int d = 0; d < D - 1; ++d) fetch(s0 + d, xr[d], dr[d]);
    for (int s = s0; s < s1; s += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        fetch(s + d + D - 1, xr[(d + D - 1) % D], dr[(d + D - 1) % D]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + d < s1) multiply(xr[d], dr[d]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // sum over the four waves (b ranges) in wave order, then wave w writes feature tile(s) w, w+4, ...
  f32x4* r4 = reinterpret_cast<f32x4*>(red);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) r4[((wave * NT + t) * 4 + c) * 64 + lane] = acc[t][c];
  __syncthreads();
  for (int t = wave; t < NT; t += 4) {
    f32x4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[c] = r4[((0 * NT + t) * 4 + c) * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) v[c] += r4[((w * NT + t) * 4 + c) * 64 + lane];
    }
    // lane (li, kq): rows f = 16t + 4kq + r, columns nc + c
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * t + 4 * kq + r;
      const f32x4 o = {v[0][r], v[1][r], v[2][r], v[3][r]};
      if (cvalid && f < g.F) *reinterpret_cast<f32x4*>(g.out + (size_t)f * g.K + ncol) = o;
    }
  }
}

template <int NT>
__global__ __launch_bounds__(256, 2) void fc_dw_kernel(FcBwdArgs g) {
  fc_dw_body<NT>(g, blockIdx.x);
}

// Both products of the fc backward in ONE pass over the activations: the ReLU mask of the data gradient and the
// column operand of the weight gradient are the SAME matrix x (the last conv layer's output), and both bodies above
// read it in the same lane layout -- lane (li, kq) holds x[16t + 4kq + r][nc .. nc+3], r = 0..3, of b-tile t: for the
// data gradient that is the mask of its four accumulator rows, for the weight gradient the column operand of k-step r
// (a step multiplies rows 16t + 4kq + r: any order of the batch rows is a valid k order as long as the dz operand
// follows it).  One workgroup per 64 columns does both, x is read from HBM once instead of twice (63 of 206 MB per
// launch at B = 512, K = 30752).
// (B % 16 == 0, 4 (KS - 1) < F <= 4 KS and matrices below 2 GB, checked by the host.)  Every operand goes through
// buffer loads -- descriptor in SGPRs, a 32-bit lane offset that never changes, the tile / row / k-step part of the
// address in the scalar offset: with flat addresses the ~45 loads of a tile kept 64-bit lane addresses alive (35 spilled
// registers, each reload a full wait in front of its load).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 buf_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// NT feature tiles of 16 go through the matrix pipe; NTAIL further features (F = 16 NT + NTAIL: 50 = 3 x 16 + 2) are
// accumulated by VALU FMAs on the x values the lane holds anyway -- a fourth tile for two features would be an eighth
// of the kernel's MFMAs.
// DX = false: the weight gradient alone (the actor's fc layer: its conv stack is detached, curl_sac.py:375-376) -- the
// same walk without the data-gradient half.
// The LayerNorm parameter gradients that curla_ln_bwd_partial left as per-workgroup partial sums [nparts][3][F] (dgamma,
// dbeta, fc bias gradient): summed in partial order by ONE extra workgroup of this launch (the last), eight loads in
// flight -- a launch less per LayerNorm backward.  partial == nullptr: nothing to do, no extra workgroup.
struct LnReduce {
  const float* partial;
  int nparts, F;
  float *dgamma, *dbeta, *dbias;
  // a second set of partial sums for the same workgroup (the CURL head's dW partials, curla_curl_head):
  // xout[i] = sum over xparts partials of xpartial[p * xlen + i]; xpartial == nullptr: none
  const float* xpartial;
  int xparts, xlen;
  float* xout;
};
__device__ __forceinline__ void ln_reduce_block(const LnReduce& r) {
  if (r.xpartial) {
    for (int i = threadIdx.x; i < r.xlen; i += blockDim.x) {
      const float* p = r.xpartial + i;
      float a = 0.f;
      int k = 0;
      for (; k + 8 <= r.xparts; k += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p[(size_t)(k + u) * r.xlen];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += t[u];
      }
      for (; k < r.xparts; ++k) a += p[(size_t)k * r.xlen];
      r.xout[i] = a;
    }
  }
  for (int i = threadIdx.x; i < 3 * r.F; i += blockDim.x) {
    const int q = i / r.F, f = i - q * r.F;
    float* out = q == 0 ? r.dgamma : q == 1 ? r.dbeta : r.dbias;
    if (!out) continue;
    const float* p = r.partial + i;
    const size_t st = (size_t)3 * r.F;
    float a = 0.f;
    int k = 0;
    for (; k + 8 <= r.nparts; k += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = p[(k + u) * st];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += t[u];
    }
    for (; k < r.nparts; ++k) a += p[k * st];
    out[f] = a;
  }
}

template <int KS, int NT, int NTAIL, bool DX>
__global__ __launch_bounds__(256, 2) void fc_bwd_kernel(FcBwdArgs g, float* dW, LnReduce lnr) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][NT][4][64 lanes] float4
  if (lnr.partial && blockIdx.x == gridDim.x - 1) {  // (the grid has one workgroup more than column blocks)
    ln_reduce_block(lnr);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 64;
  const int ncol = n0 + 4 * li;
  const bool cvalid = ncol + 3 < g.K;
  const int nc = cvalid ? ncol : n0;
  f32x4 wv[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int f = 4 * s + kq;
    wv[s] = DX ? *reinterpret_cast<const f32x4*>(g.W + (size_t)min(f, g.F - 1) * g.K + nc) : f32x4{0, 0, 0, 0};
  }
  f32x4 accw[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) accw[t][c] = f32x4{0, 0, 0, 0};
  f32x4 atail[NTAIL > 0 ? NTAIL : 1];  // [feature 16 NT + j] x this lane's 4 columns, summed over its rows
#pragma unroll
  for (int j = 0; j < (NTAIL > 0 ? NTAIL : 1); ++j) atail[j] = f32x4{0, 0, 0, 0};
  const unsigned tl_off = (unsigned)(4 * kq * g.F + 16 * NT) * 4u;  // dz[16t + 4kq + r][16 NT + j]: + 4 r F + 4 j
  const int ntiles = g.B >> 4;
  const unsigned dz_bytes = (unsigned)g.B * g.F * 4u, x_bytes = (unsigned)g.B * (unsigned)g.K * 4u;
  const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc((void*)g.dz, (short)0, (int)dz_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)g.mask, (short)0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)g.out, (short)0, (int)x_bytes, 0x00020000);
  // lane offsets (bytes) inside a b-tile's 16 rows of dz / x; everything else of an address is wave-uniform
  const unsigned dv_off = (unsigned)(li * g.F + kq) * 4u;  // dz[16t + li][4s + kq]: + 16 s
  // last k-step: features past F multiply a zero -- their lanes point past the buffer (an out-of-range load returns 0)
  const unsigned dv_last = 4 * (KS - 1) + kq < g.F ? dv_off : 0x80000000u;
  unsigned dt_off[NT];                                    // dz[16t + 4kq + r][16ft + li]: + 4 r F
#pragma unroll
  for (int ft = 0; ft < NT; ++ft) dt_off[ft] = (unsigned)(4 * kq * g.F + min(16 * ft + li, g.F - 1)) * 4u;
  const unsigned x_off = ((unsigned)(4 * kq) * (unsigned)g.K + (unsigned)nc) * 4u;  // x[16t + 4kq + r][nc]: + 4 r K
  const unsigned rowF = 4u * g.F, rowK = 4u * (unsigned)g.K;
  auto load_dv = [&](int t, float (&dv)[KS]) {
    const unsigned base = (unsigned)t * 16u * rowF;
#pragma unroll
    for (int s = 0; s < KS - 1; ++s) dv[s] = buf_f32(rdz, dv_off, base + 16u * s);
    dv[KS - 1] = buf_f32(rdz, dv_last, base + 16u * (KS - 1));
  };
  auto load_dt = [&](int t, float (&dt)[4][NT]) {
    const unsigned base = (unsigned)t * 16u * rowF;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ft = 0; ft < NT; ++ft) dt[r][ft] = buf_f32(rdz, dt_off[ft], base + r * rowF);  // (features past F: F-1 again, never stored)
  };
  auto load_mk = [&](int t, f32x4 (&mk)[4]) {
    const unsigned base = (unsigned)t * 16u * rowK;
#pragma unroll
    for (int r = 0; r < 4; ++r) mk[r] = buf_f32x4(rx, x_off, base + r * rowK);
  };
  auto load_tl = [&](int t, float (&tl)[4][NTAIL > 0 ? NTAIL : 1]) {
    const unsigned base = (unsigned)t * 16u * rowF;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < NTAIL; ++j) tl[r][j] = buf_f32(rdz, tl_off, base + r * rowF + 4u * j);
  };
  float dv[KS], dt[4][NT], tl[4][NTAIL > 0 ? NTAIL : 1];
  // x comes from HBM and is requested a whole tile ahead (two register sets); the dz operands are L2 hits and each has
  // the other product's MFMAs to arrive under
  auto tile = [&](int t, int tnext, const f32x4 (&mk)[4], f32x4 (&mk_next)[4]) {
    load_mk(tnext, mk_next);
    load_dt(t, dt);
    if (NTAIL > 0) load_tl(t, tl);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0, 0, 0, 0};
    if (DX) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = mfma16(dv[s], wv[s][c], acc[c]);
      __builtin_amdgcn_sched_barrier(0);
      load_dv(tnext, dv);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ft = 0; ft < NT; ++ft)
#pragma unroll
        for (int c = 0; c < 4; ++c) accw[ft][c] = mfma16(dt[r][ft], mk[r][c], accw[ft][c]);
#pragma unroll
    for (int j = 0; j < NTAIL; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) atail[j] += tl[r][j] * mk[r];
    if (DX) {
      const unsigned obase = (unsigned)t * 16u * rowK;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = mk[r][c] > 0.f ? v[c] : 0.f;
        if (cvalid) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, x_off, obase + r * rowK, 2);  // nt
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x4 mA[4], mB[4];
  int t = wave;  // this wave's b-tiles: wave, wave + 4, ...
  if (t < ntiles) {
    load_mk(t, mA);
    if (DX) load_dv(t, dv);
  }
  for (; t < ntiles; t += 8) {
    tile(t, min(t + 4, ntiles - 1), mA, mB);  // (past the end: the last tile again, unused)
    if (t + 4 < ntiles) tile(t + 4, min(t + 8, ntiles - 1), mB, mA);
  }
  // weight gradient: sum over the four waves (b-tiles) in wave order, then wave w writes feature tile(s) w, w+4, ...
  f32x4* r4 = reinterpret_cast<f32x4*>(red);
  f32x4* t4 = r4 + 4 * NT * 4 * 64;  // tail features: [4 waves][NTAIL][16 column groups]
  if (NTAIL > 0) {
#pragma unroll
    for (int j = 0; j < NTAIL; ++j) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {  // the four row quarters (kq) of a column group sit in lanes li, li+16, li+32, li+48
        atail[j][c] += __shfl_xor(atail[j][c], 16);
        atail[j][c] += __shfl_xor(atail[j][c], 32);
      }
      if (kq == 0) t4[(wave * NTAIL + j) * 16 + li] = atail[j];
    }
  }
#pragma unroll
  for (int ft = 0; ft < NT; ++ft)
#pragma unroll
    for (int c = 0; c < 4; ++c) r4[((wave * NT + ft) * 4 + c) * 64 + lane] = accw[ft][c];
  __syncthreads();
  for (int ft = wave; ft < NT; ft += 4) {
    f32x4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[c] = r4[((0 * NT + ft) * 4 + c) * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) v[c] += r4[((w * NT + ft) * 4 + c) * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * ft + 4 * kq + r;
      const f32x4 o = {v[0][r], v[1][r], v[2][r], v[3][r]};
      if (cvalid && f < g.F) *reinterpret_cast<f32x4*>(dW + (size_t)f * g.K + ncol) = o;
    }
  }
  if (NTAIL > 0 && tid < 16 * NTAIL) {  // (after the barrier above)
    const int j = tid >> 4, cg = tid & 15;
    f32x4 v = t4[(0 * NTAIL + j) * 16 + cg];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += t4[(w * NTAIL + j) * 16 + cg];
    const int col = n0 + 4 * cg;
    if (col + 3 < g.K) *reinterpret_cast<f32x4*>(dW + (size_t)(16 * NT + j) * g.K + col) = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Encoder fc layer, forward: split-K partial sums  P[s][b][f] = sum over the k of split s of x[b][k] W[f][k]  for up to
// four (x, W) pairs of one shape in one launch (encoder.py:67,94-101: the fc in front of the LayerNorm; the partials are
// added up by fc_ln_fwd_kernel).
//   D[f][b] tiles: A operand = W (rows f), B operand = x (columns b); k is walked in a permuted order -- a lane's quarter
//   kq of a 32-wide k-slice is 8 CONTIGUOUS floats (k0 + 8 kq + s, s = the MFMA step), so both operands are 16-byte
//   reads.  A wave owns two b-tiles (32 rows); a 256-thread workgroup covers 128 rows and one K range.  NT full feature
//   tiles go through the matrix pipe, NTAIL further features (50 = 3 x 16 + 2) are FMAs on the x values the lane holds.
//   x comes from HBM straight into registers, three slices ahead (four register sets); W -- the same for the four waves --
//   goes through LDS in chunks of four slices, double-buffered: one barrier per 192 MFMAs of a wave.
// x layouts: row-major [B][K], or BLOCKED [B / 16][K / 32][16][32] (16 samples' pixel records adjacent; a k-slice = the
// 32 channels of one pixel): a slice of a b-tile is then 2 KB contiguous and a b-tile's slices follow each other.
// ---------------------------------------------------------------------------------------------------------------------
struct FcFwdArgs {
  const float* x[4];
  const float* W[4];
  float* out[4];
  int nprob, B, F, K, nsplit, nrg;  // nrg = B / 128 row groups
  int blocked;                      // x layout: 0 row-major, 1 blocked (see above)
  long long split_stride;           // floats between two splits' partials
};

constexpr int kFcChunk = 4;            // k-slices per W chunk in LDS
constexpr int kFcPitch = 32 * kFcChunk + 4;  // floats per feature row of a chunk (16-byte aligned, off the bank stride)
// LDS row of feature row r of a 16-feature tile.  A ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- and lane (li, kq) reads 16 bytes at row
// li, float 8 kq: with rows in order (bank slot li + 2 kq mod 16) lanes li = 12, 13 of kq = 0 meet li = 10, 11 of kq = 1
// in one group: every W read 2-way conflicted (SQ_LDS_BANK_CONFLICT 0.29 of the LDS cycles, rounds 4-5).  A group always
// pairs li in [4, 12) of one kq with the other eight li of the neighbouring kq, two slots apart: rows 4 .. 11 on the even
// positions and the others on the odd ones keep the two halves of every group on different parities.
__device__ __forceinline__ int fc_w_row(int r) { return r >= 4 && r < 12 ? 2 * (r - 4) : r < 4 ? 2 * r + 1 : 2 * (r - 12) + 9; }

template <int NT, int NTAIL, bool B3 = false>
__global__ __launch_bounds__(256, 2) void fc_fwd_kernel(FcFwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [2][F][kFcPitch]
  constexpr int NTL = NTAIL > 0 ? NTAIL : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  int idx = blockIdx.x;
  const int split = idx % g.nsplit;
  idx /= g.nsplit;
  const int rg = idx % g.nrg, prob = idx / g.nrg;
  const int nslices = g.K >> 5;
  const int s0 = (int)((long long)nslices * split / g.nsplit), s1 = (int)((long long)nslices * (split + 1) / g.nsplit);
  const int t0 = rg * 8 + wave * 2;  // this wave's b-tiles t0, t0 + 1
  const unsigned rowK = 4u * (unsigned)g.K;
  auto uniform_rsrc = [](const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi32 << 32) | lo32), (short)0,
                                             (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rx = uniform_rsrc(g.x[prob], (unsigned)g.B * rowK);
  const __amdgpu_buffer_rsrc_t rw = uniform_rsrc(g.W[prob], (unsigned)g.F * rowK);
  const unsigned xslice = g.blocked ? 2048u : 128u;  // bytes from one k-slice of a b-tile to the next
  unsigned xo[2];
#pragma unroll
  for (int bt = 0; bt < 2; ++bt)
    xo[bt] = g.blocked ? (unsigned)(t0 + bt) * (unsigned)nslices * 2048u + 128u * li + 32u * kq
                       : (unsigned)(16 * (t0 + bt) + li) * rowK + 32u * kq;
  struct XS {
    f32x4 v[2][2];  // [b-tile][half]: x[row][k0 + 8 kq + 4 half + e]
  };
  auto load_x = [&](XS& X, int sl) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)min(sl, s1 - 1) * xslice);
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
      for (int h = 0; h < 2; ++h) X.v[bt][h] = buf_f32x4(rx, xo[bt] + 16u * h, so);
  };
  // W chunk c (slices s0 + 4 c ..): F rows x 128 floats, 16-byte pieces spread over the workgroup
  constexpr int NPC = 8 * kFcChunk;                 // 16-byte pieces per feature row and chunk
  const int npieces = g.F * NPC;
  constexpr int WREG = (64 * NPC + 255) / 256;       // pieces per thread (F <= 64)
  f32x4 wst[WREG];
  auto fetch_w = [&](int c) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(s0 + kFcChunk * c) * 128u);
#pragma unroll
    for (int u = 0; u < WREG; ++u) {
      const int p = tid + 256 * u;
      const int f = p / NPC, q = p - f * NPC;
      // (pieces past the last feature or past the matrix' end: out of range, zeros)
      const unsigned off = p < npieces ? (unsigned)f * rowK + 16u * q : 0x80000000u;
      wst[u] = buf_f32x4(rw, off, so);
    }
  };
  auto commit_w = [&](int buf) {
    float* dst = wl + buf * 64 * kFcPitch;
#pragma unroll
    for (int u = 0; u < WREG; ++u) {
      const int p = tid + 256 * u;
      const int f = p / NPC, q = p - f * NPC;
      // (tile rows permuted: fc_w_row; the tail features behind the full tiles -- read by all lanes at once -- stay in order)
      const int fr = f < 16 * NT ? (f & ~15) + fc_w_row(f & 15) : f;
      if (p < npieces) *reinterpret_cast<f32x4*>(dst + fr * kFcPitch + 4 * q) = wst[u];
    }
  };
  f32x4 acc[2][NT];
  f32x2 atail[2][NTL];  // (even / odd k of the lane's values apart: one packed FMA per register pair, added at the end)
#pragma unroll
  for (int bt = 0; bt < 2; ++bt) {
#pragma unroll
    for (int ft = 0; ft < NT; ++ft) acc[bt][ft] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NTL; ++j) atail[bt][j] = f32x2{0.f, 0.f};
  }
  auto multiply = [&](const XS& X, const float* wc, int sj) {  // slice sj of the chunk at wc
    f32x4 wv[NT][2], wt[NTL][2];
#pragma unroll
    for (int ft = 0; ft < NT; ++ft)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        wv[ft][h] = *reinterpret_cast<const f32x4*>(wc + (16 * ft + fc_w_row(li)) * kFcPitch + 32 * sj + 8 * kq + 4 * h);
    if (NTAIL > 0) {
#pragma unroll
      for (int j = 0; j < NTAIL; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          wt[j][h] = *reinterpret_cast<const f32x4*>(wc + (16 * NT + j) * kFcPitch + 32 * sj + 8 * kq + 4 * h);
    }
    if constexpr (B3) {
      // bf16x3 (conv_rwb.h): a slice is ONE k-step of v_mfma_f32_16x16x32_bf16 -- a lane's 8 contiguous k of x and of W
      // are exactly its operand -- six terms per fp32 product, smallest first: 36 matrix instructions of 16 cycles per
      // slice instead of 48 of 32; the split is 5.5 VALU per value (x: 16 values per lane, W: 8 per feature tile)
      Frag3 xb[2];
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) xb[bt] = g_split8(X.v[bt][0], X.v[bt][1]);
#pragma unroll
      for (int ft = 0; ft < NT; ++ft) {
        const Frag3 wf = g_split8(wv[ft][0], wv[ft][1]);
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int bt = 0; bt < 2; ++bt) {
            const gu32x4& pw = term == 0 ? wf.l : (term == 2 || term == 3) ? wf.m : wf.h;
            const gu32x4& px = (term == 0 || term == 3 || term == 5) ? xb[bt].h : (term == 1) ? xb[bt].l : xb[bt].m;
            acc[bt][ft] = g_mfma_bf16(pw, px, acc[bt][ft]);
          }
      }
    } else {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ft = 0; ft < NT; ++ft)
#pragma unroll
          for (int bt = 0; bt < 2; ++bt) acc[bt][ft] = mfma16(wv[ft][h][e], X.v[bt][h][e], acc[bt][ft]);
    }
    if (NTAIL > 0) {
#pragma unroll
      for (int j = 0; j < NTAIL; ++j)
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            atail[bt][j] += f32x2{X.v[bt][h][0], X.v[bt][h][1]} * f32x2{wt[j][h][0], wt[j][h][1]};
            atail[bt][j] += f32x2{X.v[bt][h][2], X.v[bt][h][3]} * f32x2{wt[j][h][2], wt[j][h][3]};
          }
    }
  };
  const int nchunks = (s1 - s0 + kFcChunk - 1) / kFcChunk;
  if (nchunks > 0) {
    XS X0, X1, X2, X3;
    fetch_w(0);
    load_x(X0, s0), load_x(X1, s0 + 1), load_x(X2, s0 + 2);
    commit_w(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      const float* wc = wl + (c & 1) * 64 * kFcPitch;
      const int sl = s0 + kFcChunk * c;
      fetch_w(c + 1);  // (unconditional -- a conditional request makes the compiler wait for ALL loads at the next use;
                       // past the split's end it reads the neighbour's slices or, past the matrix, zeros: never used)
      // four slices; x set i mod 4 holds slice i, requested three slices ahead (slices past the split's end are computed
      // on re-read data with W rows that are zero past the matrix: they are skipped instead)
      load_x(X3, sl + 3);
      multiply(X0, wc, 0);
      load_x(X0, sl + 4);
      if (sl + 1 < s1) multiply(X1, wc, 1);
      load_x(X1, sl + 5);
      if (sl + 2 < s1) multiply(X2, wc, 2);
      load_x(X2, sl + 6);
      if (sl + 3 < s1) multiply(X3, wc, 3);
      commit_w((c + 1) & 1);
      __syncthreads();
    }
  }
  // partial sums of this split: P[split][b][f]; lane (li, kq) holds features 16 ft + 4 kq + r of row 16 t + li
  float* out = g.out[prob] + (size_t)split * g.split_stride;
#pragma unroll
  for (int bt = 0; bt < 2; ++bt) {
    const int b = 16 * (t0 + bt) + li;
    float* row = out + (size_t)b * g.F;
#pragma unroll
    for (int ft = 0; ft < NT; ++ft) {
      const int f = 16 * ft + 4 * kq;
      if (f + 3 < g.F) {
        *reinterpret_cast<f32x2*>(row + f) = f32x2{acc[bt][ft][0], acc[bt][ft][1]};
        *reinterpret_cast<f32x2*>(row + f + 2) = f32x2{acc[bt][ft][2], acc[bt][ft][3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (f + r < g.F) row[f + r] = acc[bt][ft][r];
      }
    }
    if (NTAIL > 0) {
#pragma unroll
      for (int j = 0; j < NTAIL; ++j) {  // the four quarters of a slice's k sit in the four lane groups: add them in order
        float v = atail[bt][j][0] + atail[bt][j][1];
        const float q1 = __shfl(v, li + 16), q2 = __shfl(v, li + 32), q3 = __shfl(v, li + 48);
        if (kq == 0) row[16 * NT + j] = ((v + q1) + q2) + q3;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Products with a SMALL output and a long k: the data gradient of an MLP's first layer ([B x 1024] x [1024 x 54]),
// its weight gradient ([1024 x 54] over B rows), the CURL bilinear products ([B x B] x [B x 50], [50 x 50] over B
// rows; curl_sac.py:129-139, 219-220, 406-423).  The tiled kernel gives them a few dozen workgroups that each walk
// the whole k range one LDS tile at a time -- 12-19 us of latency for 0.1 GFLOP.  Here a workgroup owns ONE 16 x 16
// output tile, its NW waves split k between them in chunks of 16 (wave w takes chunks w, w + NW, ...), operands go
// from global memory straight into MFMA registers with every load of a wave in flight at once, and the NW partial
// tiles are added in wave order through LDS.  k is walked in a permuted order inside a chunk (MFMA step s of
// quarter-lane-group q multiplies k = 16c + 4q + s), so a row-major operand is one float4 per lane and chunk.
// ---------------------------------------------------------------------------
template <bool AK, bool BKM, int NW, int U>
__device__ __forceinline__ void gemm_small_tile(const GemmArgs& g, const int bx, const int by, const int batch,
                                                f32x4 (&red)[NW][64], float (&red_cs)[NW][16]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int bo = batch / g.nb_inner, bi = batch - bo * g.nb_inner;
  const float* A = g.A + bi * g.sA + bo * g.sA2;
  const float* B = g.B + bi * g.sB + bo * g.sB2;
  float* C = g.C + bi * g.sC + bo * g.sC2;
  const int m0 = by * 16, n0 = bx * 16;
  const int mi = min(m0 + li, g.M - 1), ni = min(n0 + li, g.N - 1);  // (edge lanes re-read the last row; never stored)
  // a lane's element (row r, k) of chunk c, step s: k = 16c + 4kq + s
  const float* pa = AK ? A + (size_t)(4 * kq) * g.lda + mi : A + (size_t)mi * g.lda + 4 * kq;
  const float* pb = BKM ? B + (size_t)(4 * kq) * g.ldb + ni : B + (size_t)ni * g.ldb + 4 * kq;
  const size_t ca = AK ? (size_t)16 * g.lda : 16, cb = BKM ? (size_t)16 * g.ldb : 16;
  // colsum: the workgroups of the first column of tiles also add up the A operands they load anyway
  const bool do_cs = g.colsum && bx == 0;  // block-uniform
  float asum = 0.f;
  const int T = (g.K / 16) / NW;  // chunks per wave (the host guarantees K % (16 * NW) == 0)
  struct Frag {
    float a[4], b[4];
  };
  auto ld = [&](int t, Frag& f) {
    const size_t c = (size_t)(wave + NW * min(t, T - 1));  // (past the end: the last chunk again, unused)
    const float* a = pa + c * ca;
    const float* b = pb + c * cb;
    if (AK) {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.a[s] = a[(size_t)s * g.lda];
    } else if (g.vecA) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(a);
      f.a[0] = v[0], f.a[1] = v[1], f.a[2] = v[2], f.a[3] = v[3];
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.a[s] = a[s];
    }
    if (BKM) {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.b[s] = b[(size_t)s * g.ldb];
    } else if (g.vecB) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(b);
      f.b[0] = v[0], f.b[1] = v[1], f.b[2] = v[2], f.b[3] = v[3];
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.b[s] = b[s];
    }
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  Frag cur[U], nxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) ld(u, cur[u]);
  for (int t0 = 0; t0 < T; t0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) ld(t0 + U + u, nxt[u]);  // unconditional: stays in flight under the MFMAs below
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (t0 + u < T) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma16(cur[u].b[s], cur[u].a[s], acc);
        if (do_cs) asum += (cur[u].a[0] + cur[u].a[1]) + (cur[u].a[2] + cur[u].a[3]);
      }
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = nxt[u];
  }
  if (do_cs) {  // the four k-quarters of a row sit in lanes li, li+16, li+32, li+48
    asum += __shfl_xor(asum, 16);
    asum += __shfl_xor(asum, 32);
    if (lane < 16) red_cs[wave][lane] = asum;
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 1; w < NW; ++w) acc += red[w][lane];
  if (do_cs && lane < 16 && m0 + lane < g.M) {
    float t = red_cs[0][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += red_cs[w][lane];
    g.colsum[bi * g.sColsum + bo * g.sColsum2 + m0 + lane] = t * g.alpha;
  }
  const int m = m0 + li, n = n0 + 4 * kq;  // (N-side operand first: a lane holds 4 consecutive n of one m)
  if (m >= g.M || n >= g.N) return;
  acc = acc * g.alpha;
  if ((g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && n + 3 < g.N) {
    *reinterpret_cast<f32x4*>(C + (size_t)m * g.ldc + n) = acc;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < g.N) C[(size_t)m * g.ldc + n + r] = acc[r];
  }
}

template <bool AK, bool BKM, int NW, int U>
__global__ __launch_bounds__(64 * NW) void gemm_small_kernel(GemmArgs g) {
  __shared__ f32x4 red[NW][64];
  __shared__ float red_cs[NW][16];
  gemm_small_tile<AK, BKM, NW, U>(g, blockIdx.x, blockIdx.y, blockIdx.z, red, red_cs);
}

// the two small products of a linear layer's backward pass (see gemm_pair_kernel) in one launch
template <int NW, int U>
__global__ __launch_bounds__(64 * NW) void gemm_small_pair_kernel(GemmArgs g1, GemmArgs g2, int n1, int gx1, int gy1,
                                                                  int gx2, int gy2) {
  __shared__ f32x4 red[NW][64];
  __shared__ float red_cs[NW][16];
  int bid = blockIdx.x;
  if (bid < n1) {
    const int r = bid / gx1;
    gemm_small_tile<true, true, NW, U>(g1, bid - r * gx1, r % gy1, r / gy1, red, red_cs);
  } else {
    bid -= n1;
    const int r = bid / gx2;
    gemm_small_tile<false, true, NW, U>(g2, bid - r * gx2, r % gy2, r / gy2, red, red_cs);
  }
}

// sum split-K partials: C[m][n] = sum_s P[s][m][n] (+bias, ReLU)
__global__ void splitk_reduce_kernel(const float* P, int nsplit, long long sSplit, int M, int N, int ldp, float* C,
                                     int ldc, const float* bias, int relu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int m = i / N, n = i - m * N;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += P[k * sSplit + (size_t)m * ldp + n];
  if (bias) s += bias[n];
  if (relu) s = fmaxf(s, 0.f);
  C[(size_t)m * ldc + n] = s;
}

}  // namespace

extern "C" {

// small output, long k: one workgroup per 16 x 16 tile, k split over its waves (gemm_small_kernel)
static bool small_shape(int M, int N, int K, int nbatch) {
  const long long t32 = (long long)((M + 31) / 32) * ((N + 31) / 32) * nbatch;
  return K >= 256 && K % 64 == 0 && t32 <= 128 && nbatch <= 65535;
}

// which kernel a product takes: the small-output kernel (16 or 4 waves per tile) or the tiled one with its tile shape
struct GemmPlan {
  bool small, wide, fast, b3;
  int tbm, tbn;
};

// the tiled kernel's arithmetic (option gemm_mfma): f32 = the f32-input MFMA everywhere; b3 = bf16x3 on the bf16 matrix
// cores for every interior, aligned tile; auto = bf16x3 where the 128 x 64 tile applies, f32 elsewhere.  History
// (profiles/r05_checks/r05_gemm_b3_vs_f32.txt): with the f32 form's own 64 x 64 / 64 x 32 / 32 x 32 tiles bf16x3 bought
// nothing (37.9 against 43.2 us, 45.1 / 45.2, 19.4 / 17.7) -- three bf16 images are 1.5 x the LDS bytes of the f32 tile
// and the matrix instructions take a sixth of the time, so those tiles are bound by LDS stores and fragment reads, not by
// the matrix pipe.  A 128 x 64 tile (a wave: 64 x 32, six fragment reads per 48 matrix instructions) halves the LDS
// traffic per product.
static int gemm_mfma_opt() { return curla_opt(kOptGemmMfma); }
static bool gemm_b3() { return gemm_mfma_opt() == 2; }

// whether the 128 x 64 bf16x3 tile can take this product at all (whole tiles, aligned operands, no k split)
static bool big_tile_ok(const GemmArgs& g) {
  return gemm_mfma_opt() != 1 && g.ksplit == 1 && g.vecA && g.vecB && g.K % BK == 0 && g.K >= 4 * BK && g.M % 128 == 0 &&
         g.N % 64 == 0 && !g.colsum;
}
static long long big_tile_count(const GemmArgs& g) { return (long long)(g.M / 128) * (g.N / 64) * g.nbatch; }

static int gemm_plan(GemmArgs& g, int a_kmajor, int b_kmajor, GemmPlan& p, bool force_big = false) {
  const int M = g.M, N = g.N, K = g.K, nbatch = g.nbatch, ksplit = g.ksplit;
  p.small = ksplit == 1 && !g.bias && !g.mask && !g.relu && g.nptr == 0 && small_shape(M, N, K, nbatch);
  p.wide = false, p.fast = false, p.b3 = false, p.tbm = p.tbn = 16;
  if (g.colsum && !p.small) return CURLA_ERR_UNSUPPORTED;
  if (p.small) {
    const long long t16 = (long long)((M + 15) / 16) * ((N + 15) / 16) * nbatch;
    p.wide = t16 < 64 && K % 256 == 0;  // a handful of tiles: 16 waves each
    return CURLA_OK;
  }
  int kc = (K + ksplit - 1) / ksplit;
  kc = (kc + BK - 1) / BK * BK;  // chunk boundaries stay float4-aligned
  g.kchunk = kc;
  g.stream_c = (long long)M * N * nbatch * (long long)sizeof(float) >= (32LL << 20);
  // tile shape: 64x64 when that already fills the chip twice, else 64x32, else 32x32 (more, smaller
  // workgroups: at one workgroup per CU nothing hides the L2 latency of the single-buffered k loop)
  const long long cu2 = 2LL * curla_cu_count();
  const long long zb = (long long)nbatch * ksplit;
  const long long wgs64 = (long long)((N + 63) / 64) * ((M + 63) / 64) * zb;
  const long long wgs6432 = (long long)((N + 31) / 32) * ((M + 63) / 64) * zb;
  int tbm = 64, tbn = 64;
  // (the encoder fc forward, [512 x 30752] x [30752 x 50] split 32 ways along K, lands here with 32-wide tiles and
  // so streams its 63 MB activation operand twice, 117 MB of HBM traffic per launch; the single-pass alternative,
  // one 64-wide N tile with a 64-way split, was measured: 26.9 us against 25.5 us -- the product is bound by the
  // padded MFMA work (N = 50 -> 64) and load latency, not by HBM bytes, and more, smaller workgroups hide those better)
  if (wgs64 < cu2 && N > 32) tbn = 32;
  // a single row of tiles streaming a k-major B (the fc weight gradient, M = 50): the wider tile reads 256-B
  // instead of 128-B pieces of each HBM row and is faster even at one workgroup per CU
  if (tbn == 32 && M <= 64 && a_kmajor && b_kmajor && 2 * wgs64 >= cu2) tbn = 64;
  if (tbn == 32 && wgs6432 < cu2 && M > 32) tbm = 32;
  // the 128 x 64 bf16x3 tile where it gives every CU a workgroup (or the caller has counted a pair's tiles together)
  const int tile_opt = curla_opt(kOptGemmTile);
  bool big = big_tile_ok(g) && (force_big || big_tile_count(g) >= curla_cu_count());
  switch (tile_opt) {  // option gemm_tile (options.h; tools/gemm_shapes.py): a forced tile shape
    case 1: tbm = 64, tbn = 64, big = false; break;
    case 2: tbm = 64, tbn = 32, big = false; break;
    case 3: tbm = 32, tbn = 32, big = false; break;
    case 4: big = big_tile_ok(g); break;
    default: break;
  }
  if (big) {
    p.tbm = 128, p.tbn = 64, p.fast = true, p.b3 = true;
    g.kchunk = g.K;
    return CURLA_OK;
  }
  p.tbm = tbm, p.tbn = tbn;
  // interior + aligned everywhere: the k loop runs without bounds / alignment tests
  // (a k-major operand still needs whole tiles: its float4 runs along the rows)
  p.fast = g.vecA && g.vecB && (K % BK == 0) && (g.kchunk % BK == 0) && (!a_kmajor || M % tbm == 0) &&
           (!b_kmajor || N % tbn == 0);
  p.b3 = p.fast && gemm_b3();
  return CURLA_OK;
}

static int gemm_launch(GemmArgs& g, int a_kmajor, int b_kmajor, hipStream_t st) {
  const int M = g.M, N = g.N, nbatch = g.nbatch, ksplit = g.ksplit;
  GemmPlan p;
  const int prc = gemm_plan(g, a_kmajor, b_kmajor, p);
  if (prc != CURLA_OK) return prc;
  if (p.small) {
    const bool wide = p.wide;
    const dim3 grid((N + 15) / 16, (M + 15) / 16, nbatch);
#define CURLA_GEMM_SMALL(AKM, BKMAJ)                                                              \
  do {                                                                                            \
    if (wide)                                                                                     \
      hipLaunchKernelGGL((gemm_small_kernel<AKM, BKMAJ, 16, 2>), grid, dim3(1024), 0, st, g);     \
    else                                                                                          \
      hipLaunchKernelGGL((gemm_small_kernel<AKM, BKMAJ, 4, 8>), grid, dim3(256), 0, st, g);       \
  } while (0)
    if (a_kmajor && b_kmajor)
      CURLA_GEMM_SMALL(true, true);
    else if (a_kmajor)
      CURLA_GEMM_SMALL(true, false);
    else if (b_kmajor)
      CURLA_GEMM_SMALL(false, true);
    else
      CURLA_GEMM_SMALL(false, false);
#undef CURLA_GEMM_SMALL
    return curla_launch_status();
  }
  const int tbm = p.tbm, tbn = p.tbn;
  const bool fast = p.fast;
  const bool b3 = p.b3;
#define CURLA_GEMM_LAUNCH3(AKM, BKMAJ, TM, TN, FS, B3F)                                                     \
  hipLaunchKernelGGL((gemm_kernel<AKM, BKMAJ, TM, TN, FS, B3F>), dim3(((N + TN - 1) / TN) * ((M + TM - 1) / TM) * nbatch * ksplit), \
                     dim3(256), 0, st, g)
#define CURLA_GEMM_LAUNCH2(AKM, BKMAJ, TM, TN)               \
  do {                                                       \
    if (b3)                                                  \
      CURLA_GEMM_LAUNCH3(AKM, BKMAJ, TM, TN, true, true);    \
    else if (fast)                                           \
      CURLA_GEMM_LAUNCH3(AKM, BKMAJ, TM, TN, true, false);   \
    else                                                     \
      CURLA_GEMM_LAUNCH3(AKM, BKMAJ, TM, TN, false, false);  \
  } while (0)
#define CURLA_GEMM_LAUNCH(AKM, BKMAJ)                        \
  do {                                                       \
    if (tbm == 128)                                          \
      hipLaunchKernelGGL((gemm_kernel<AKM, BKMAJ, 128, 64, true, true, 512>), dim3((N / 64) * (M / 128) * nbatch), dim3(512), 0, st, g); \
    else if (tbm == 32)                                      \
      CURLA_GEMM_LAUNCH2(AKM, BKMAJ, 32, 32);                \
    else if (tbn == 32)                                      \
      CURLA_GEMM_LAUNCH2(AKM, BKMAJ, 64, 32);                \
    else                                                     \
      CURLA_GEMM_LAUNCH2(AKM, BKMAJ, 64, 64);                \
  } while (0)
  if (a_kmajor && b_kmajor)
    CURLA_GEMM_LAUNCH(true, true);
  else if (a_kmajor)
    CURLA_GEMM_LAUNCH(true, false);
  else if (b_kmajor)
    CURLA_GEMM_LAUNCH(false, true);
  else
    CURLA_GEMM_LAUNCH(false, false);
#undef CURLA_GEMM_LAUNCH3
#undef CURLA_GEMM_LAUNCH2
#undef CURLA_GEMM_LAUNCH
  return curla_launch_status();
}

int curla_gemm(const float* A, int a_kmajor, int lda, long long strideA, const float* B, int b_kmajor, int ldb,
               long long strideB, float* C, int ldc, long long strideC, int M, int N, int K, int nbatch, int ksplit,
               long long split_stride, float alpha, const float* bias, long long strideBias, int relu,
               const float* mask, int ldmask, long long strideMask, void* stream) {
  CURLA_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && nbatch > 0 && ksplit > 0);
  CURLA_REQUIRE(ksplit == 1 || (!bias && !mask && !relu));
  GemmArgs g;
  g.A = A, g.B = B, g.C = C, g.bias = bias, g.mask = mask;
  g.M = M, g.N = N, g.K = K, g.lda = lda, g.ldb = ldb, g.ldc = ldc, g.ldmask = ldmask;
  g.sA = strideA, g.sB = strideB, g.sC = strideC, g.sBias = strideBias, g.sMask = strideMask, g.sSplit = split_stride;
  g.nbatch = nbatch, g.ksplit = ksplit;
  g.alpha = alpha, g.relu = relu;
  g.vecA = (lda % 4 == 0) && (strideA % 4 == 0) && aligned16(A);
  g.vecB = (ldb % 4 == 0) && (strideB % 4 == 0) && aligned16(B);
  g.nptr = 0;
  g.colsum = nullptr, g.sColsum = g.sColsum2 = 0;
  g.nb_inner = nbatch, g.sA2 = g.sB2 = g.sC2 = g.sBias2 = g.sMask2 = 0;
  return gemm_launch(g, a_kmajor, b_kmajor, static_cast<hipStream_t>(stream));
}

int curla_gemm_small_shape(int M, int N, int K, int nbatch) { return small_shape(M, N, K, nbatch) ? 1 : 0; }

int curla_gemm_colsum(const float* A, int a_kmajor, int lda, long long strideA, const float* B, int b_kmajor, int ldb,
                      long long strideB, float* C, int ldc, long long strideC, int M, int N, int K, int nbatch,
                      float* colsum, long long strideColsum, void* stream) {
  CURLA_REQUIRE(A && B && C && colsum && M > 0 && N > 0 && K > 0 && nbatch > 0);
  GemmArgs g;
  g.A = A, g.B = B, g.C = C, g.bias = nullptr, g.mask = nullptr;
  g.M = M, g.N = N, g.K = K, g.lda = lda, g.ldb = ldb, g.ldc = ldc, g.ldmask = 0;
  g.sA = strideA, g.sB = strideB, g.sC = strideC, g.sBias = 0, g.sMask = 0, g.sSplit = 0;
  g.nbatch = nbatch, g.ksplit = 1;
  g.alpha = 1.f, g.relu = 0;
  g.vecA = (lda % 4 == 0) && (strideA % 4 == 0) && aligned16(A);
  g.vecB = (ldb % 4 == 0) && (strideB % 4 == 0) && aligned16(B);
  g.nptr = 0;
  g.nb_inner = nbatch, g.sA2 = g.sB2 = g.sC2 = g.sBias2 = g.sMask2 = 0;
  g.colsum = colsum, g.sColsum = strideColsum, g.sColsum2 = 0;
  return gemm_launch(g, a_kmajor, b_kmajor, static_cast<hipStream_t>(stream));
}

static void gemm_args_plain(GemmArgs& g, const float* A, int lda, long long sA, const float* B, int ldb, long long sB,
                            float* C, int ldc, long long sC, int M, int N, int K, int nbatch) {
  g.A = A, g.B = B, g.C = C, g.bias = nullptr, g.mask = nullptr;
  g.M = M, g.N = N, g.K = K, g.lda = lda, g.ldb = ldb, g.ldc = ldc, g.ldmask = 0;
  g.sA = sA, g.sB = sB, g.sC = sC, g.sBias = 0, g.sMask = 0, g.sSplit = 0;
  g.nbatch = nbatch, g.ksplit = 1, g.kchunk = 0;
  g.alpha = 1.f, g.relu = 0, g.stream_c = 0;
  g.vecA = (lda % 4 == 0) && (sA % 4 == 0) && aligned16(A);
  g.vecB = (ldb % 4 == 0) && (sB % 4 == 0) && aligned16(B);
  g.nptr = 0;
  g.nb_inner = nbatch, g.sA2 = g.sB2 = g.sC2 = g.sBias2 = g.sMask2 = 0;
  g.colsum = nullptr, g.sColsum = g.sColsum2 = 0;
}

int curla_linear_bwd(const float* dy, long long stride_dy, const float* x, long long stride_x, const float* W,
                     long long stride_W, const float* mask, long long stride_mask, float* dW, long long stride_dW,
                     float* db, long long stride_db, float* dx, long long stride_dx, int B, int N, int K, int nbatch,
                     void* stream) {
  CURLA_REQUIRE(dy && x && W && dW && dx && B > 0 && N > 0 && K > 0 && nbatch > 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  GemmArgs g1, g2;
  // dW[n][k] = sum_b dy[b][n] x[b][k]: both operands k-major (k = batch row)
  gemm_args_plain(g1, dy, N, stride_dy, x, K, stride_x, dW, K, stride_dW, N, K, B, nbatch);
  g1.colsum = db, g1.sColsum = stride_db;
  // dx[b][k] = sum_n dy[b][n] W[n][k], zero where mask <= 0: W k-major
  gemm_args_plain(g2, dy, N, stride_dy, W, K, stride_W, dx, K, stride_dx, B, K, N, nbatch);
  g2.mask = mask, g2.ldmask = K, g2.sMask = stride_mask;
  GemmPlan p1, p2;
  const bool split = curla_opt(kOptLinearBwd) == 1;  // (option linear_bwd, options.h)
  // the two products' 128 x 64 tiles are counted together: one launch holds both
  const bool both_big = !split && big_tile_ok(g1) && big_tile_ok(g2) && big_tile_count(g1) + big_tile_count(g2) >= curla_cu_count();
  int rc = gemm_plan(g1, 1, 1, p1, both_big);
  if (rc != CURLA_OK) return rc;
  rc = gemm_plan(g2, 0, 1, p2, both_big);
  if (rc != CURLA_OK) return rc;
  const int T1 = p1.small ? 16 : 0, T2 = p2.small ? 16 : 0;
  const int gx1 = (g1.N + (T1 ? T1 : p1.tbn) - 1) / (T1 ? T1 : p1.tbn), gy1 = (g1.M + (T1 ? T1 : p1.tbm) - 1) / (T1 ? T1 : p1.tbm);
  const int gx2 = (g2.N + (T2 ? T2 : p2.tbn) - 1) / (T2 ? T2 : p2.tbn), gy2 = (g2.M + (T2 ? T2 : p2.tbm) - 1) / (T2 ? T2 : p2.tbm);
  const long long n1 = (long long)gx1 * gy1 * nbatch, n2 = (long long)gx2 * gy2 * nbatch;
  if (!split && n1 + n2 < (1LL << 30)) {
    if (p1.small && p2.small && !p1.wide && !p2.wide) {
      hipLaunchKernelGGL((gemm_small_pair_kernel<4, 8>), dim3((unsigned)(n1 + n2)), dim3(256), 0, st, g1, g2, (int)n1, gx1,
                         gy1, gx2, gy2);
      return curla_launch_status();
    }
    if (!p1.small && !p2.small && p1.fast && p2.fast && p1.tbm == 128 && p2.tbm == 128) {
      // (the data gradient's tiles run over k = N, the weight gradient's over k = the batch rows: the longer ones first)
      if (g2.K >= g1.K)
        hipLaunchKernelGGL((gemm_pair_kernel<128, 64, 128, 64, true, true, 512>), dim3((unsigned)(n1 + n2)), dim3(512), 0, st, g1,
                           g2, (int)n1, gx1, gy1, gx2, gy2, (int)n2);
      else
        hipLaunchKernelGGL((gemm_pair_kernel<128, 64, 128, 64, true, false, 512>), dim3((unsigned)(n1 + n2)), dim3(512), 0, st, g1,
                           g2, (int)n1, gx1, gy1, gx2, gy2, (int)n2);
      return curla_launch_status();
    }
    if (!p1.small && !p2.small && p1.fast && p2.fast && p1.b3 == p2.b3) {
#define CURLA_PAIR(TM1, TN1, TM2, TN2)                                                                                   \
  if (p1.tbm == TM1 && p1.tbn == TN1 && p2.tbm == TM2 && p2.tbn == TN2) {                                               \
    if (p1.b3)                                                                                                           \
      hipLaunchKernelGGL((gemm_pair_kernel<TM1, TN1, TM2, TN2, true>), dim3((unsigned)(n1 + n2)), dim3(256), 0, st, g1,  \
                         g2, (int)n1, gx1, gy1, gx2, gy2, (int)n2);                                                      \
    else                                                                                                                 \
      hipLaunchKernelGGL((gemm_pair_kernel<TM1, TN1, TM2, TN2>), dim3((unsigned)(n1 + n2)), dim3(256), 0, st, g1, g2,    \
                         (int)n1, gx1, gy1, gx2, gy2, (int)n2);                                                          \
    return curla_launch_status();                                                                                        \
  }
      // (the shapes the twin-Q and the actor MLPs produce at batch 512 and 1024, hidden 1024)
      CURLA_PAIR(64, 64, 64, 32)
      CURLA_PAIR(64, 32, 32, 32)
      CURLA_PAIR(64, 64, 64, 64)
      CURLA_PAIR(64, 32, 64, 32)
#undef CURLA_PAIR
    }
  }
  rc = gemm_launch(g1, 1, 1, st);
  if (rc != CURLA_OK) return rc;
  return gemm_launch(g2, 0, 1, st);
}

int curla_gemm_nested(const float* A, int a_kmajor, int lda, long long strideA, long long strideA2, const float* B,
                      int b_kmajor, int ldb, long long strideB, long long strideB2, float* C, int ldc, long long strideC,
                      long long strideC2, int M, int N, int K, int nbatch, int nbatch2, float alpha, const float* bias,
                      long long strideBias, long long strideBias2, int relu, const float* mask, int ldmask,
                      long long strideMask, long long strideMask2, void* stream) {
  CURLA_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && nbatch > 0 && nbatch2 > 0);
  GemmArgs g;
  g.A = A, g.B = B, g.C = C, g.bias = bias, g.mask = mask;
  g.M = M, g.N = N, g.K = K, g.lda = lda, g.ldb = ldb, g.ldc = ldc, g.ldmask = ldmask;
  g.sA = strideA, g.sB = strideB, g.sC = strideC, g.sBias = strideBias, g.sMask = strideMask, g.sSplit = 0;
  g.nbatch = nbatch * nbatch2, g.ksplit = 1;
  g.alpha = alpha, g.relu = relu;
  g.vecA = (lda % 4 == 0) && (strideA % 4 == 0) && (strideA2 % 4 == 0) && aligned16(A);
  g.vecB = (ldb % 4 == 0) && (strideB % 4 == 0) && (strideB2 % 4 == 0) && aligned16(B);
  g.nptr = 0;
  g.colsum = nullptr, g.sColsum = g.sColsum2 = 0;
  g.nb_inner = nbatch, g.sA2 = strideA2, g.sB2 = strideB2, g.sC2 = strideC2, g.sBias2 = strideBias2, g.sMask2 = strideMask2;
  return gemm_launch(g, a_kmajor, b_kmajor, static_cast<hipStream_t>(stream));
}

int curla_gemm_multi(int nprob, const float* const* A, const float* const* B, float* const* C, int lda, int ldb, int ldc,
                     int M, int N, int K, int ksplit, long long split_stride, void* stream) {
  CURLA_REQUIRE(nprob > 0 && nprob <= 4 && A && B && C && M > 0 && N > 0 && K > 0 && ksplit > 0);
  GemmArgs g;
  g.A = g.B = nullptr, g.C = nullptr, g.bias = nullptr, g.mask = nullptr;
  g.M = M, g.N = N, g.K = K, g.lda = lda, g.ldb = ldb, g.ldc = ldc, g.ldmask = 0;
  g.sA = g.sB = g.sC = g.sBias = g.sMask = 0, g.sSplit = split_stride;
  g.nbatch = nprob, g.ksplit = ksplit;
  g.alpha = 1.f, g.relu = 0;
  g.vecA = (lda % 4 == 0), g.vecB = (ldb % 4 == 0);
  g.nptr = nprob;
  g.colsum = nullptr, g.sColsum = g.sColsum2 = 0;
  g.nb_inner = nprob, g.sA2 = g.sB2 = g.sC2 = g.sBias2 = g.sMask2 = 0;
  for (int i = 0; i < nprob; ++i) {
    CURLA_REQUIRE(A[i] && B[i] && C[i]);
    g.Ap[i] = A[i], g.Bp[i] = B[i], g.Cp[i] = C[i];
    g.vecA = g.vecA && aligned16(A[i]);
    g.vecB = g.vecB && aligned16(B[i]);
  }
  for (int i = nprob; i < 4; ++i) g.Ap[i] = g.Bp[i] = nullptr, g.Cp[i] = nullptr;
  return gemm_launch(g, 0, 0, static_cast<hipStream_t>(stream));
}

int curla_fc_fwd_multi(int nprob, const float* const* x, const float* const* W, float* const* partial, int B, int F, int K,
                       int nsplit, long long split_stride, int x_blocked, void* stream) {
  CURLA_REQUIRE(nprob > 0 && nprob <= 4 && x && W && partial && B > 0 && F > 0 && K > 0 && nsplit > 0);
  // the instantiated shapes: 50..64 even features (50: three tiles + two features on FMAs), whole 128-row groups, 32-wide
  // k-slices, at least one slice per split, 32-bit byte offsets; everything else: CURLA_ERR_UNSUPPORTED (the caller
  // takes curla_gemm_multi)
  if (F < 50 || F > 64 || F % 2 != 0 || B % 128 != 0 || K % 32 != 0 || nsplit > K / 32 ||
      (long long)B * K * 4 >= (1LL << 31) || (long long)F * K * 4 >= (1LL << 31) || split_stride % 2 != 0)
    return CURLA_ERR_UNSUPPORTED;
  FcFwdArgs g;
  for (int i = 0; i < 4; ++i) {
    g.x[i] = i < nprob ? x[i] : nullptr, g.W[i] = i < nprob ? W[i] : nullptr, g.out[i] = i < nprob ? partial[i] : nullptr;
    if (i < nprob) {
      CURLA_REQUIRE(x[i] && W[i] && partial[i]);
      if (!aligned16(x[i]) || !aligned16(W[i]) || (reinterpret_cast<uintptr_t>(partial[i]) & 7)) return CURLA_ERR_UNSUPPORTED;
    }
  }
  g.nprob = nprob, g.B = B, g.F = F, g.K = K, g.nsplit = nsplit, g.nrg = B / 128, g.split_stride = split_stride;
  g.blocked = x_blocked ? 1 : 0;
  const int grid = nprob * g.nrg * nsplit;
  const size_t lds = (size_t)2 * 64 * kFcPitch * sizeof(float);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define CURLA_FC_FWD(NTF, NTL, B3F)                                                                                       \
  do {                                                                                                                    \
    if (curla_set_dyn_lds(reinterpret_cast<const void*>(fc_fwd_kernel<NTF, NTL, B3F>), lds) != CURLA_OK) return CURLA_ERR_LAUNCH; \
    hipLaunchKernelGGL((fc_fwd_kernel<NTF, NTL, B3F>), dim3(grid), dim3(256), lds, st, g);                                \
  } while (0)
  const bool b3 = gemm_mfma_opt() != 1;  // option gemm_mfma: f32 keeps the f32-input MFMA
  if (F == 50) {
    if (b3) CURLA_FC_FWD(3, 2, true); else CURLA_FC_FWD(3, 2, false);
  } else {
    if (b3) CURLA_FC_FWD(4, 0, true); else CURLA_FC_FWD(4, 0, false);
  }
#undef CURLA_FC_FWD
  return curla_launch_status();
}

static int fc_bwd_check(const float* dz, const float* big, const float* out, int B, int F, int K) {
  CURLA_REQUIRE(dz && big && out && B > 0 && F > 0 && K > 0);
  if (F > 4 * kFcMaxSteps || (K % 4) != 0 || !aligned16(big) || !aligned16(out)) return CURLA_ERR_UNSUPPORTED;
  return CURLA_OK;
}

int curla_fc_dx(const float* dz, const float* W, const float* mask, float* dx, int B, int F, int K, void* stream) {
  int rc = fc_bwd_check(dz, W, dx, B, F, K);
  if (rc != CURLA_OK) return rc;
  if (mask && !aligned16(mask)) return CURLA_ERR_UNSUPPORTED;
  FcBwdArgs g;
  g.dz = dz, g.W = W, g.mask = mask, g.out = dx, g.B = B, g.F = F, g.K = K;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((K + 63) / 64);
#define CURLA_FC_DX(KS) hipLaunchKernelGGL(fc_dx_kernel<KS>, grid, dim3(256), 0, st, g)
  const int ks = (F + 3) / 4;
  if (ks <= 4) CURLA_FC_DX(4);
  else if (ks <= 8) CURLA_FC_DX(8);
  else if (ks <= 13) CURLA_FC_DX(13);
  else CURLA_FC_DX(16);
#undef CURLA_FC_DX
  return curla_launch_status();
}

// a separate launch for the deferred LayerNorm sums, for the shapes whose fc backward does not take the one-pass kernel
__global__ void ln_reduce_kernel(LnReduce r) { ln_reduce_block(r); }

static int ln_reduce_args(LnReduce& r, const float* partial, int nparts, int F, float* dgamma, float* dbeta,
                          float* dbias) {
  r.partial = partial, r.nparts = nparts, r.F = F, r.dgamma = dgamma, r.dbeta = dbeta, r.dbias = dbias;
  r.xpartial = nullptr, r.xparts = r.xlen = 0, r.xout = nullptr;
  if (!partial) return CURLA_OK;
  CURLA_REQUIRE(nparts > 0 && dgamma && dbeta);
  return CURLA_OK;
}

static int fc_dw_impl(const float* dz, const float* x, float* dW, int B, int F, int K, void* stream, const LnReduce& lnr) {
  int rc = fc_bwd_check(dz, x, dW, B, F, K);
  if (rc != CURLA_OK) return rc;
  FcBwdArgs g;
  g.dz = dz, g.W = x, g.mask = nullptr, g.out = dW, g.B = B, g.F = F, g.K = K;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (F == 50 && B % 16 == 0 && (long long)B * K * 4 < (1LL << 31)) {  // the one-pass kernel's weight-gradient half
    FcBwdArgs gx;
    gx.dz = dz, gx.W = nullptr, gx.mask = x, gx.out = nullptr, gx.B = B, gx.F = F, gx.K = K;
    const size_t lds1 = (size_t)4 * 4 * 4 * 64 * sizeof(f32x4);
    if (curla_set_dyn_lds(reinterpret_cast<const void*>(fc_bwd_kernel<13, 3, 2, false>), lds1) != CURLA_OK)
      return CURLA_ERR_LAUNCH;
    hipLaunchKernelGGL((fc_bwd_kernel<13, 3, 2, false>), dim3((K + 63) / 64 + (lnr.partial ? 1 : 0)), dim3(256), lds1, st,
                       gx, dW, lnr);
    return curla_launch_status();
  }
  if (lnr.partial) hipLaunchKernelGGL(ln_reduce_kernel, dim3(1), dim3(256), 0, st, lnr);
  const dim3 grid((K + 63) / 64);
  const int nt = (F + 15) / 16;
  const size_t lds = (size_t)4 * nt * 4 * 64 * sizeof(f32x4);  // 16 KB per feature tile
#define CURLA_FC_DW(NT)                                                                                               \
  do {                                                                                                                \
    if (curla_set_dyn_lds(reinterpret_cast<const void*>(fc_dw_kernel<NT>), (size_t)4 * NT * 4 * 64 * sizeof(f32x4)) != \
        CURLA_OK) /* once per device, thread-safe */                                                                  \
      return CURLA_ERR_LAUNCH;                                                                                        \
    hipLaunchKernelGGL(fc_dw_kernel<NT>, grid, dim3(256), lds, st, g);                                                \
  } while (0)
  if (nt == 1) CURLA_FC_DW(1);
  else if (nt == 2) CURLA_FC_DW(2);
  else if (nt == 3) CURLA_FC_DW(3);
  else CURLA_FC_DW(4);
#undef CURLA_FC_DW
  return curla_launch_status();
}

int curla_fc_dw(const float* dz, const float* x, float* dW, int B, int F, int K, void* stream) {
  return fc_dw_impl(dz, x, dW, B, F, K, stream, LnReduce{});
}

int curla_fc_dw_ln(const float* dz, const float* x, float* dW, int B, int F, int K, const float* ln_partial, int nparts,
                   float* dgamma, float* dbeta, float* dbias_in, void* stream) {
  LnReduce r;
  int rc = ln_reduce_args(r, ln_partial, nparts, F, dgamma, dbeta, dbias_in);
  if (rc != CURLA_OK) return rc;
  return fc_dw_impl(dz, x, dW, B, F, K, stream, r);
}

static int fc_bwd_impl(const float* dz, const float* W, const float* x, float* dx, float* dW, int B, int F, int K,
                       void* stream, const LnReduce& lnr) {
  int rc = fc_bwd_check(dz, W, dx, B, F, K);
  if (rc != CURLA_OK) return rc;
  if ((rc = fc_bwd_check(dz, x, dW, B, F, K)) != CURLA_OK) return rc;
  FcBwdArgs gx;
  gx.dz = dz, gx.W = W, gx.mask = x, gx.out = dx, gx.B = B, gx.F = F, gx.K = K;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nblk = (K + 63) / 64;
  const int ks = (F + 3) / 4, nt = (F + 15) / 16;
  // the one-pass kernel is instantiated for 49..52 features, whole 16-row tiles and matrices below 2 GB (32-bit buffer
  // offsets); anything else takes the two streaming kernels
  if (ks != 13 || nt != 4 || B % 16 != 0 || (long long)B * K * 4 >= (1LL << 31)) {  // (i.e. F = 49..52)
    if ((rc = curla_fc_dx(dz, W, x, dx, B, F, K, stream)) != CURLA_OK) return rc;
    return fc_dw_impl(dz, x, dW, B, F, K, stream, lnr);
  }
  const size_t lds = (size_t)4 * 4 * 4 * 64 * sizeof(f32x4);  // (the 3 + 2 form needs less: 48 KB + 2 KB)
  // (once per kernel and device, thread-safe: curla_set_dyn_lds)
  if (F == 50) {  // the default feature width: 3 tiles on the matrix pipe + 2 features on VALU FMAs
    if (curla_set_dyn_lds(reinterpret_cast<const void*>(fc_bwd_kernel<13, 3, 2, true>), lds) != CURLA_OK) return CURLA_ERR_LAUNCH;
    hipLaunchKernelGGL((fc_bwd_kernel<13, 3, 2, true>), dim3(nblk + (lnr.partial ? 1 : 0)), dim3(256), lds, st, gx, dW, lnr);
  } else {
    if (curla_set_dyn_lds(reinterpret_cast<const void*>(fc_bwd_kernel<13, 4, 0, true>), lds) != CURLA_OK) return CURLA_ERR_LAUNCH;
    hipLaunchKernelGGL((fc_bwd_kernel<13, 4, 0, true>), dim3(nblk + (lnr.partial ? 1 : 0)), dim3(256), lds, st, gx, dW, lnr);
  }
  return curla_launch_status();
}

int curla_fc_bwd(const float* dz, const float* W, const float* x, float* dx, float* dW, int B, int F, int K,
                 void* stream) {
  return fc_bwd_impl(dz, W, x, dx, dW, B, F, K, stream, LnReduce{});
}

int curla_fc_bwd_ln(const float* dz, const float* W, const float* x, float* dx, float* dW, int B, int F, int K,
                    const float* ln_partial, int nparts, float* dgamma, float* dbeta, float* dbias_in, void* stream) {
  LnReduce r;
  int rc = ln_reduce_args(r, ln_partial, nparts, F, dgamma, dbeta, dbias_in);
  if (rc != CURLA_OK) return rc;
  return fc_bwd_impl(dz, W, x, dx, dW, B, F, K, stream, r);
}

int curla_fc_bwd_ln2(const float* dz, const float* W, const float* x, float* dx, float* dW, int B, int F, int K,
                     const float* ln_partial, int nparts, float* dgamma, float* dbeta, float* dbias_in,
                     const float* extra_partial, int extra_parts, int extra_len, float* extra_out, void* stream) {
  CURLA_REQUIRE(ln_partial && extra_partial && extra_out && extra_parts > 0 && extra_len > 0);
  LnReduce r;
  int rc = ln_reduce_args(r, ln_partial, nparts, F, dgamma, dbeta, dbias_in);
  if (rc != CURLA_OK) return rc;
  r.xpartial = extra_partial, r.xparts = extra_parts, r.xlen = extra_len, r.xout = extra_out;
  return fc_bwd_impl(dz, W, x, dx, dW, B, F, K, stream, r);
}

int curla_splitk_reduce(const float* partial, int nsplit, long long split_stride, int M, int N, int ldp, float* C,
                        int ldc, const float* bias, int relu, void* stream) {
  CURLA_REQUIRE(partial && C && nsplit > 0 && M > 0 && N > 0);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((M * N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     partial, nsplit, split_stride, M, N, ldp, C, ldc, bias, relu);
  return curla_launch_status();
}

}  // extern "C"
