// 3x3 convolution kernels of the CNNEncoder hot path, gfx950.
//
// Reference semantics: encoder.py:54-63 (Conv2d k=3, first layer stride 2, the
// rest stride 1, no padding) + encoder.py:77-90 (obs/255, relu(conv)).  The
// backward kernels are the autograd of those lines.
//
// Layout: activations are NHWC fp32 in HBM ([B][H][W][32]); conv weights stay
// in the reference OIHW layout (they are nn.Parameters shared with Adam) and
// are re-gathered into MFMA operand registers at kernel start.  Every inner
// product runs on the exact-f32 matrix pipe (v_mfma_f32_16x16x4_f32), which has
// the same 157 TFLOP/s roof as the fp32 vector pipe but needs one operand VGPR
// per lane instead of 2 per FMA.  GEMM view of the forward:
//   D[cout (16 per wave)][pixel pair (16 per tile)] += U[cout][k] * V[k][pixel pair],  k = (row tap, cin),
// with the x direction in 1-D Winograd F(2,3) form (4 products per two outputs).  A 256-thread workgroup
// stages a band of input rows in LDS (pixel stride padded 32 -> 36 floats); its 4 waves are 2 output-channel
// halves x 2 tile slots.  Two such workgroups share a CU (<= 80 KB LDS, <= 256 VGPRs each): while one stages
// or stores, the other's waves keep the matrix pipe fed.  What bounds the tile loops is VALU issue time (a VALU
// instruction and an MFMA cannot issue in the same cycle), so everything in them is counted in instructions:
// packed fp32 adds for the Winograd transform, division-free pixel walks, epilogues deferred into the next tile.
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.h"

namespace {

constexpr int kLdsPix = 36;   // floats per pixel in LDS (32 channels + 4 pad: conflict-free b128 reads)
constexpr int kWStride = 289; // LDS row stride while re-gathering the OIHW weights
constexpr int kMaxPf = 18;    // float4 staging registers per thread (band <= 256*18/8 = 576 pixels, 81 KB of LDS)

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

// 128 bytes of zeros in HBM: where a lane with nothing to multiply points its operand load (a select on the loaded
// VALUE would make the wave wait for the load right behind it instead of one or more k-steps later)
__device__ float g_zero_px[32];

// Timing-only ablations for tools/kbench.py (results are wrong when set); compiled out of the product library.
#ifdef CURLA_ABLATE
int g_ablate = 0;
#define ABL(bit) (a.dbg & (bit))
#define ABL_HOST g_ablate
#else
#define ABL(bit) 0
#define ABL_HOST 0
#endif

struct ConvS1Args {
  const float* in;   // [B][Hs][Ws][32]
  const float* w;    // OIHW [32][32][3][3]
  const float* aux;  // FWD: bias[32]; DGRAD: activation below, [B][Ho][Wo][32] (ReLU mask)
  float* out;        // [B][Ho][Wo][32]
  int B, Hs, Ws, Ho, Wo, pad, th, nbands;
  int h1;  // height of band 0 (>= th; the other bands are th rows, the last one what is left)
  int qstep, rstep;  // 32 = qstep * PW + rstep, PW = pixel pairs per output row
  int dbg;
  // forward only: a second problem of the same geometry with its own weights (B2 samples; 0 = none).  Its items
  // follow the first problem's, so a workgroup re-builds its weight registers at most once.
  const float* in2;
  const float* w2;
  const float* aux2;
  float* out2;
  int B2;
};

// ---------------------------------------------------------------------------
// stride-1 32->32 conv: forward (bias+ReLU) and data-gradient (full correlation
// with the flipped/transposed filter, ReLU mask of the layer below fused in).
// The x direction uses Winograd F(2,3) (4 MFMA products per two outputs instead of 6): the input
// transform is 4 vector adds on the LDS window right before the MFMAs, the output transform 4 adds
// per pair; the y direction and the channel sums are the plain accumulation.
// ---------------------------------------------------------------------------
// (the kernel body as a device function of (block id, block count): conv_s1_kernel runs it over the whole grid,
// bwd_s1_kernel over the second part of a grid whose first part is the weight-gradient kernel's)
template <int MODE>
__device__ __forceinline__ void conv_s1_body(const ConvS1Args& a, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int WT = a.Wo + 2;

  // ---- weights -> registers (A operand).  A wave owns ONE 16-channel output
  // tile (mt) for all its pixel tiles, so it keeps 72 weight registers, not 144;
  // wave pairs (2p, 2p+1) share pixel tiles.  k-step (q,e) of tap t covers
  // cin = 16q + 4kq' + e over the four lane groups kq'.
  const int mt = wave & 1, tslot = wave >> 1;
  float wu[3][4][8];
  f32x4 bias4 = {0, 0, 0, 0};
  auto load_weights = [&](const float* wsrc, const float* aux) {
    {
      // 9216 weights = 2304 float4, 9 per thread, all in flight at once (OIHW rows are 288 floats = 72 float4)
      f32x4 wv[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) wv[u] = reinterpret_cast<const f32x4*>(wsrc)[tid + u * 256];
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const int i4 = tid + u * 256;
        const int r = i4 / 72, c = (i4 - r * 72) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[r * kWStride + c + e] = wv[u][e];
      }
    }
    __syncthreads();
    // 1-D Winograd F(2,3) along x: two adjacent outputs share one 4-pixel window and need 4 products
    // per (row tap, cin) instead of 6.  Filter transform per row tap dy (g0,g1,g2 = the three x taps):
    //   U0 = g0, U1 = (g0+g1+g2)/2, U2 = (g0-g1+g2)/2, U3 = g2
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int co = mt * 16 + li;
        const int ci = 16 * (s >> 2) + 4 * kq + (s & 3);
        float gx[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int t = dy * 3 + dx;
          gx[dx] = (MODE == MODE_FWD) ? lds[co * kWStride + ci * 9 + t] : lds[ci * kWStride + co * 9 + (8 - t)];
        }
        wu[dy][0][s] = gx[0];
        wu[dy][1][s] = 0.5f * (gx[0] + gx[1] + gx[2]);
        wu[dy][2][s] = 0.5f * (gx[0] - gx[1] + gx[2]);
        wu[dy][3][s] = gx[2];
      }
    if (MODE == MODE_FWD) bias4 = *reinterpret_cast<const f32x4*>(aux + mt * 16 + 4 * kq);
    __syncthreads();
  };
  load_weights(a.w, a.aux);

  f32x2 wt = {0, 0};  // winograd_bt_pk's temporary, live for the whole kernel (see common.h)
  // The items of one problem, as a lambda so that the forward's optional second problem (its own weights) is a second
  // loop after a weight reload, not a branch inside the loop (which costs the forward kernel 34 spilled registers).
  auto run = [&](const float* in_base, const float* aux_base, float* out_base, int Bc, int item, int item_end,
                 int item0) {
  for (; item < item_end; item += nblk) {
    const int local = item - item0;
    const int band = local / Bc, b = local - band * Bc;  // band-major: every workgroup sees every band size
    const int y0 = band == 0 ? 0 : a.h1 + (band - 1) * a.th;
    const int tha = band == 0 ? a.h1 : min(a.th, a.Ho - y0);
    // ---- stage the band: all loads in flight at once, then the LDS writes.  The
    // exposed latency is covered by the CU's second workgroup, whose phases are
    // not synchronised with this one's.
    {
      const int n4 = (tha + 2) * WT * 8;
      f32x4 pf[kMaxPf];
      if (MODE == MODE_FWD) {
        // the band (tha+2 full rows) is one contiguous run of HBM: no index arithmetic
        const f32x4* src = reinterpret_cast<const f32x4*>(in_base + ((size_t)(b * a.Hs + y0) * a.Ws) * 32);
#pragma unroll
        for (int u = 0; u < kMaxPf; ++u) {
          const int f = tid + u * 256;
          f32x4 v = {0, 0, 0, 0};
          if (f < n4 && !(ABL(1) && item != bid)) v = src[f];
          pf[u] = v;
        }
      } else {
        // zero-padded band: walk (row, col) incrementally (32 pixels per step), no divisions.  Buffer loads whose
        // descriptor spans exactly this sample's image: rows above / below it are out of range and read zeros by
        // themselves (a negative offset is a huge unsigned one); columns left / right of it would land in a
        // neighbouring row, so their lanes are pointed past the buffer.  No bounds branches, no 64-bit lane addresses.
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(in_base + (size_t)b * a.Hs * a.Ws * 32), (short)0, a.Hs * a.Ws * 128, 0x00020000);
        int r = (tid >> 3) / WT, c = (tid >> 3) - r * WT;
        const int ch = tid & 7;
#pragma unroll
        for (int u = 0; u < kMaxPf; ++u) {
          const int f = tid + u * 256;
          const int sy = y0 + r - a.pad, sx = c - a.pad;
          const bool ok = f < n4 && (unsigned)sx < (unsigned)a.Ws && !(ABL(1) && item != bid);
          const unsigned voff = ok ? (unsigned)(((sy * a.Ws + sx) * 32 + ch * 4) * 4) : 0x80000000u;
          pf[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, voff, 0, 0));
          c += 32;
          if (WT >= 32) {  // wave-uniform: at most one row wrap per 32-pixel step
            const bool wrap = c >= WT;
            c = wrap ? c - WT : c;
            r = wrap ? r + 1 : r;
          } else {
            while (c >= WT) c -= WT, ++r;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kMaxPf; ++u) {
        const int f = tid + u * 256;
        if (f < n4 && !(ABL(2) && item != bid))
          *reinterpret_cast<f32x4*>(lds + (f >> 3) * kLdsPix + (f & 7) * 4) = pf[u];
      }
    }
    __syncthreads();

    // a lane owns a horizontal PAIR of output pixels (x0 = 2j, 2j+1); a tile is 16 pairs
    const int PW = (a.Wo + 1) >> 1;            // pairs per output row (the last one is half valid when Wo is odd)
    const int npairs = tha * PW;
    const int ntiles = (npairs + 15) >> 4;
    int ty = (tslot * 16 + li) / PW, j = (tslot * 16 + li) - ty * PW;
    // The output transform + bias/ReLU (mask) + stores of a tile are deferred into the second half-step of the
    // NEXT tile: there they issue between that tile's MFMAs instead of waiting for the matrix pipe to drain
    // with nothing else to do (the epilogue at the tile's own end cost 17 % of the kernel).
    // Two accumulator sets alternate between consecutive tiles, so the deferred epilogue reads registers no
    // MFMA of the current tile writes (a copy would wait for the pipe all the same).
    f32x4 accA[4], accB[4], pma = {0, 0, 0, 0}, pmb = {0, 0, 0, 0};
    // per-item (wave-uniform) base pointers: a tile only adds a 32-bit element offset
    float* const out_item = out_base + ((size_t)(b * a.Ho + y0) * a.Wo) * 32;
    const float* const aux_item = aux_base + ((size_t)(b * a.Ho + y0) * a.Wo) * 32;
    const __amdgpu_buffer_rsrc_t raux = __builtin_amdgcn_make_buffer_rsrc(
        (void*)aux_item, (short)0, MODE == MODE_DGRAD ? tha * a.Wo * 128 : 0, 0x00020000);
    int pg = 0;
    bool ppv = false, psecond = false;
    auto epilogue = [&](const f32x4 (&pacc)[4]) {
      if (ppv && !ABL(4)) {
        // output transform A^T m: y(x0) = m0+m1+m2, y(x0+1) = m1-m2-m3
        f32x4 ya = pacc[0] + pacc[1] + pacc[2];
        f32x4 yb = pacc[1] - pacc[2] - pacc[3];
        if (MODE == MODE_FWD) {  // (the bias came in through the accumulators' initial values)
#pragma unroll
          for (int r = 0; r < 4; ++r) ya[r] = fmaxf(ya[r], 0.f), yb[r] = fmaxf(yb[r], 0.f);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) ya[r] = pma[r] > 0.f ? ya[r] : 0.f, yb[r] = pmb[r] > 0.f ? yb[r] : 0.f;
        }
        if (ABL(8) && ya[0] != 12345.678f) return;  // timing only: transforms without the stores
        if (ABL(16)) {  // timing only: each wave writes whole 128-B lines (wrong placement)
          float* q = out_item + pg - mt * 16 - 4 * kq + mt * 32 + 4 * kq;
          *reinterpret_cast<f32x4*>(q) = ya;
          *reinterpret_cast<f32x4*>(q + 16) = yb;
          return;
        }
        // streaming stores: the 64-B half lines a wave writes are not read again by this kernel; keeping them
        // out of the L2's way is worth 6-9 % of the kernel
        if (ABL(32)) {  // timing only: ordinary (L2 write-back) stores
          *reinterpret_cast<f32x4*>(out_item + pg) = ya;
          if (psecond) *reinterpret_cast<f32x4*>(out_item + pg + 32) = yb;
          return;
        }
        __builtin_nontemporal_store(ya, reinterpret_cast<f32x4*>(out_item + pg));
        if (psecond) __builtin_nontemporal_store(yb, reinterpret_cast<f32x4*>(out_item + pg + 32));
      }
    };
    auto tile = [&](f32x4 (&acc)[4], const f32x4 (&pacc)[4], int t, bool have_prev) {
      const bool pv = t * 16 + li < npairs;
      if (!pv) ty = 0, j = 0;
      // (24-bit multiplies: full-rate v_mad_u32_u24 instead of the quarter-rate 32-bit forms; every index here is
      // far below 2^24)
      const float* base = lds + __mul24(__mul24(ty, WT) + 2 * j, kLdsPix) + 4 * kq;
      const int x0 = 2 * j;
      const int g = (__mul24(ty, a.Wo) + x0) * 32 + mt * 16 + 4 * kq;  // element offset inside the item's output band
      const bool second = x0 + 1 < a.Wo;
      f32x4 ma = {0, 0, 0, 0}, mb = {0, 0, 0, 0};
      if (MODE == MODE_DGRAD && !ABL(4)) {  // ReLU mask of the layer below: in flight for a whole tile
        // (buffer loads, no branch: lanes without a pixel point past the item's range and read zeros)
        ma = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(raux, pv ? (unsigned)g * 4u : 0x80000000u, 0, 0));
        mb = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(raux, (pv && second) ? (unsigned)(g + 32) * 4u : 0x80000000u, 0, 0));
      }
      // y(x0) = m0+m1+m2 and y(x0+1) = m1-m2-m3: starting m0 at +bias and m3 at -bias adds the bias to both
      acc[0] = bias4, acc[1] = f32x4{0, 0, 0, 0}, acc[2] = f32x4{0, 0, 0, 0}, acc[3] = -bias4;
      // 6 half-steps (3 row taps x 2 cin halves); the 4 window reads of the next half-step are issued
      // before the 16 MFMAs of the current one
      f32x4 d[2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) d[0][c] = *reinterpret_cast<const f32x4*>(base + c * kLdsPix);
#pragma unroll
      for (int h = 0; h < 6; ++h) {
        const int dy = h >> 1, q = h & 1;
        if (h < 5) {
          const int ndy = (h + 1) >> 1, nq = (h + 1) & 1;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            d[(h + 1) & 1][c] = *reinterpret_cast<const f32x4*>(base + (ndy * WT + c) * kLdsPix + 16 * nq);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 d0 = d[h & 1][0], d1 = d[h & 1][1], d2 = d[h & 1][2], d3 = d[h & 1][3];
        // input transform B^T d = (d0-d2, d1+d2, d2-d1, d1-d3), two channels per packed VALU op
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          f32x2 v0 = {d0[2 * p], d0[2 * p + 1]}, e1 = {d1[2 * p], d1[2 * p + 1]};
          f32x2 v2 = {d2[2 * p], d2[2 * p + 1]}, v3 = {d3[2 * p], d3[2 * p + 1]};
          winograd_bt_pk(v0, e1, v2, v3, wt);
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int e = 2 * p + r;
            acc[0] = mfma16(wu[dy][0][4 * q + e], v0[r], acc[0]);
            acc[1] = mfma16(wu[dy][1][4 * q + e], wt[r], acc[1]);
            acc[2] = mfma16(wu[dy][2][4 * q + e], v2[r], acc[2]);
            acc[3] = mfma16(wu[dy][3][4 * q + e], v3[r], acc[3]);
          }
        }
        if (h == 1 && have_prev) epilogue(pacc);
        __builtin_amdgcn_sched_barrier(0);
      }
      pma = ma, pmb = mb, pg = g, ppv = pv, psecond = second;
      // the lane's pair index advances by 32 per tile step = a.qstep whole rows + a.rstep pairs (host-computed), plus
      // at most one more wrap: five VALU instructions, no branch, whatever the row length
      j += a.rstep, ty += a.qstep;
      const bool wrap = j >= PW;
      j = wrap ? j - PW : j;
      ty = wrap ? ty + 1 : ty;
    };
    int t = tslot, done = 0;
    for (; t < ntiles; t += 4) {
      tile(accA, accB, t, done > 0);
      ++done;
      if (t + 2 < ntiles) {
        tile(accB, accA, t + 2, true);
        ++done;
      }
    }
    if (done & 1)
      epilogue(accA);
    else if (done)
      epilogue(accB);
    __syncthreads();
  }
  };
  const int nitems1 = a.B * a.nbands;
  run(a.in, a.aux, a.out, a.B, bid, nitems1, 0);
  if (MODE == MODE_FWD && a.B2 > 0) {
    // this workgroup's walk k, k+G, k+2G, ... over the concatenated item list continues in the second problem
    const int G = nblk;
    const int first2 = bid + G * ((nitems1 - bid + G - 1) / G);
    load_weights(a.w2, a.aux2);
    run(a.in2, a.aux2, a.out2, a.B2, first2, nitems1 + a.B2 * a.nbands, nitems1);
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void conv_s1_kernel(ConvS1Args a) {
  conv_s1_body<MODE>(a, blockIdx.x, gridDim.x);
}

// The whole stack of stride-1 layers (layers 2..L of the encoder) of up to two minibatches in ONE launch.  With the
// batch sizes multiples of the grid size, workgroup k of the persistent grid processes samples k, k+G, ... of every
// layer (the item walk is band-major over `item % B`), i.e. it OWNS its samples: layer l+1 only reads what the same
// workgroup wrote for layer l, so no other workgroup has to be waited for -- a workgroup-scope fence and a barrier
// between layers make its own stores visible to its own loads, which then come out of the L2 instead of HBM.  Each
// launch saved is 4-9 us at these sizes.
constexpr int kMaxStack = 6;
struct ConvS1StackArgs {
  int nlayers, B, B2, Hs0, Ws0;
  const float* in0;
  const float* in0_2;
  const float* w[kMaxStack];
  const float* bias[kMaxStack];
  float* out[kMaxStack];
  const float* w2[kMaxStack];
  const float* bias2[kMaxStack];
  float* out2[kMaxStack];
  int th[kMaxStack], h1[kMaxStack], nbands[kMaxStack];
};

__global__ __launch_bounds__(256, 2) void conv_s1_stack_kernel(ConvS1StackArgs S) {
  for (int l = 0; l < S.nlayers; ++l) {
    ConvS1Args a;
    a.in = l == 0 ? S.in0 : S.out[l - 1], a.w = S.w[l], a.aux = S.bias[l], a.out = S.out[l];
    a.in2 = l == 0 ? S.in0_2 : S.out2[l - 1], a.w2 = S.w2[l], a.aux2 = S.bias2[l], a.out2 = S.out2[l];
    a.B = S.B, a.B2 = S.B2;
    a.Hs = S.Hs0 - 2 * l, a.Ws = S.Ws0 - 2 * l, a.pad = 0, a.Ho = a.Hs - 2, a.Wo = a.Ws - 2;
    a.th = S.th[l], a.h1 = S.h1[l], a.nbands = S.nbands[l];
    const int PW = (a.Wo + 1) >> 1;
    a.qstep = 32 / PW, a.rstep = 32 - a.qstep * PW;
    a.dbg = 0;
    conv_s1_body<MODE_FWD>(a, blockIdx.x, gridDim.x);
    // this workgroup's outputs of layer l are (only) its own inputs of layer l + 1
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

#include "conv_rw.h"

// Row-walk kernels (conv_rw.h).  Forward: 512-thread workgroups, ONE per CU, persistent over the samples they own;
// 96 KB of LDS hold the transformed filters of both problems of a layer, so the eight waves draw their steps from one
// pool (minibatch one and two together) and nothing but the layer boundary synchronises them.
__global__ __launch_bounds__(512, 2) void conv_rw_fwd_kernel(rw::Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int l = 0; l < A.nlayers; ++l) {
    rw::build_filter<MODE_FWD, 512>(lds, A.p[l][0].w, A.p[l][1].B > 0 ? A.p[l][1].w : nullptr, threadIdx.x);
    __syncthreads();
    rw::run_layer<MODE_FWD, 8>(A.g[l], A.p[l][0], A.p[l][1], lds, blockIdx.x, gridDim.x);
    if (l + 1 < A.nlayers) {
      // this workgroup's outputs of layer l are (only) its own inputs of layer l + 1; the barrier also keeps the
      // filter in LDS until every wave has finished reading it
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
}

// data gradient alone (256-thread workgroups: the form that shares a launch with the weight gradient below)
__global__ __launch_bounds__(256, 2) void conv_rw_dgrad_kernel(rw::Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  rw::build_filter<MODE_DGRAD, 256>(lds, A.p[0][0].w, nullptr, threadIdx.x);
  __syncthreads();
  rw::run_layer<MODE_DGRAD, 4>(A.g[0], A.p[0][0], A.p[0][1], lds, blockIdx.x, gridDim.x);
}

#include "conv1_rw.h"

// first layer from a float NHWC minibatch in row-walk form (conv1_rw.h): 512-thread workgroups, one per CU, persistent
// over the samples they own (8 waves share a sample's steps: small minibatches still fill the SIMDs twice)
template <int C>
__global__ __launch_bounds__(512, 2) void conv1_rw_fwd_kernel(rw::Conv1Args a) {
  rw::conv1_body<C, 8>(a, blockIdx.x, gridDim.x);
}

template <int C>
__global__ __launch_bounds__(512, 2) void wgrad1_rw_kernel(rw::Wgrad1Args a) {
  rw::wgrad1_body<C, 8>(a, blockIdx.x, gridDim.x);
}

#include "conv1_u8_rw.h"

// workgroup shape of the uint8 row-walk forward: two 512-thread workgroups per CU (4 waves per SIMD).  Five 256-thread
// ones (5 waves per SIMD, what ~100 VGPRs allow) were slower inside update(): 402 against 407 update()/s
constexpr int kC1U8Threads = 512, kC1U8PerCU = 2;

// first layer straight from the uint8 ring in row-walk form (conv1_u8_rw.h): 512-thread workgroups, two per CU, equal
// shares of the pool of steps of both minibatches (the second with its own weights: both images sit in LDS)
template <int C>
__global__ __launch_bounds__(kC1U8Threads, kC1U8PerCU) void conv1_u8_rw_fwd_kernel(rw::Conv1U8Args a) {
  __shared__ __attribute__((aligned(16))) float lds_img[2 * rw::conv1_u8_image_floats<C>()];
  rw::conv1_u8_stage_weights<C, kC1U8Threads>(lds_img, a.p[0].w, a.p[0].bias, a.scale, threadIdx.x);
  if (a.p[1].B > 0)
    rw::conv1_u8_stage_weights<C, kC1U8Threads>(lds_img + rw::conv1_u8_image_floats<C>(), a.p[1].w, a.p[1].bias, a.scale,
                                                threadIdx.x);
  __syncthreads();
  if ((a.Ws * C) % 4 == 0)
    rw::conv1_u8_body<C, kC1U8Threads / 64, true>(a, lds_img, blockIdx.x, gridDim.x);
  else
    rw::conv1_u8_body<C, kC1U8Threads / 64, false>(a, lds_img, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------
// first layer: Cin = C (9 or 12 ...), stride 2, input either the uint8 replay
// frames (gather by index + random-crop offsets + /255 fused into the load) or
// a float NCHW tensor in [0,255] (the reference's tensor contract).
// K index of the GEMM is k = dy*KR + (dx*C + c), KR = 3C rounded up to 4; the
// (dx,c) run is contiguous in an HWC row, so one LDS row holds it directly.
// ---------------------------------------------------------------------------
struct Conv1Args {
  const void* src;     // SRC_U8: frames [N][Hs][Ws][C] u8;  SRC_F32: [B][C][Hc][Wc] f32
  const int64_t* idx;  // [B] frame index (u8 source) or nullptr -> b
  const int32_t* h1;   // [B] crop row offset or nullptr -> 0
  const int32_t* w1;   // [B] crop col offset or nullptr -> 0
  const float* w;      // OIHW [32][C][3][3]
  const float* bias;   // [32]
  float* out;          // [B][Ho][Wo][32]
  int B, C, Hs, Ws, Hc, Wc, Ho, Wo, th, nbands;
  float scale;
  int dbg;
  // uint8 forward only: a second minibatch from the same ring with its own weights (B2 samples; 0 = none), whose
  // workgroups follow the first one's in the same launch
  const int64_t* idx2;
  const int32_t* h1_2;
  const int32_t* w1_2;
  const float* w2;
  const float* bias2;
  float* out2;
  int B2;
};

enum { SRC_U8 = 0, SRC_F32 = 1, SRC_NHWC = 2 };  // u8 ring / float NCHW tensor / float NHWC tensor

__device__ __forceinline__ int conv1_row_stride(int Wc, int C) { return ((Wc * C + 3) & ~3) + 4; }
__device__ __forceinline__ bool aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Stage input rows [r0, r0+rows) of sample b's (cropped) image into LDS as
// f32 HWC rows of stride RS, scaled by `scale`.
template <int SRC>
__device__ __forceinline__ void conv1_stage(float* lds, const void* src, const int64_t* idx, const int32_t* h1,
                                            const int32_t* w1, int b, int C, int Hs, int Ws, int Hc, int Wc, int r0,
                                            int rows, int RS, float scale, int tid, int nthreads) {
  const int rowf = Wc * C;
  if (SRC == SRC_U8) {
    const int64_t fi = idx ? idx[b] : b;
    const int oh = h1 ? h1[b] : 0, ow = w1 ? w1[b] : 0;
    const uint8_t* frame = static_cast<const uint8_t*>(src) + (size_t)fi * Hs * Ws * C;
    const int G = (rowf + 3) >> 2;
    for (int i = tid; i < rows * G; i += nthreads) {
      const int r = i / G, g = i - r * G;
      const uint8_t* p = frame + ((size_t)(oh + r0 + r) * Ws + ow) * C + 4 * g;
      // (pointer arithmetic, not an integer round trip: the loads stay global_load, not flat_load)
      const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
      const uint32_t* q = reinterpret_cast<const uint32_t*>(p - sh);
      const uint32_t d0 = q[0];
      const uint32_t d1 = sh ? q[1] : 0u;   // only touch the next dword when the run straddles it
      const uint32_t v = __builtin_amdgcn_alignbyte(d1, d0, sh);
      f32x4 o;
      o[0] = (float)(v & 0xff) * scale;
      o[1] = (float)((v >> 8) & 0xff) * scale;
      o[2] = (float)((v >> 16) & 0xff) * scale;
      o[3] = (float)(v >> 24) * scale;
      *reinterpret_cast<f32x4*>(lds + r * RS + 4 * g) = o;
    }
  } else if (SRC == SRC_NHWC) {
    // float NHWC minibatch (augmented observations): rows are contiguous runs of Wc*C floats
    const float* img = static_cast<const float*>(src) + ((size_t)b * Hc + r0) * rowf;
    if ((rowf & 3) == 0 && aligned16_dev(img)) {
      const int G = rowf >> 2;
      for (int i = tid; i < rows * G; i += nthreads) {
        const int r = i / G, g = i - r * G;
        f32x4 v = *reinterpret_cast<const f32x4*>(img + (size_t)r * rowf + 4 * g);
        *reinterpret_cast<f32x4*>(lds + r * RS + 4 * g) = v * scale;
      }
    } else {
      for (int i = tid; i < rows * rowf; i += nthreads) {
        const int r = i / rowf, e = i - r * rowf;
        lds[r * RS + e] = img[(size_t)r * rowf + e] * scale;
      }
    }
  } else {
    const float* img = static_cast<const float*>(src) + (size_t)b * C * Hc * Wc;
    const int n = rows * rowf;
    for (int i = tid; i < n; i += nthreads) {
      const int x = i % Wc;
      const int t = i / Wc;
      const int r = t % rows, c = t / rows;
      lds[r * RS + x * C + c] = img[((size_t)c * Hc + r0 + r) * Wc + x] * scale;
    }
  }
  // The k-steps of a tap row cover KR = 3C rounded up to 4 values: at an odd crop width the last pixel's run ends at
  // the row's end and its padding value is the float BEHIND the row.  Its weight is zero, but 0 x (whatever bit
  // pattern an earlier kernel left in LDS: NaN, Inf) is NaN, which the ReLU then turns into 0 -- a wrong, finite
  // output.  The slack behind every row is zeroed here (the uint8 path wrote whole groups of four: behind those).
#ifndef CURLA_TEST_NO_SLACK_ZERO  // (defined only by a one-off build that checks the regression test can fail)
  {
    const int first = SRC == SRC_U8 ? (rowf + 3) & ~3 : rowf;
    const int pad = RS - first;  // 4..7 floats
    for (int i = tid; i < rows * pad; i += nthreads) {
      const int r = i / pad, e = i - r * pad;
      lds[r * RS + first + e] = 0.f;
    }
  }
#endif
}

template <int SRC, int C>
__global__ __launch_bounds__(512) void conv1_fwd_kernel(Conv1Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KR = (3 * C + 3) & ~3;
  constexpr int NS = 3 * KR / 4;  // k-steps
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int RS = conv1_row_stride(a.Wc, C);

  constexpr int KQ = KR / 4;  // k-steps per tap row
  for (int i = tid; i < 32 * C * 9; i += 512) lds[i] = a.w[i];
  __syncthreads();
  // k = 4 s + kq of the GEMM is (dy, rr) = (s / KQ, 4 (s % KQ) + kq): KR is a multiple of 4, so the tap row is the
  // same for all lanes of a k-step and a lane's operand sits at a COMPILE-TIME offset (dy, 4 (s % KQ)) from its own
  // base (pixel, kq) -- one address per tile instead of one add per k-step (a VALU instruction is a cycle the f32
  // matrix pipe idles: conv_rw.h)
  float wr[NS][2];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int dy = s / KQ, rr = 4 * (s % KQ) + kq;
    const bool ok = rr < 3 * C;
    const int dx = ok ? rr / C : 0, c = ok ? rr - dx * C : 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) wr[s][mt] = ok ? lds[((mt * 16 + li) * C + c) * 9 + dy * 3 + dx] : 0.f;
  }
  f32x4 bias4[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) bias4[mt] = *reinterpret_cast<const f32x4*>(a.bias + mt * 16 + 4 * kq);
  __syncthreads();

  // persistent: the weight registers above are built once per workgroup, not once per band (a band is ~2 us of
  // tile work at 168x168x12 -- the per-band weight phase was a quarter of the kernel)
  const int nitems = a.B * a.nbands;
  const int qstep = 128 / a.Wo, rstep = 128 - qstep * a.Wo;  // 8 waves x 16 pixels further
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int band = item / a.B, b = item - band * a.B;  // band-major: every workgroup sees every band size
    const int y0 = band * a.th;
    const int tha = min(a.th, a.Ho - y0);
    conv1_stage<SRC>(lds, a.src, a.idx, a.h1, a.w1, b, C, a.Hs, a.Ws, a.Hc, a.Wc, 2 * y0, 2 * tha + 1, RS, a.scale, tid,
                     512);
    __syncthreads();

    const int npix = tha * a.Wo;
    const int ntiles = (npix + 15) >> 4;
    float* const out_item = a.out + ((size_t)(b * a.Ho + y0) * a.Wo) * 32 + 4 * kq;
    int ty = (wave * 16 + li) / a.Wo, x = (wave * 16 + li) - ty * a.Wo;  // walked incrementally: no division per tile
    for (int t = wave; t < ntiles; t += 8) {
      const bool pv = t * 16 + li < npix;
      if (!pv) ty = 0, x = 0;
      const float* base = lds + __mul24(2 * ty, RS) + __mul24(2 * x, C) + kq;
      // the 3 tap rows: RS is a run-time stride, so each row has its own base register; inside a row the k-steps are
      // immediates.  All NS reads of the tile are issued up front (they are independent of the accumulators).
      float bv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) bv[s] = base[(s / KQ) * RS + 4 * (s % KQ)];
      f32x4 acc[2] = {bias4[0], bias4[1]};  // bias through the accumulators' initial values
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        acc[0] = mfma16(wr[s][0], bv[s], acc[0]);
        acc[1] = mfma16(wr[s][1], bv[s], acc[1]);
      }
      if (pv) {
        float* o = out_item + (__mul24(ty, a.Wo) + x) * 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4 v = acc[mt];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o + mt * 16));
        }
      }
      x += rstep, ty += qstep;  // at most one more wrap
      const bool wrap = x >= a.Wo;
      x = wrap ? x - a.Wo : x;
      ty = wrap ? ty + 1 : ty;
    }
    __syncthreads();  // every wave is done with the band before the next one is staged over it
  }
}

// ---------------------------------------------------------------------------
// first layer, uint8 ring source, bytes kept as bytes in LDS (4x less LDS than the float band: a whole
// 76x76x9 crop is 52 KB, so a workgroup takes one sample with no halo re-reads and two workgroups share a
// CU).  Staging is a pure byte copy: 16-byte runs of the (arbitrarily aligned) crop row are rebuilt from
// aligned dword loads with v_alignbyte; u8 -> f32 and the /255 happen when the MFMA B operand is read.
// k = 4s + kq of the GEMM is (dy, rr) = (s / (KR/4), 4 (s % (KR/4)) + kq): since KR is a multiple of 4 the
// tap row dy is wave-uniform per k-step and the byte offset inside the row differs per lane only by kq.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int conv1_row_bytes(int Wc, int C) { return ((Wc * C + 15) & ~15) + 16; }

// One pass of the byte staging, split into its two halves so that a kernel can put other work between the
// loads and the LDS stores: U 16-byte runs per thread, all their dword loads in flight together.
template <int U>
struct Conv1StageRegs {
  uint32_t dw[U][5], sh[U];
  int dst[U];
};

template <int U>
__device__ __forceinline__ void conv1_stage_u8_issue(Conv1StageRegs<U>& rg, const uint8_t* frame, int oh, int ow, int C,
                                                     int Ws, int Wc, int r0, int rows, int RSb, int i0, int tid,
                                                     int nthreads) {
  const int runs = (Wc * C + 15) >> 4;  // 16-byte runs per row
  const int total = rows * runs;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const int i = i0 + tid + k * nthreads;
    const bool ok = i < total;
    const int ic = ok ? i : 0;
    const int r = ic / runs, g = ic - r * runs;
    const uint8_t* p = frame + ((size_t)(oh + r0 + r) * Ws + ow) * C + 16 * g;
    // (pointer arithmetic, not an integer round trip: the loads stay global_load, not flat_load)
    rg.sh[k] = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p - rg.sh[k]);
    rg.dst[k] = ok ? r * RSb + 16 * g : -1;
#pragma unroll
    for (int e = 0; e < 4; ++e) rg.dw[k][e] = ok ? q[e] : 0u;
    rg.dw[k][4] = (ok && rg.sh[k]) ? q[4] : 0u;  // only touch the fifth dword when the run straddles it
  }
}

template <int U>
__device__ __forceinline__ void conv1_stage_u8_commit(const Conv1StageRegs<U>& rg, uint8_t* lds) {
#pragma unroll
  for (int k = 0; k < U; ++k) {
    uint4 o;
    o.x = __builtin_amdgcn_alignbyte(rg.dw[k][1], rg.dw[k][0], rg.sh[k]);
    o.y = __builtin_amdgcn_alignbyte(rg.dw[k][2], rg.dw[k][1], rg.sh[k]);
    o.z = __builtin_amdgcn_alignbyte(rg.dw[k][3], rg.dw[k][2], rg.sh[k]);
    o.w = __builtin_amdgcn_alignbyte(rg.dw[k][4], rg.dw[k][3], rg.sh[k]);
    if (rg.dst[k] >= 0) *reinterpret_cast<uint4*>(lds + rg.dst[k]) = o;
  }
}

__device__ __forceinline__ void conv1_stage_u8(uint8_t* lds, const uint8_t* frames, const int64_t* idx,
                                               const int32_t* h1, const int32_t* w1, int b, int C, int Hs, int Ws,
                                               int Wc, int r0, int rows, int RSb, int tid, int nthreads,
                                               int first_run = 0) {
  const int64_t fi = idx ? idx[b] : b;
  const int oh = h1 ? h1[b] : 0, ow = w1 ? w1[b] : 0;
  const uint8_t* frame = frames + (size_t)fi * Hs * Ws * C;
  const int total = rows * ((Wc * C + 15) >> 4);
  constexpr int U = 4;
  for (int i0 = first_run; i0 < total; i0 += nthreads * U) {
    Conv1StageRegs<U> rg;
    conv1_stage_u8_issue<U>(rg, frame, oh, ow, C, Ws, Wc, r0, rows, RSb, i0, tid, nthreads);
    conv1_stage_u8_commit<U>(rg, lds);
  }
}

// The same byte staging dealt out BY ROW: wave w of NW takes crop rows w, w + NW, ..., lane g the row's g-th 16-byte
// run (lanes past the row idle).  A row's address, its misalignment and its LDS offset are then wave-uniform -- scalar
// registers and scalar arithmetic -- and a run costs a lane one address add instead of an integer division by the run
// count and 64-bit pointer arithmetic (the element-per-thread form above: ~40 VALU instructions per run, ~3800 wave
// instructions per 76x76x9 crop against the 2700 of the multiply loop that follows).  UR rows per wave and call.
template <int UR>
struct Conv1RowRegs {
  uint32_t dw[UR][5];
  uint32_t sh[UR];  // (wave-uniform)
  int dst[UR];      // (wave-uniform row offset; < 0: no row)
};

template <int UR>
__device__ __forceinline__ void conv1_stage_rows_issue(Conv1RowRegs<UR>& rg, const uint8_t* crop, int pitch, int nbytes,
                                                       int rows, int RSb, int r_first, int wave, int nwaves, int lane) {
  const bool lane_on = 16 * lane < nbytes;
#pragma unroll
  for (int k = 0; k < UR; ++k) {
    const int r = r_first + wave + k * nwaves;  // (uniform)
    rg.dst[k] = r < rows ? r * RSb : -1;
    const uint8_t* p = crop + (size_t)min(r, rows - 1) * pitch;
    rg.sh[k] = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p - rg.sh[k]) + 4 * lane;
    // (registers of rows / lanes that load nothing stay undefined: the commit never stores them)
    if (r < rows && lane_on) {
#pragma unroll
      for (int e = 0; e < 4; ++e) rg.dw[k][e] = q[e];
      if (rg.sh[k]) rg.dw[k][4] = q[4];  // only touch the fifth dword when the run straddles it
    }
  }
}

template <int UR>
__device__ __forceinline__ void conv1_stage_rows_commit(const Conv1RowRegs<UR>& rg, uint8_t* lds, int nbytes, int lane) {
  const bool lane_on = 16 * lane < nbytes;
#pragma unroll
  for (int k = 0; k < UR; ++k) {
    uint4 o;
    o.x = __builtin_amdgcn_alignbyte(rg.dw[k][1], rg.dw[k][0], rg.sh[k]);
    o.y = __builtin_amdgcn_alignbyte(rg.dw[k][2], rg.dw[k][1], rg.sh[k]);
    o.z = __builtin_amdgcn_alignbyte(rg.dw[k][3], rg.dw[k][2], rg.sh[k]);
    o.w = __builtin_amdgcn_alignbyte(rg.dw[k][4], rg.dw[k][3], rg.sh[k]);
    if (rg.dst[k] >= 0 && lane_on) *reinterpret_cast<uint4*>(lds + rg.dst[k] + 16 * lane) = o;
  }
}

// all rows [r_first, rows) of a crop, UR per wave and pass
template <int UR>
__device__ __forceinline__ void conv1_stage_rows(uint8_t* lds, const uint8_t* crop, int pitch, int nbytes, int rows,
                                                 int RSb, int r_first, int wave, int nwaves, int lane) {
  for (int r0 = r_first; r0 < rows; r0 += UR * nwaves) {
    Conv1RowRegs<UR> rg;
    conv1_stage_rows_issue<UR>(rg, crop, pitch, nbytes, rows, RSb, r0, wave, nwaves, lane);
    conv1_stage_rows_commit<UR>(rg, lds, nbytes, lane);
  }
}

// float4 copy of `n4` contiguous float4 from HBM into the pixel-padded LDS band layout (8 float4 per pixel ->
// stride kLdsPix floats), U loads in flight per thread per pass
__device__ __forceinline__ void stage_band_f32(float* lds_band, const float* src, int n4, int tid, int nthreads) {
  constexpr int U = 6;
  for (int f0 = 0; f0 < n4; f0 += nthreads * U) {
    f32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int f = f0 + tid + k * nthreads;
      v[k] = f < n4 ? *reinterpret_cast<const f32x4*>(src + (size_t)f * 4) : f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int f = f0 + tid + k * nthreads;
      if (f < n4) *reinterpret_cast<f32x4*>(lds_band + (f >> 3) * kLdsPix + (f & 7) * 4) = v[k];
    }
  }
}

template <int C>
__global__ __launch_bounds__(512) void conv1_fwd_u8_kernel(Conv1Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KR = (3 * C + 3) & ~3;
  constexpr int KQ = KR / 4;      // k-steps per tap row
  constexpr int NS = 3 * KQ;      // k-steps
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int RSb = conv1_row_bytes(a.Wc, C);

  // (two problems in one launch: the workgroups of the second minibatch follow the first's and take its weights)
  const int nitems1 = a.B * a.nbands;
  const bool second = (int)blockIdx.x >= nitems1;
  const int item = second ? blockIdx.x - nitems1 : blockIdx.x;
  const int Bc = second ? a.B2 : a.B;
  if (second) a.idx = a.idx2, a.h1 = a.h1_2, a.w1 = a.w1_2, a.w = a.w2, a.bias = a.bias2, a.out = a.out2;
  const int band = item / Bc, b = item - band * Bc;
  const int y0 = band * a.th;
  const int tha = min(a.th, a.Ho - y0);
  // the first staging pass (7 x 512 runs: a whole 76x76x9 crop) is issued before the weight phase, whose two
  // barriers and LDS gather then run under the loads' latency; its LDS stores come after (same LDS region)
  constexpr int U0 = 7;
  Conv1StageRegs<U0> rg0;
  const uint8_t* frame0;
  int oh0, ow0;
  {
    const int64_t fi = a.idx ? a.idx[b] : b;
    oh0 = a.h1 ? a.h1[b] : 0, ow0 = a.w1 ? a.w1[b] : 0;
    frame0 = static_cast<const uint8_t*>(a.src) + (size_t)fi * a.Hs * a.Ws * C;
    if (!ABL(1)) conv1_stage_u8_issue<U0>(rg0, frame0, oh0, ow0, C, a.Ws, a.Wc, 2 * y0, 2 * tha + 1, RSb, 0, tid, 512);
  }

  // weights -> MFMA A-operand registers through a k-major LDS image [dy][rr (KR, zero padded)][cout 32], rr = dx*C + c,
  // with the 1/255 of `obs / 255.` (encoder.py:78) folded in: the index arithmetic is paid once per weight while
  // staging (5 per thread), and every lane then reads its 2 x NS values at compile-time offsets from ONE base
  // (the per-register gather out of the OIHW image cost ~10 VALU for each of the 42 registers of every lane).
  for (int i = tid; i < 3 * KR * 32; i += 512) {
    const int co = i & 31, k = i >> 5;
    const int dy = k / KR, rr = k - dy * KR;
    const int dx = rr / C, c = rr - dx * C;
    lds[i] = rr < 3 * C ? a.w[(co * C + c) * 9 + dy * 3 + dx] * a.scale : 0.f;
  }
  __syncthreads();
  float wr[NS][2];
  {
    const float* wl = lds + kq * 32 + li;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) wr[s][mt] = wl[((s / KQ) * KR + 4 * (s % KQ)) * 32 + mt * 16];
  }
  f32x4 bias4[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) bias4[mt] = *reinterpret_cast<const f32x4*>(a.bias + mt * 16 + 4 * kq);
  __syncthreads();

  uint8_t* ldsb = reinterpret_cast<uint8_t*>(lds);
  if (!ABL(1)) {
    conv1_stage_u8_commit<U0>(rg0, ldsb);
    conv1_stage_u8(ldsb, static_cast<const uint8_t*>(a.src), a.idx, a.h1, a.w1, b, C, a.Hs, a.Ws, a.Wc, 2 * y0,
                   2 * tha + 1, RSb, tid, 512, /*first_run=*/U0 * 512);  // taller bands: the rest
  }
  __syncthreads();

  const int npix = tha * a.Wo;
  const int ntiles = ABL(64) ? 0 : (npix + 15) >> 4;
  int ty = (wave * 16 + li) / a.Wo, x = (wave * 16 + li) - ty * a.Wo;
  const int qstep = 128 / a.Wo, rstep = 128 - qstep * a.Wo;
  for (int t = wave; t < ntiles; t += 8) {
    const bool pv = t * 16 + li < npix;
    if (!pv) ty = 0, x = 0;
    const uint8_t* base = ldsb + 2 * ty * RSb + 2 * x * C + kq;
    // all NS byte reads of the tile are issued first; each conversion is then placed one k-step ahead of the MFMA
    // pair that consumes it (left alone, the compiler emits read -> wait -> convert -> s_nop -> 2 MFMAs chains that
    // expose the LDS latency and a VALU->MFMA hazard stall on every k-step: tools/micro/conv1_loop.hip)
    uint32_t raw[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) raw[s] = base[(s / KQ) * RSb + 4 * (s % KQ)];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[2] = {bias4[0], bias4[1]};  // bias through the accumulators' initial values
    // (the conversions are volatile asm so that instruction selection cannot sink them next to their users; the
    // first one carries the 2 wait states a VALU write needs before an MFMA reads it, every other one has the two
    // MFMAs of the previous k-step between itself and its reader: tools/check_asm_hazards.py scans the ISA)
    float cur;
    asm volatile("v_cvt_f32_ubyte0 %0, %1\n\ts_nop 1" : "=v"(cur) : "v"(raw[0]));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float nxt = cur;
      if (s + 1 < NS) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(nxt) : "v"(raw[s + 1]));
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mfma16(wr[s][0], cur, acc[0]);
      acc[1] = mfma16(wr[s][1], cur, acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    if (pv && !ABL(4)) {
      const size_t g = ((size_t)(b * a.Ho + y0 + ty) * a.Wo + x) * 32 + 4 * kq;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 v = acc[mt];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.out + g + mt * 16));
      }
    }
    x += rstep, ty += qstep;  // 8 waves x 16 pixels further: qstep rows + rstep columns, at most one more wrap
    const bool wrap = x >= a.Wo;
    x = wrap ? x - a.Wo : x;
    ty = wrap ? ty + 1 : ty;
  }
}

// ---------------------------------------------------------------------------
// The same layer, HYBRID form: the banded kernel's input side (one workgroup per sample, the crop staged as bytes in
// LDS by one burst of aligned loads -- which is what makes that kernel indifferent to where the ring slots come from)
// with the row walk's compute loop (conv1_u8_rw.h: a wave owns 16 output columns and walks down; a lane group's
// E = ceil(3C/4) operand bytes of an input row are CONTIGUOUS, here read from LDS as aligned dwords + v_alignbyte, two
// new rows per 6 E MFMAs) instead of one byte read + one (row, column) walk per k-step.  The sample's steps (strips x
// rows, rw::Geom) are split evenly over the 8 waves.  Only for crops that fit one band (nbands == 1).
// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(512, 2) void conv1_u8_walk_kernel(Conv1Args a, rw::Geom G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int E = (3 * C + 3) / 4, KR = 4 * E;
  constexpr int NLD = (E + 3 + 3) / 4;  // aligned dwords that hold a run starting at byte 0..3 of the first
  constexpr int NWD = (E + 3) / 4;      // dwords of the run once it starts at byte 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int RSb = conv1_row_bytes(a.Wc, C);
  const bool second = (int)blockIdx.x >= a.B;
  const int b = second ? blockIdx.x - a.B : blockIdx.x;
  if (second) a.idx = a.idx2, a.h1 = a.h1_2, a.w1 = a.w1_2, a.w = a.w2, a.bias = a.bias2, a.out = a.out2;
  // the crop's bytes: requested before the weight phase, stored to LDS after it (same region)
  constexpr int U0 = 10;  // rows per wave in flight across the weight phase (8 waves x 10 >= the 77 rows of a 76x76 crop)
  Conv1RowRegs<U0> rg0;
  const int crop_rows = 2 * a.Ho + 1, crop_bytes = a.Wc * C;
  const uint8_t* crop0;
  {
    const int64_t fi = a.idx ? rw::const_load(a.idx, b) : (int64_t)b;  // (scalar loads: b is wave-uniform)
    const int oh0 = a.h1 ? rw::const_load(a.h1, b) : 0, ow0 = a.w1 ? rw::const_load(a.w1, b) : 0;
    crop0 = static_cast<const uint8_t*>(a.src) + ((size_t)fi * a.Hs + oh0) * a.Ws * C + (size_t)ow0 * C;
  }
  conv1_stage_rows_issue<U0>(rg0, crop0, a.Ws * C, crop_bytes, crop_rows, RSb, 0, wave, 8, lane);  // (host: <= 64 runs per row)
  for (int i = tid; i < 3 * KR * 32; i += 512) {
    const int co = i & 31, k = i >> 5;
    const int dy = k / KR, rr = k - dy * KR;
    const int dx = rr / C, c = rr - dx * C;
    lds[i] = rr < 3 * C ? a.w[(co * C + c) * 9 + dy * 3 + dx] * a.scale : 0.f;
  }
  __syncthreads();
  float wr[3][E][2];  // lane (li = cout, kq): W[cout][dy][rr = E kq + e] * scale
  {
    const float* wl = lds + (E * kq) * 32 + li;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) wr[dy][e][mt] = wl[(dy * KR + e) * 32 + mt * 16];
  }
  f32x4 bias4[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) bias4[mt] = *reinterpret_cast<const f32x4*>(a.bias + mt * 16 + 4 * kq);
  __syncthreads();
  uint8_t* ldsb = reinterpret_cast<uint8_t*>(lds);
  conv1_stage_rows_commit<U0>(rg0, ldsb, crop_bytes, lane);
  conv1_stage_rows<4>(ldsb, crop0, a.Ws * C, crop_bytes, crop_rows, RSb, /*r_first=*/U0 * 8, wave, 8, lane);
  __syncthreads();

  const int lo = G.steps * wave / 8, hi = G.steps * (wave + 1) / 8;
  const int out_row = a.Wo * 128;
  const __amdgpu_buffer_rsrc_t rout = rw::uniform_rsrc(a.out + (size_t)b * a.Ho * a.Wo * 32, a.Ho * out_row);
  for (int g = lo; g < hi;) {
    int k, sb, n_strip;
    if (g < G.nfull * G.Ho) {
      k = g / G.Ho, sb = g - k * G.Ho, n_strip = G.Ho;
    } else {
      const int q = (g - G.nfull * G.Ho) / G.nr;
      k = G.nfull + q, sb = g - G.nfull * G.Ho - q * G.nr, n_strip = G.nr;
    }
    const int n = hi - g < n_strip - sb ? hi - g : n_strip - sb;
    g += n;
    int x, y0;
    bool lane_on;
    if (k < G.nfull) {
      x = 16 * k + li, y0 = 0, lane_on = true;
    } else {
      const int u = (k - G.nfull) * 16 + li;
      const int col = u / G.nseg, sg = u - col * G.nseg;
      lane_on = col < G.brem;
      x = 16 * G.nfull + col, y0 = sg * G.nr;
    }
    // (lanes without a column, and rows past the crop, read whatever sits in LDS: finite bytes; nothing of it is stored)
    const int Y = lane_on ? min(y0 + sb, a.Ho - 1) : 0;
    const int xx = lane_on ? x : 0;
    const unsigned run = (unsigned)(2 * Y * RSb + 2 * xx * C + E * kq);
    const unsigned sh = run & 3u;
    const uint8_t* rowp = ldsb + (run & ~3u);
    unsigned vo = lane_on ? (unsigned)(((y0 + sb) * a.Wo + x) * 128 + kq * 16) : 0x80000000u;
    const int rmax = 2 * a.Ho - 2 * Y;  // last crop row (relative to 2 Y) that exists in LDS
    struct Raw {
      uint32_t d[NLD];
    };
    struct Row {
      float v[E];
    };
    auto load_row = [&](Raw& R, int r) {  // crop row 2 Y + r (clamped into the staged image)
      const uint32_t* p = reinterpret_cast<const uint32_t*>(rowp + __mul24(min(r, rmax), RSb));  // (24-bit: no 64-bit mad)
#pragma unroll
      for (int j = 0; j < NLD; ++j) R.d[j] = p[j];
    };
    auto convert = [&](Row& F, const Raw& R) {
      rw::RawBytes<NWD> Wd;
#pragma unroll
      for (int j = 0; j < NWD; ++j) Wd.d[j] = __builtin_amdgcn_alignbyte(j + 1 < NLD ? R.d[j + 1 < NLD ? j + 1 : j] : 0u, R.d[j], sh);
#pragma unroll
      for (int e = 0; e < E; ++e) F.v[e] = rw::byte_f32<NWD>(Wd, e);
    };
    auto mma_row = [&](f32x4 (&acc)[2], const Row& F, const int dy) {
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma16(wr[dy][e][mt], F.v[e], acc[mt]);
    };
    struct Pair {
      Raw a, b;
    };
    auto step = [&](const Row& r0, Row& r1, Row& r2, const Pair& cur, Pair& nxt, const int t) {
      load_row(nxt.a, 2 * t + 3);
      load_row(nxt.b, 2 * t + 4);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[2] = {bias4[0], bias4[1]};
      mma_row(acc, r0, 0);
      __builtin_amdgcn_sched_barrier(0);
      convert(r1, cur.a);
      convert(r2, cur.b);
      __builtin_amdgcn_sched_barrier(0);
      mma_row(acc, r1, 1);
      mma_row(acc, r2, 2);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 v = acc[mt];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rw::relu_bits(v[r]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rout,
                                               vo + mt * 64u, 0, 2);
      }
      vo += out_row;
    };
    Row S0, S1, S2, S3, S4;
    Pair P0, P1;
    {
      Raw R0;
      load_row(R0, 0), load_row(P0.a, 1), load_row(P0.b, 2);
      convert(S0, R0);
    }
    for (int t = 0;;) {  // rows of step t sit in sets (2t, 2t+1, 2t+2) mod 5, its bytes in pair t mod 2
      step(S0, S1, S2, P0, P1, t);
      if (++t >= n) break;
      step(S2, S3, S4, P1, P0, t);
      if (++t >= n) break;
      step(S4, S0, S1, P0, P1, t);
      if (++t >= n) break;
      step(S1, S2, S3, P1, P0, t);
      if (++t >= n) break;
      step(S3, S4, S0, P0, P1, t);
      if (++t >= n) break;
      step(S0, S1, S2, P1, P0, t);
      if (++t >= n) break;
      step(S2, S3, S4, P0, P1, t);
      if (++t >= n) break;
      step(S4, S0, S1, P1, P0, t);
      if (++t >= n) break;
      step(S1, S2, S3, P0, P1, t);
      if (++t >= n) break;
      step(S3, S4, S0, P1, P0, t);
      if (++t >= n) break;
    }
  }
}

// ---------------------------------------------------------------------------
// weight gradient, stride-1 32->32:  dW[co][ci][tap] = sum_pixels g[p][co] * in[p+tap][ci]
// GEMM view: D[co][ci] per tap, K = pixels (4 per MFMA).  36 accumulator tiles
// (2 x 2 x 9) live in registers for the whole persistent workgroup; partial
// sums leave through one slab per workgroup and a deterministic second pass.
// ---------------------------------------------------------------------------
struct WgradS1Args {
  const float* in;  // [B][Hi][Wi][32]
  const float* g;   // [B][Ho][Wo][32]
  float* partial;   // [grid][kPartial]
  int B, Hi, Wi, Ho, Wo, th, nbands;
};
constexpr int kPartialS1 = 32 * 288 + 32;

// WALK: how a lane finds the pixel pair of its k-step.  0: per-lane (row, column) counters, any size.  1 / 2 (even / odd
// Wo, rows of at least 8 pairs): the k-step's first pair is walked in SCALAR registers and a lane is a constant offset
// from it (+ one conditional row-wrap correction); the gradient comes through buffer loads whose descriptor ends at the
// band's end, so pairs past the band read zeros.  The per-lane walk cost ~30 VALU instructions per 24 MFMAs -- and
// a VALU instruction holds the SIMD's issue port for its 4 cycles while the matrix pipe waits.
template <int WALK>
__device__ __forceinline__ void wgrad_s1_body(const WgradS1Args& a, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = WALK ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  // a wave owns the 16 output channels mt*16.. of dW; wave pairs share pixels.
  // Winograd F(3,2) along x (the transpose of the forward's F(2,3)): for a horizontal pair of
  // gradient pixels (g0,g1) and its 4-pixel input window (d0..d3), the three x taps need 4 products
  //   M0 += g0 (d0-d2), M1 += (g0+g1)(d1+d2), M2 += (g0-g1)(d2-d1), M3 += -g1 (d1-d3)
  // summed over all pairs; dW(dx=0,1,2) = M0+(M1+M2)/2, (M1-M2)/2, (M1+M2)/2+M3 once at the end.
  const int mt = wave & 1, uslot = wave >> 1;
  f32x4 acc[3][4][2];  // [dy][k][cin tile]
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[dy][k][ct] = f32x4{0, 0, 0, 0};
  float bsum = 0.f;
  f32x2 wt = {0, 0};  // winograd_bt_pk's temporary, live for the whole kernel (see common.h)

  const int PW = (a.Wo + 1) >> 1;  // pixel pairs per gradient row
  const int qstep = 8 / PW, rstep = 8 - qstep * PW;
  const int nitems = a.B * a.nbands;
  for (int item = bid; item < nitems; item += nblk) {
    const int band = item / a.B, b = item - band * a.B;  // band-major: every workgroup sees every band size
    const int y0 = band * a.th;
    const int tha = min(a.th, a.Ho - y0);
    // band = (tha+2) input rows, contiguous in HBM.  The gradient rows are NOT staged: a lane's gradient operand is
    // channel mt*16+li of the two pixels of its pair, every value is needed by exactly one wave, and the 16 lanes
    // of a group read 64 contiguous bytes -- each wave loads them from HBM/L2 itself, three k-steps ahead.  The
    // LDS then holds input rows only: bands of 13 rows instead of 6 at 37x37 (half the items, 12 % less halo).
    const int nin = (tha + 2) * a.Wi * 8;
    {
      const float* pin = a.in + ((size_t)(b * a.Hi + y0) * a.Wi) * 32;
      f32x4 pf[kMaxPf];
#pragma unroll
      for (int u = 0; u < kMaxPf; ++u) {
        const int f = tid + u * 256;
        f32x4 v = {0, 0, 0, 0};
        if (f < nin) v = *reinterpret_cast<const f32x4*>(pin + (size_t)f * 4);
        pf[u] = v;
      }
#pragma unroll
      for (int u = 0; u < kMaxPf; ++u) {
        const int f = tid + u * 256;
        if (f < nin) *reinterpret_cast<f32x4*>(lds + (f >> 3) * kLdsPix + (f & 7) * 4) = pf[u];
      }
      // the pixel behind the band: with an odd row length the last pair's window reaches one pixel past the last
      // row; its product has a zero gradient factor, but 0 x (whatever LDS held) must not be NaN
      if (tid < kLdsPix / 4) *reinterpret_cast<f32x4*>(lds + (tha + 2) * a.Wi * kLdsPix + tid * 4) = f32x4{0, 0, 0, 0};
    }
    __syncthreads();

    const int npairs = tha * PW;
    const float* const gband = a.g + ((size_t)(b * a.Ho + y0) * a.Wo) * 32 + mt * 16 + li;
    const int nunits = ((npairs + 7) >> 3) << 1;  // 4 pairs per MFMA k-step, 2 k-steps per group of 8 pairs
    // This wave's k-steps are units u = uslot, uslot+2, ...; the lane's pair of unit u is q(u) = (u>>1)*8 + (u&1) + 2*kq
    // (pairs of a k-step are 2 apart = 4 pixels: the two lane groups of an LDS half hit disjoint banks).  Both
    // fetchers are called for consecutive k-steps in order and keep their pair's (row, column) incrementally
    // (+8 pairs per call), each with its own copy because the gradient fetch runs ahead of the window fetch.
    int gy = ((uslot & 1) + 2 * kq) / PW, gj = ((uslot & 1) + 2 * kq) - gy * PW;
    int fy = gy, fj = gj;
    auto advance = [&](int& y, int& j) {  // 8 pairs further = qstep rows + rstep pairs, at most one more wrap
      j += rstep, y += qstep;
      const bool wrap = j >= PW;
      j = wrap ? j - PW : j;
      y = wrap ? y + 1 : y;
    };
    auto gfetch = [&](int u, float (&gv)[2]) {
      const int q = (u >> 1) * 8 + (u & 1) + 2 * kq;
      const bool pv = (u < nunits) && (q < npairs);
      const int x0 = 2 * gj;
      const float* gp = gband + (gy * a.Wo + x0) * 32;
      const bool second = pv && x0 + 1 < a.Wo;
      gv[0] = *(pv ? gp : g_zero_px + li);  // (lanes past the band load a zero: no select on the loaded value)
      gv[1] = *(second ? gp + 32 : g_zero_px + li);
      advance(gy, gj);
    };
    auto dfetch = [&](int u, f32x2 (&dv)[3][4]) {
      const int q = (u >> 1) * 8 + (u & 1) + 2 * kq;
      const bool pv = (u < nunits) && (q < npairs);
      const int ty = pv ? fy : 0, x0 = pv ? 2 * fj : 0;
      const float* ip = lds + (ty * a.Wi + x0) * kLdsPix + li;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float* qd = ip + (dy * a.Wi + c) * kLdsPix;
          dv[dy][c] = f32x2{qd[0], qd[16]};  // the two cin tiles of one window pixel (one ds_read2_b32)
        }
      advance(fy, fj);
    };
    auto mma = [&](const float (&gv)[2], f32x2 (&dv)[3][4]) {
      bsum += gv[0] + gv[1];
      const float g0 = gv[0], g1 = gv[0] + gv[1], g2 = gv[0] - gv[1], g3 = -gv[1];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        // (d0-d2, d1+d2, d2-d1, d1-d3) for both cin tiles with 4 packed adds
        winograd_bt_pk(dv[dy][0], dv[dy][1], dv[dy][2], dv[dy][3], wt);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          acc[dy][0][ct] = mfma16(g0, dv[dy][0][ct], acc[dy][0][ct]);
          acc[dy][1][ct] = mfma16(g1, wt[ct], acc[dy][1][ct]);
          acc[dy][2][ct] = mfma16(g2, dv[dy][2][ct], acc[dy][2][ct]);
          acc[dy][3][ct] = mfma16(g3, dv[dy][3][ct], acc[dy][3][ct]);
        }
      }
    };
    // ---- WALK 1 / 2: the same two fetchers on a scalar walk ----
    // k-step of unit u: first pair q0 = (u >> 1) * 8 + uslot at (row sy, pair sj), scalar; lane kq's pair is 2 kq
    // further, wrapped into the next row when sj + 2 kq >= PW (PW >= 8 > 6: at most once).
    //   gradient floats from the band start: (sy Wo + 2 sj) 32  [scalar]  +  4 kq 32 + mt 16 + li  [lane]
    //                                        + (Wo - 2 PW) 32 if wrapped (0 / -32 for even / odd Wo)
    //   window floats in LDS:                (sy Wi + 2 sj) 36  [scalar]  +  4 kq 36 + li          [lane]
    //                                        + (Wi - 2 PW) 36 if wrapped
    // A pair past the band end reads gradient zeros (buffer range) and, in LDS, the last valid pair's window (its
    // address is clamped: 0 x finite).  Odd Wo: the second pixel of a row's last pair is the next row's first in
    // memory -- those lanes load from past the buffer instead.
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.g + ((size_t)(b * a.Ho + y0) * a.Wo) * 32), (short)0, WALK ? tha * a.Wo * 128 : 0, 0x00020000);
    const unsigned g_lane = (unsigned)(4 * kq * 32 + mt * 16 + li) * 4u;
    const unsigned g_lane_w = g_lane + (unsigned)((a.Wo - 2 * PW) * 128);
    const unsigned d_lane = (unsigned)(4 * kq * kLdsPix + li) * 4u;
    const unsigned d_lane_w = d_lane + (unsigned)((a.Wi - 2 * PW) * kLdsPix * 4);
    const unsigned d_max = (unsigned)(((tha - 1) * a.Wi + 2 * (PW - 1)) * kLdsPix + 15) * 4u;
    const unsigned d_row = (unsigned)(a.Wi * kLdsPix * 4);
    int sgy = 0, sgj = uslot, sfy = 0, sfj = uslot;  // (uslot < 2 <= PW)
    auto sadvance = [&](int& y, int& j) {
      j += rstep, y += qstep;
      if (j >= PW) j -= PW, y += 1;
    };
    auto gfetch_s = [&](float (&gv)[2]) {
      const unsigned soff = (unsigned)((sgy * a.Wo + 2 * sgj) * 128);
      const int jl = sgj + 2 * kq;
      const bool wrapped = jl >= PW;
      if (WALK == 1) {  // even Wo: rows of pairs are contiguous, nothing depends on the wrap
        gv[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, g_lane, soff, 0));
        gv[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, g_lane, soff + 128u, 0));
      } else {
        const unsigned v0 = wrapped ? g_lane_w : g_lane;
        const bool last = (jl == PW - 1) || (jl == 2 * PW - 1);
        const unsigned v1 = last ? 0x80000000u : v0;
        gv[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, v0, soff, 0));
        gv[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, v1, soff + 128u, 0));
      }
      sadvance(sgy, sgj);
    };
    auto dfetch_s = [&](f32x2 (&dv)[3][4]) {
      const unsigned sbase = (unsigned)((sfy * a.Wi + 2 * sfj) * kLdsPix * 4);
      const bool wrapped = sfj + 2 * kq >= PW;
      const unsigned off = min((wrapped ? d_lane_w : d_lane) + sbase, d_max);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const float* row = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + off + dy * d_row);
#pragma unroll
        for (int c = 0; c < 4; ++c) dv[dy][c] = f32x2{row[c * kLdsPix], row[c * kLdsPix + 16]};
      }
      sadvance(sfy, sfj);
    };
    // software pipeline over this wave's k-steps: window reads (LDS) one step ahead in two register sets, gradient
    // values (HBM/L2) three steps ahead in four; a k-step past the band multiplies zeros and is skipped
    float g0[2], g1[2], g2[2], g3[2];
    f32x2 dA[3][4], dB[3][4];
    if (WALK) {
      gfetch_s(g0), gfetch_s(g1), gfetch_s(g2);
      dfetch_s(dA);
      for (int u = uslot; u < nunits; u += 8) {
        gfetch_s(g3);
        dfetch_s(dB);
        __builtin_amdgcn_sched_barrier(0);
        mma(g0, dA);
        __builtin_amdgcn_sched_barrier(0);
        gfetch_s(g0);
        dfetch_s(dA);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 2 < nunits) mma(g1, dB);
        __builtin_amdgcn_sched_barrier(0);
        gfetch_s(g1);
        dfetch_s(dB);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 4 < nunits) mma(g2, dA);
        __builtin_amdgcn_sched_barrier(0);
        gfetch_s(g2);
        dfetch_s(dA);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 6 < nunits) mma(g3, dB);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      gfetch(uslot, g0), gfetch(uslot + 2, g1), gfetch(uslot + 4, g2);
      dfetch(uslot, dA);
      for (int u = uslot; u < nunits; u += 8) {
        gfetch(u + 6, g3);
        dfetch(u + 2, dB);
        __builtin_amdgcn_sched_barrier(0);
        mma(g0, dA);
        __builtin_amdgcn_sched_barrier(0);
        gfetch(u + 8, g0);
        dfetch(u + 4, dA);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 2 < nunits) mma(g1, dB);
        __builtin_amdgcn_sched_barrier(0);
        gfetch(u + 10, g1);
        dfetch(u + 6, dB);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 4 < nunits) mma(g2, dA);
        __builtin_amdgcn_sched_barrier(0);
        gfetch(u + 12, g2);
        dfetch(u + 8, dA);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 6 < nunits) mma(g3, dB);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  // output transform (linear, so applied once to the accumulated products), then the cross-wave
  // sum in a fixed order (deterministic) and one slab per workgroup
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  for (int w = 0; w < 2; ++w) {
    if (uslot == w) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 hs = 0.5f * (acc[dy][1][ct] + acc[dy][2][ct]);
          const f32x4 hd = 0.5f * (acc[dy][1][ct] - acc[dy][2][ct]);
          const f32x4 dw[3] = {acc[dy][0][ct] + hs, hd, hs + acc[dy][3][ct]};
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = mt * 16 + 4 * kq + r, ci = ct * 16 + li;
              float* d = lds + co * 288 + ci * 9 + dy * 3 + dx;
              *d = (w == 0) ? dw[dx][r] : *d + dw[dx][r];
            }
        }
      if (kq == 0) {
        float* d = lds + 32 * 288 + mt * 16 + li;
        *d = (w == 0) ? bsum : *d + bsum;
      }
    }
    __syncthreads();
  }
  float* slab = a.partial + (size_t)bid * kPartialS1;
  for (int i = tid; i < kPartialS1; i += 256) slab[i] = lds[i];
}

template <int WALK>
__global__ __launch_bounds__(256, 2) void wgrad_s1_kernel(WgradS1Args a) {
  wgrad_s1_body<WALK>(a, blockIdx.x, gridDim.x);
}

// Weight gradient and data gradient of one layer in ONE launch: both only read the layer's output gradient, so they
// need not wait for each other.  The first nw workgroups run the weight-gradient body, the rest the data-gradient
// body; the hardware starts the second set as the first one's workgroups retire, so the tail of one kernel and the
// ramp of the next overlap instead of being separated by a kernel boundary (4-9 us per launch at these sizes).
template <int WALK>
__global__ __launch_bounds__(256, 2) void bwd_s1_kernel(WgradS1Args wa, ConvS1Args da, int nw) {
  if ((int)blockIdx.x < nw)
    wgrad_s1_body<WALK>(wa, blockIdx.x, nw);
  else
    conv_s1_body<MODE_DGRAD>(da, (int)blockIdx.x - nw, (int)gridDim.x - nw);
}

// the same launch with the data gradient in its row-walk form (conv_rw.h)
template <int WALK>
__global__ __launch_bounds__(256, 2) void bwd_rw_kernel(WgradS1Args wa, rw::Args da, int nw) {
  if ((int)blockIdx.x < nw) {
    wgrad_s1_body<WALK>(wa, blockIdx.x, nw);
  } else {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    rw::build_filter<MODE_DGRAD, 256>(lds, da.p[0][0].w, nullptr, threadIdx.x);
    __syncthreads();
    rw::run_layer<MODE_DGRAD, 4>(da.g[0], da.p[0][0], da.p[0][1], lds, (int)blockIdx.x - nw, (int)gridDim.x - nw);
  }
}

#include "conv_rw_wgrad.h"

// weight gradient in its row-walk form (conv_rw_wgrad.h), alone and in one launch with the row-walk data gradient
__global__ __launch_bounds__(256, 2) void wgrad_rw_kernel(rw::WgradArgs wa) {
  rw::wgrad_body<4>(wa, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256, 2) void bwd_rw2_kernel(rw::WgradArgs wa, rw::Args da, int nw) {
  if ((int)blockIdx.x < nw) {
    rw::wgrad_body<4>(wa, blockIdx.x, nw);
  } else {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    rw::build_filter<MODE_DGRAD, 256>(lds, da.p[0][0].w, nullptr, threadIdx.x);
    __syncthreads();
    rw::run_layer<MODE_DGRAD, 4>(da.g[0], da.p[0][0], da.p[0][1], lds, (int)blockIdx.x - nw, (int)gridDim.x - nw);
  }
}

// ---------------------------------------------------------------------------
// weight gradient of the first layer (stride 2, Cin = C, input re-read from
// the uint8 frames / float tensor exactly as the forward does).
// D[co][k'] with k' = dy*KR + dx*C + c (the forward's K index), K = pixels.
// ---------------------------------------------------------------------------
struct Wgrad1Args {
  const void* src;
  const int64_t* idx;
  const int32_t* h1;
  const int32_t* w1;
  const float* g;  // [B][Ho][Wo][32]
  float* partial;  // [grid][32*C*9 + 32]
  int B, C, Hs, Ws, Hc, Wc, Ho, Wo, th, nbands;
  float scale;
  unsigned lds_bytes;  // dynamic LDS of the launch (the uint8 kernel sizes its final cross-wave sum by it)
  int dbg;
};

template <int SRC, int C>
__global__ __launch_bounds__(512) void wgrad1_kernel(Wgrad1Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KR = (3 * C + 3) & ~3;
  constexpr int NT = (3 * KR + 15) / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int RS = conv1_row_stride(a.Wc, C);
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0, 0, 0, 0};
  float bsum[2] = {0.f, 0.f};
  int koff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int k = t * 16 + li;
    const int dy = k / KR, rr = k - dy * KR;
    koff[t] = (dy < 3) ? dy * RS + rr : 0;
  }

  const int nitems = a.B * a.nbands;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int band = item / a.B, b = item - band * a.B;  // band-major: every workgroup sees every band size
    const int y0 = band * a.th;
    const int tha = min(a.th, a.Ho - y0);
    conv1_stage<SRC>(lds, a.src, a.idx, a.h1, a.w1, b, C, a.Hs, a.Ws, a.Hc, a.Wc, 2 * y0, 2 * tha + 1, RS, a.scale,
                     tid, 512);
    __syncthreads();
    // (the gradient operand is loaded from HBM/L2 straight into MFMA registers, as in wgrad1_u8_kernel: every pixel
    // is needed by exactly one wave, and lanes with nothing to multiply read a zero page)
    const float* const gband = a.g + ((size_t)(b * a.Ho + y0) * a.Wo) * 32 + li;
    const int npix = tha * a.Wo;
    const int nunits = ((npix + 15) >> 4) << 2;
    // unit u's pixel of lane group kq is p = (u >> 2) * 16 + (u & 3) + 4 kq; this wave's units are u = wave, wave + 8,
    // ...: p advances by 32 per unit -- walked incrementally as (row, column), no division per unit
    int p = (wave >> 2) * 16 + (wave & 3) + 4 * kq;
    int ty = p / a.Wo, x = p - ty * a.Wo;
    const int qstep = 32 / a.Wo, rstep = 32 - qstep * a.Wo;
    for (int u = wave; u < nunits; u += 8) {
      const bool pv = p < npix;
      const float* gp = pv ? gband + p * 32 : g_zero_px;
      const float a0 = gp[0], a1 = gp[16];
      bsum[0] += a0;
      bsum[1] += a1;
      const float* ip = lds + (pv ? __mul24(2 * ty, RS) + __mul24(2 * x, C) : 0);
      float bv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t] = ip[koff[t]];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[0][t] = mfma16(a0, bv[t], acc[0][t]);
        acc[1][t] = mfma16(a1, bv[t], acc[1][t]);
      }
      p += 32, x += rstep, ty += qstep;  // at most one more wrap
      const bool wrap = x >= a.Wo;
      x = wrap ? x - a.Wo : x;
      ty = wrap ? ty + 1 : ty;
    }
    __syncthreads();
  }

  bsum[0] += __shfl_xor(bsum[0], 16);
  bsum[0] += __shfl_xor(bsum[0], 32);
  bsum[1] += __shfl_xor(bsum[1], 16);
  bsum[1] += __shfl_xor(bsum[1], 32);
  const int nw = 32 * C * 9;
  for (int w = 0; w < 8; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int k = t * 16 + li;
          const int dy = k / KR, rr = k - dy * KR;
          if (dy < 3 && rr < 3 * C) {
            const int dx = rr / C, c = rr - dx * C;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = mt * 16 + 4 * kq + r;
              float* d = lds + (co * C + c) * 9 + dy * 3 + dx;
              *d = (w == 0) ? acc[mt][t][r] : *d + acc[mt][t][r];
            }
          }
        }
      if (kq == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          float* d = lds + nw + mt * 16 + li;
          *d = (w == 0) ? bsum[mt] : *d + bsum[mt];
        }
      }
    }
    __syncthreads();
  }
  float* slab = a.partial + (size_t)blockIdx.x * (nw + 32);
  for (int i = tid; i < nw + 32; i += 512) slab[i] = lds[i];
}

// first-layer weight gradient from the uint8 ring with the input band kept as bytes in LDS (see
// conv1_fwd_u8_kernel).  The gradient operand never enters LDS: lane (li, kq) of a k-step needs channels li and
// 16 + li of ONE pixel, every pixel is needed by exactly one wave, and the 16 lanes of a group read 64 contiguous
// bytes -- so each wave loads its own operand values from HBM/L2 one k-step ahead (the other three waves of the SIMD
// cover the latency).  That removes three quarters of the staging volume (53 KB of gradients per 16 KB of bytes at
// 84x84x9) and lets a workgroup take a whole crop as bytes.
// NW waves per workgroup: 8 (two workgroups per CU) or 4 (four smaller ones: the stage -> barrier -> multiply ->
// barrier phases of a workgroup do not overlap each other, so what covers a workgroup's staging is the number of
// OTHER workgroups on its CU that are multiplying at that moment)
template <int C, int NW>
__global__ __launch_bounds__(64 * NW) void wgrad1_u8_kernel(Wgrad1Args a) {
  constexpr int NTHR = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // k' = dy * 3C + dx * C + c, unpadded (9C values): NT tiles of 16.  At C = 9 that is 81 = 5 tiles + ONE column; a
  // sixth tile for it would be a sixth of all MFMAs, so that column (dy = 2, dx = 2, c = C-1) is accumulated by two
  // VALU FMAs per k-step instead (lane (li, kq) holds the gradient of channels li / 16+li at its pixel anyway).
  constexpr int K9 = 9 * C;
  constexpr bool TAIL = (K9 % 16 == 1);
  constexpr int NT = TAIL ? K9 / 16 : (K9 + 15) / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: the piece bookkeeping below stays scalar)
  const int li = lane & 15, kq = lane >> 4;
  const int RSb = conv1_row_bytes(a.Wc, C);
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0, 0, 0, 0};
  f32x2 bsum = {0.f, 0.f};   // (pairs: one packed add / fma for both halves of the output channels)
  f32x2 atail = {0.f, 0.f};
  int koff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int k = t * 16 + li;
    const int dy = k / (3 * C), rr = k - dy * (3 * C);
    koff[t] = (dy < 3) ? dy * RSb + rr : 0;
  }
  const int koff_tail = 2 * RSb + 3 * C - 1;
  const int nitems = a.B * a.nbands;
  uint8_t* ldsb = reinterpret_cast<uint8_t*>(lds);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int band = item / a.B, b = item - band * a.B;
    const int y0 = band * a.th;
    const int tha = min(a.th, a.Ho - y0);
    if (!ABL(1)) {
      const int64_t fi = a.idx ? rw::const_load(a.idx, b) : (int64_t)b;  // (scalar loads: b is wave-uniform)
      const int oh = a.h1 ? rw::const_load(a.h1, b) : 0, ow = a.w1 ? rw::const_load(a.w1, b) : 0;
      const uint8_t* crop = static_cast<const uint8_t*>(a.src) + ((size_t)fi * a.Hs + oh + 2 * y0) * a.Ws * C + (size_t)ow * C;
      if (a.Wc * C <= 64 * 16)  // a lane per 16-byte run of a row
        conv1_stage_rows<10>(ldsb, crop, a.Ws * C, a.Wc * C, 2 * tha + 1, RSb, 0, wave, NW, lane);
      else
        conv1_stage_u8(ldsb, static_cast<const uint8_t*>(a.src), a.idx, a.h1, a.w1, b, C, a.Hs, a.Ws, a.Wc, 2 * y0,
                       2 * tha + 1, RSb, tid, NTHR);
    }
    __syncthreads();
    const float* const gband = a.g + ((size_t)(b * a.Ho + y0) * a.Wo) * 32 + li;  // the band's pixels are contiguous
    // The walk.  A k-step ("unit") is 4 pixels, lane group kq takes one of them; units come in PIECES of UNR, and the
    // band's pieces are dealt to the waves round-robin (piece w, w + NW, ...).  Two kinds of piece:
    //   row piece:    UNR consecutive units of one output row (pixels 4u + kq): only the floor(Wo / 4) WHOLE units of
    //                 a row, so that no unit multiplies fewer than 4 pixels;
    //   column piece: the Wo % 4 columns a row's whole units leave over, walked DOWN -- a unit is rows 4u + kq of one
    //                 such column (at Wo = 37: 10 units for column 36 instead of a quarter-filled tenth unit in each
    //                 of the 37 rows: 343 units per crop, not 370).
    // In a row piece everything a unit needs sits at a COMPILE-TIME offset from per-piece bases: the byte operands at
    // (piece base + koff[t]) + 8 C jj, the two gradient values at piece base + 128 jj floats -- no per-unit address
    // arithmetic (the pixel-order walk spent ~45 VALU instructions per 12 MFMAs on it: matrix pipe busy 46 %).  Pieces
    // of 3 (not whole rows) so that the waves' shares differ by at most one piece: 115 pieces over 8 waves = 15 each
    // at most, 45 units, where whole rows gave 5 rows x 10 units.  The bytes are multiplied unscaled, `scale` is
    // applied once to the accumulated sums.
    constexpr int UNR = 3;
    const int fullu = a.Wo >> 2, remc = a.Wo & 3;
    const int cpr = (fullu + UNR - 1) / UNR;  // row pieces per row
    const int cpc = (((tha + 3) >> 2) + UNR - 1) / UNR;  // column pieces per left-over column
    const int nrow = tha * cpr, npieces = nrow + remc * cpc;
    const int kdiv = NW / max(cpr, 1), kmod = NW - kdiv * cpr;
    struct Piece {  // (wave-uniform)
      int q, ty, c;  // row piece: row ty, piece c of the row;  column piece (q >= nrow): left-over column ty, piece c
    };
    auto column_of = [&](Piece& p) {
      const int qq = p.q - nrow;
      p.ty = qq / cpc, p.c = qq - p.ty * cpc;
    };
    auto next_piece = [&](Piece& p) {
      p.q += NW;
      if (p.q < nrow) {
        p.c += kmod, p.ty += kdiv;
        if (p.c >= cpr) p.c -= cpr, ++p.ty;
      } else {
        column_of(p);
      }
    };
    // gradient values of a piece (from HBM/L2, one piece ahead of their use; pixels past the band: zero page)
    auto gload = [&](const Piece& p, f32x2 (&av)[UNR]) {
      if (p.q < nrow) {
        const float* gp = gband + (p.ty * a.Wo + 4 * UNR * p.c + kq) * 32;
        if (UNR * (p.c + 1) <= fullu) {
#pragma unroll
          for (int jj = 0; jj < UNR; ++jj) av[jj] = f32x2{gp[128 * jj], gp[128 * jj + 16]};
        } else {
#pragma unroll
          for (int jj = 0; jj < UNR; ++jj) {
            const float* g1 = UNR * p.c + jj < fullu ? gp + 128 * jj : g_zero_px;
            av[jj] = f32x2{g1[0], g1[16]};
          }
        }
      } else {
        const int x = 4 * fullu + p.ty;
#pragma unroll
        for (int jj = 0; jj < UNR; ++jj) {
          const int y = 4 * (UNR * p.c + jj) + kq;
          const float* g1 = (p.ty < remc && y < tha) ? gband + (y * a.Wo + x) * 32 : g_zero_px;
          av[jj] = f32x2{g1[0], g1[16]};
        }
      }
    };
    // operand bytes of a piece, LDS -> registers (also one piece ahead: the multiply below never waits for LDS).
    // Units past the row / the band get an address inside the band: their gradient is the zero page's.
    constexpr int NB = NT + (TAIL ? 1 : 0);
    struct Raw {
      uint32_t b[UNR][NB];
    };
    auto bload_unit = [&](const uint8_t* ub, uint32_t (&b)[NB]) {
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = ub[koff[t]];
      if (TAIL) b[NT] = ub[koff_tail];
    };
    auto bload = [&](const Piece& p, Raw& R) {
      if (p.q < nrow) {
        const uint8_t* rowb = ldsb + __mul24(2 * p.ty, RSb) + 2 * (4 * UNR * p.c + kq) * C;
        if (UNR * (p.c + 1) <= fullu) {  // the usual piece: compile-time offsets
#pragma unroll
          for (int jj = 0; jj < UNR; ++jj) bload_unit(rowb + 8 * C * jj, R.b[jj]);
        } else {
#pragma unroll
          for (int jj = 0; jj < UNR; ++jj) bload_unit(UNR * p.c + jj < fullu ? rowb + 8 * C * jj : rowb, R.b[jj]);
        }
      } else {
        const uint8_t* colb = ldsb + 2 * min(4 * fullu + p.ty, a.Wo - 1) * C;
#pragma unroll
        for (int jj = 0; jj < UNR; ++jj)
          bload_unit(colb + __mul24(2 * min(4 * (UNR * p.c + jj) + kq, tha - 1), RSb), R.b[jj]);
      }
    };
    auto mma = [&](const Raw& R, const f32x2 (&av)[UNR]) {
#pragma unroll
      for (int jj = 0; jj < UNR; ++jj) {
        float bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = (float)R.b[jj][t];
        bsum += av[jj];
        if (TAIL) {
          const float bt = (float)R.b[jj][NB - 1];
          atail += av[jj] * f32x2{bt, bt};
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[0][t] = mfma16(av[jj][0], bv[t], acc[0][t]);
          acc[1][t] = mfma16(av[jj][1], bv[t], acc[1][t]);
        }
#ifndef CURLA_WG1_NOGROUP
        // a unit's conversions ahead of its MFMAs (a conversion right in front of the MFMA that reads it costs wait states)
        __builtin_amdgcn_sched_group_barrier(0x002, NB + 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);
#endif
      }
    };
    f32x2 avA[UNR], avB[UNR];
    Raw RA, RB;
    Piece cur;
    cur.q = ABL(2) ? npieces : wave;
    if (cur.q < nrow) cur.ty = cur.q / cpr, cur.c = cur.q - cur.ty * cpr;
    else column_of(cur);
    gload(cur, avA), bload(cur, RA);
    while (cur.q < npieces) {  // (wave-uniform)
      Piece nxt = cur;
      next_piece(nxt);
      gload(nxt, avB), bload(nxt, RB);  // (past the last piece: a column piece outside the band -- zero page, clamped bytes)
      __builtin_amdgcn_sched_barrier(0);
      mma(RA, avA);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
      if (cur.q >= npieces) break;
      next_piece(nxt);
      gload(nxt, avA), bload(nxt, RA);
      __builtin_amdgcn_sched_barrier(0);
      mma(RB, avB);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] *= a.scale;

  bsum[0] += __shfl_xor(bsum[0], 16);
  bsum[0] += __shfl_xor(bsum[0], 32);
  bsum[1] += __shfl_xor(bsum[1], 16);
  bsum[1] += __shfl_xor(bsum[1], 32);
  if (TAIL) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      atail[i] *= a.scale;
      atail[i] += __shfl_xor(atail[i], 16);
      atail[i] += __shfl_xor(atail[i], 32);
    }
  }
  // Cross-wave sum (waves in order: fixed order, reproducible) and the slab.  Every wave deposits its accumulator
  // tiles lane-contiguously (one ds_write_b128 per tile, no index arithmetic), as many tiles per pass as the LDS
  // holds for NW waves; after one barrier the threads add the NW copies of their slot and scatter the four sums
  // straight into the slab -- instead of NW serialised read-modify-write rounds over the output layout.
  const int nw = 32 * C * 9;
  float* slab = a.partial + (size_t)blockIdx.x * (nw + 32);
  if (ABL(4)) {  // timing only: no cross-wave sum, no slab
    float t = bsum[0] + bsum[1] + atail[0] + atail[1];
#pragma unroll
    for (int q = 0; q < 2 * NT; ++q) t += acc[q / NT][q % NT][0] + acc[q / NT][q % NT][3];
    if (t == 12345.678f) slab[tid] = t;
    return;
  }
  f32x4* l4 = reinterpret_cast<f32x4*>(lds);
  const int TC = max(1, min(2 * NT, (int)(a.lds_bytes / (NW * 1024))));  // tiles per pass (1 KB per tile and wave)
  __syncthreads();
  for (int t0 = 0; t0 < 2 * NT; t0 += TC) {
    const int nt = min(TC, 2 * NT - t0);
#pragma unroll
    for (int q = 0; q < 2 * NT; ++q)  // (tile q = mt * NT + t; compile-time register index, runtime range test)
      if (q >= t0 && q < t0 + nt) l4[(wave * TC + (q - t0)) * 64 + lane] = acc[q / NT][q % NT];
    __syncthreads();
    for (int sl = tid; sl < nt * 64; sl += NTHR) {
      const int q = t0 + sl / 64, ln = sl & 63;
      f32x4 v = l4[(0 * TC + (q - t0)) * 64 + ln];
      for (int w = 1; w < NW; ++w) v += l4[(w * TC + (q - t0)) * 64 + ln];
      const int mt = q / NT, t = q - mt * NT;
      const int k = t * 16 + (ln & 15);
      const int dy = k / (3 * C), rr = k - dy * (3 * C);
      if (dy < 3) {
        const int dx = rr / C, c = rr - dx * C;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = mt * 16 + 4 * (ln >> 4) + r;
          slab[(co * C + c) * 9 + dy * 3 + dx] = v[r];
        }
      }
    }
    __syncthreads();
  }
  if (kq == 0) {
    lds[(wave * 2 + 0) * 16 + li] = bsum[0], lds[(wave * 2 + 1) * 16 + li] = bsum[1];
    if (TAIL) lds[NW * 32 + (wave * 2 + 0) * 16 + li] = atail[0], lds[NW * 32 + (wave * 2 + 1) * 16 + li] = atail[1];
  }
  __syncthreads();
  if (tid < 32) {
    float v = lds[tid];  // wave 0: [mt][li] = tid
    for (int w = 1; w < NW; ++w) v += lds[w * 32 + tid];
    slab[nw + tid] = v;
    if (TAIL) {  // the column the tiles leave out: (dy, dx, c) = (2, 2, C-1) of output channel tid
      float t = lds[NW * 32 + tid];
      for (int w = 1; w < NW; ++w) t += lds[NW * 32 + w * 32 + tid];
      slab[(tid * C + (C - 1)) * 9 + 8] = t;
    }
  }
}


// second pass: dW = sum over workgroup slabs.  32 elements x 32 slab-groups per
// block; each group adds its slabs in slab order, the 32 group sums are added in
// group order (fixed order => bitwise reproducible).
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* partial, int nslabs, int nw, float* dw,
                                                            float* db) {
  __shared__ float sm[32][33];
  const int c = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + c;
  const int n = nw + 32;
  float s = 0.f;
  if (i < n) {
    int k = part;
    for (; k + 7 * 32 < nslabs; k += 8 * 32) {  // 8 slabs in flight, added in slab order
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)(k + 32 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; k < nslabs; k += 32) s += partial[(size_t)k * n + i];
  }
  sm[part][c] = s;
  __syncthreads();
  if (part == 0 && i < n) {
    float t = sm[0][c];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += sm[k][c];
    if (i < nw)
      dw[i] = t;
    else
      db[i - nw] = t;
  }
}

// the same for up to kMaxReduceJobs weight gradients in ONE launch: the backward pass of an encoder leaves one slab set
// per conv layer and nothing reads dW before the pass is over, so the per-layer reductions need not be launches of
// their own (each would cost the 4.8 us launch floor for ~1 us of work)
constexpr int kMaxReduceJobs = 8;
struct ReduceJobs {
  const float* partial[kMaxReduceJobs];
  float* dw[kMaxReduceJobs];
  float* db[kMaxReduceJobs];
  int nslabs[kMaxReduceJobs], nw[kMaxReduceJobs], first_block[kMaxReduceJobs + 1];
  int njobs;
};

__global__ __launch_bounds__(1024) void wgrad_reduce_multi_kernel(ReduceJobs J) {
  __shared__ float sm[32][33];
  int j = 0;
  while (j + 1 < J.njobs && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
  const float* partial = J.partial[j];
  const int nslabs = J.nslabs[j], nw = J.nw[j];
  const int c = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int i = ((int)blockIdx.x - J.first_block[j]) * 32 + c;
  const int n = nw + 32;
  float s = 0.f;
  if (i < n) {
    int k = part;
    for (; k + 7 * 32 < nslabs; k += 8 * 32) {  // 8 slabs in flight, added in slab order
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)(k + 32 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; k < nslabs; k += 32) s += partial[(size_t)k * n + i];
  }
  sm[part][c] = s;
  __syncthreads();
  if (part == 0 && i < n) {
    float t = sm[0][c];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += sm[k][c];
    if (i < nw)
      J.dw[j][i] = t;
    else
      J.db[j][i - nw] = t;
  }
}

// ------------------------------ host-side planning ------------------------------
constexpr int kMaxLds = 160 * 1024;
constexpr int kBandPx = 567;  // pixels a band may hold: (567 + 1 slack) * 144 B = 79.9 KB of LDS -> 2 workgroups per CU
static_assert(kBandPx * 8 <= 256 * kMaxPf, "band must fit the staging registers");

// Rows per band for the stride-1 kernels.  The band (th+2 input rows, plus th
// gradient rows for wgrad) must fit kBandPx pixels.  Candidates split Ho into
// nb near-equal bands; the score is the fraction of useful 16-pixel tile slots
// (tiles are dealt to `nslots` wave groups) times the halo re-read factor.
int plan_band_s1(int Ho, int Wo, int px_per_row_extra, int budget_px, int unit_px, int nslots, bool pairs = false) {
  int th_max = 0;
  for (int th = 1; th <= Ho; ++th)
    if ((th + 2) * (Wo + 2) + px_per_row_extra * th <= budget_px) th_max = th;
  if (th_max == 0) return 1;
  int best = th_max;
  double best_eff = -1.0;
  const int nb0 = (Ho + th_max - 1) / th_max;
  for (int nb = nb0; nb <= nb0 + 3 && nb <= Ho; ++nb) {
    const int th = (Ho + nb - 1) / nb;
    if (th > th_max) continue;
    const int nbands = (Ho + th - 1) / th;
    double work = 0, slots = 0;
    for (int bnd = 0; bnd < nbands; ++bnd) {
      const int tha = (bnd == nbands - 1) ? Ho - bnd * th : th;
      const int units = pairs ? tha * ((Wo + 1) / 2) : tha * Wo;  // work items per band (pixel pairs or pixels)
      const int tiles = (units + unit_px - 1) / unit_px;
      work += (pairs ? tha * Wo / 2.0 : tha * Wo) / (double)unit_px;
      slots += ((tiles + nslots - 1) / nslots) * nslots;
    }
    const double eff = work / slots * (double)th / (th + 2) * (1.0 - 0.01 * nbands);  // small per-band fixed cost
    if (eff > best_eff) best_eff = eff, best = th;
  }
  return best;
}

// Bands of the forward / data-gradient kernel: band 0 has h1 rows, the others th (the last what is left), all
// within the LDS budget.  A band's 16-pair tiles are dealt to two wave pairs, so an odd tile count idles one of
// them for a tile; letting band 0 differ (31 rows = 11 + 10 + 10 instead of 11 + 11 + 9) buys even counts.
// Score = useful tile slots / dealt tile slots x rows / staged rows (halo), minus a small per-band cost.
void plan_bands_conv_s1(int Ho, int Wo, int budget_px, int* th_out, int* h1_out, int* nb_out) {
  const int PW = (Wo + 1) / 2;
  int th_max = 0;
  for (int th = 1; th <= Ho; ++th)
    if ((th + 2) * (Wo + 2) <= budget_px) th_max = th;
  if (th_max == 0) {
    *th_out = 1, *h1_out = 1, *nb_out = Ho;
    return;
  }
  double best = -1.0;
  auto slots = [&](int rows) {
    const int tiles = (rows * PW + 15) / 16;
    return (double)((tiles + 1) / 2 * 2);
  };
  for (int th = 1; th <= th_max; ++th)
    for (int variant = 0; variant < 2; ++variant) {
      int h1, nb;
      if (variant == 0) {
        h1 = th, nb = (Ho + th - 1) / th;  // uniform, last band short
      } else {
        nb = (Ho - th) / th + 1;  // first band takes the remainder: th <= h1 < 2 th
        h1 = Ho - (nb - 1) * th;
        if (nb < 2 || h1 > th_max) continue;
      }
      if (h1 >= Ho) h1 = Ho, nb = 1;
      double s = slots(h1);
      int rows_left = Ho - h1;
      for (int b = 1; b < nb; ++b) {
        const int r = rows_left < th ? rows_left : th;
        s += slots(r);
        rows_left -= r;
      }
      const double eff = (Ho * Wo / 32.0) / s * (double)Ho / (Ho + 2.0 * nb) * (1.0 - 0.01 * nb);
      if (eff > best) best = eff, *th_out = th, *h1_out = h1, *nb_out = nb;
    }
}

// Dynamic LDS limit of a kernel: raised once per (kernel, device) to the largest size this library ever asks for --
// never lowered, never set per launch (curla_set_dyn_lds, common.h).
template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > (size_t)kMaxLds) return CURLA_ERR_UNSUPPORTED;
  return curla_set_dyn_lds(reinterpret_cast<const void*>(kernel), kMaxLds);
}

// Which stride-1 forward / data-gradient kernels run: the row-walk form (conv_rw.h) unless CURLA_S1_IMPL=band asks for
// the banded one (kept for A/B timing; both pass the same tests).
bool use_rw() {
  static const bool band = getenv("CURLA_S1_IMPL") && !strcmp(getenv("CURLA_S1_IMPL"), "band");
  return !band;
}
// ... and the weight gradient: CURLA_S1_WGRAD=band keeps the banded kernel beside the row-walk forward / data gradient
// CURLA_C1_U8=rw: the LDS-free row-walk uint8 first-layer forward (conv1_u8_rw.h) instead of the default (the hybrid
// conv1_u8_walk_kernel: crop staged in LDS, row walk out of LDS; CURLA_C1_U8=band: the banded loop).  Measured on
// 1024 + 512 / 512 + 512 samples of configs[1]: alone, re-reading the same ring slots out of the Infinity Cache, the
// LDS-free walk takes 100 / 66 us against the banded loop's 128 / 86; on slots drawn afresh for every launch from a ring
// of gigabytes -- what update() does -- 114 us on average against 104 (hybrid: 102).
bool use_rw_u8() {  // (read at every call -- two per update -- so that one process can run both: the tests do)
  const char* e = getenv("CURLA_C1_U8");
  return use_rw() && e && !strcmp(e, "rw");
}

bool use_rw_wgrad() {
  static const bool band = getenv("CURLA_S1_WGRAD") && !strcmp(getenv("CURLA_S1_WGRAD"), "band");
  return use_rw() && !band;
}

// The row-walk forward keeps (pixel pair, 32 channels) of a whole row in flight per wave; any width works, the strips
// only get more numerous.  Limits: byte offsets inside one sample must fit 31 bits.
bool rw_supported(int Hi, int Wi) { return (long long)(Hi + 2) * (Wi + 2) * 128 < (1LL << 30); }

int launch_rw_fwd(int nlayers, const float* in, const float* const* w, const float* const* bias, float* const* out, int B,
                  const float* in2, const float* const* w2, const float* const* bias2, float* const* out2, int B2, int Hi,
                  int Wi, bool owned, hipStream_t st) {
  rw::Args A;
  A.nlayers = nlayers;
  for (int l = 0; l < rw::kMaxLayers; ++l) {
    const bool on = l < nlayers;
    const int hi = Hi - 2 * l, wi = Wi - 2 * l;
    A.g[l] = on ? rw::plan(hi, wi, hi - 2, wi - 2) : rw::Geom{};
    A.p[l][0] = on ? rw::Problem{l == 0 ? in : out[l - 1], w[l], bias[l], out[l], B} : rw::Problem{};
    A.p[l][1] = (on && B2 > 0) ? rw::Problem{l == 0 ? in2 : out2[l - 1], w2[l], bias2[l], out2[l], B2} : rw::Problem{};
    if (on && (hi < 3 || wi < 3)) return CURLA_ERR_UNSUPPORTED;
  }
  if (!rw_supported(Hi, Wi)) return CURLA_ERR_UNSUPPORTED;
  const int cus = curla_cu_count();
  const int bmax = B > B2 ? B : B2;
  const int grid = owned ? cus : (bmax < cus ? bmax : cus);
  const size_t lds = (size_t)(B2 > 0 ? 2 : 1) * rw::kWFloats * sizeof(float);
  int rc = set_lds(conv_rw_fwd_kernel, lds);
  if (rc != CURLA_OK) return rc;
  hipLaunchKernelGGL(conv_rw_fwd_kernel, dim3(grid), dim3(512), lds, st, A);
  return curla_launch_status();
}

// data-gradient arguments: input = the layer's output gradient [B][Ho][Wo][32], output [B][Ho+2][Wo+2][32]
rw::Args rw_dgrad_args(const float* g, const float* w, const float* act_below, float* gin, int B, int Ho, int Wo) {
  rw::Args A;
  A.nlayers = 1;
  for (int l = 0; l < rw::kMaxLayers; ++l) A.g[l] = rw::Geom{}, A.p[l][0] = rw::Problem{}, A.p[l][1] = rw::Problem{};
  A.g[0] = rw::plan(Ho, Wo, Ho + 2, Wo + 2);
  A.p[0][0] = rw::Problem{g, w, act_below, gin, B};
  return A;
}

int launch_conv_s1(int mode, const float* in, const float* w, const float* aux, float* out, int B, int Hs, int Ws,
                   hipStream_t st, const float* in2 = nullptr, const float* w2 = nullptr, const float* aux2 = nullptr,
                   float* out2 = nullptr, int B2 = 0) {
  if (use_rw() && rw_supported(Hs, Ws)) {
    if (mode == MODE_FWD)
      return launch_rw_fwd(1, in, &w, &aux, &out, B, in2, &w2, &aux2, &out2, B2, Hs, Ws, false, st);
    const rw::Args A = rw_dgrad_args(in, w, aux, out, B, Hs, Ws);
    const int cap = 2 * curla_cu_count();
    const size_t lds = rw::kWFloats * sizeof(float);
    int rc = set_lds(conv_rw_dgrad_kernel, lds);
    if (rc != CURLA_OK) return rc;
    hipLaunchKernelGGL(conv_rw_dgrad_kernel, dim3(B < cap ? B : cap), dim3(256), lds, st, A);
    return curla_launch_status();
  }
  ConvS1Args a;
  a.in = in, a.w = w, a.aux = aux, a.out = out;
  a.in2 = in2, a.w2 = w2, a.aux2 = aux2, a.out2 = out2, a.B2 = B2;
  a.B = B, a.Hs = Hs, a.Ws = Ws;
  a.pad = mode == MODE_FWD ? 0 : 2;
  a.Ho = mode == MODE_FWD ? Hs - 2 : Hs + 2;
  a.Wo = mode == MODE_FWD ? Ws - 2 : Ws + 2;
  if (a.Ho <= 0 || a.Wo <= 0 || (a.Wo + 2) * 3 > kBandPx) return CURLA_ERR_UNSUPPORTED;
  plan_bands_conv_s1(a.Ho, a.Wo, kBandPx, &a.th, &a.h1, &a.nbands);
  const int PW = (a.Wo + 1) / 2;
  a.qstep = 32 / PW, a.rstep = 32 - a.qstep * PW;
  a.dbg = ABL_HOST;
  size_t lds = ((size_t)(a.h1 + 2) * (a.Wo + 2) + 1) * kLdsPix * sizeof(float);  // +1 pixel: 4th window pixel of the last pair
  const size_t wl = (size_t)32 * kWStride * sizeof(float);
  if (lds < wl) lds = wl;
  const int nitems = (B + B2) * a.nbands;
  const int grid = nitems < 2 * curla_cu_count() ? nitems : 2 * curla_cu_count();
  int rc;
  if (mode == MODE_FWD) {
    if ((rc = set_lds(conv_s1_kernel<MODE_FWD>, lds)) != CURLA_OK) return rc;
    hipLaunchKernelGGL(conv_s1_kernel<MODE_FWD>, dim3(grid), dim3(256), lds, st, a);
  } else {
    if ((rc = set_lds(conv_s1_kernel<MODE_DGRAD>, lds)) != CURLA_OK) return rc;
    hipLaunchKernelGGL(conv_s1_kernel<MODE_DGRAD>, dim3(grid), dim3(256), lds, st, a);
  }
  return curla_launch_status();
}

int plan_band_conv1(int Ho, int Wo, int Wc, int C, int g_px_per_row, size_t lds_budget) {
  const int RS = ((Wc * C + 3) & ~3) + 4;
  int best = 1;
  double best_eff = -1.0;
  for (int th = 1; th <= Ho; ++th) {
    const size_t bytes = ((size_t)(2 * th + 1) * RS + (size_t)g_px_per_row * th * kLdsPix) * sizeof(float);
    if (bytes > lds_budget) break;
    const int nb = (Ho + th - 1) / th;
    double work = 0, slots = 0;
    for (int bnd = 0; bnd < nb; ++bnd) {
      const int tha = (bnd == nb - 1) ? Ho - bnd * th : th;
      const int tiles = (tha * Wo + 15) / 16;
      work += tha * Wo / 16.0;
      slots += ((tiles + 7) / 8) * 8;
    }
    const double eff = work / slots * (2.0 * th) / (2.0 * th + 1 + 2);
    if (eff > best_eff) best_eff = eff, best = th;
  }
  return best;
}

}  // namespace

// ------------------------------------ C ABI ------------------------------------
extern "C" {

#ifdef CURLA_ABLATE
void curla_debug_ablate(int flags) { g_ablate = flags; }
#endif

int curla_conv3x3_s1_fwd(const float* in, const float* w, const float* bias, float* out, int B, int Hi, int Wi,
                         int channels, void* stream) {
  CURLA_REQUIRE(in && w && bias && out && B > 0 && Hi >= 3 && Wi >= 3);
  if (channels != 32) return CURLA_ERR_UNSUPPORTED;
  CURLA_REQUIRE(aligned16(in) && aligned16(out) && aligned16(bias) && aligned16(w));
  return launch_conv_s1(MODE_FWD, in, w, bias, out, B, Hi, Wi, static_cast<hipStream_t>(stream));
}

int curla_conv3x3_s1_fwd2(const float* in, const float* w, const float* bias, float* out, int B, const float* in2,
                          const float* w2, const float* bias2, float* out2, int B2, int Hi, int Wi, int channels,
                          void* stream) {
  CURLA_REQUIRE(in && w && bias && out && B > 0 && in2 && w2 && bias2 && out2 && B2 > 0 && Hi >= 3 && Wi >= 3);
  if (channels != 32) return CURLA_ERR_UNSUPPORTED;
  CURLA_REQUIRE(aligned16(in) && aligned16(out) && aligned16(bias) && aligned16(w));
  CURLA_REQUIRE(aligned16(in2) && aligned16(out2) && aligned16(bias2) && aligned16(w2));
  return launch_conv_s1(MODE_FWD, in, w, bias, out, B, Hi, Wi, static_cast<hipStream_t>(stream), in2, w2, bias2, out2, B2);
}

int curla_conv3x3_s1_fwd_stack(int nlayers, const float* in, const float* const* w, const float* const* bias,
                               float* const* out, int B, const float* in2, const float* const* w2,
                               const float* const* bias2, float* const* out2, int B2, int Hi, int Wi, int channels,
                               void* stream) {
  CURLA_REQUIRE(nlayers > 0 && nlayers <= kMaxStack && in && w && bias && out && B > 0 && Hi >= 3 && Wi >= 3);
  CURLA_REQUIRE(B2 == 0 || (in2 && w2 && bias2 && out2));
  if (channels != 32) return CURLA_ERR_UNSUPPORTED;
  if (use_rw()) {
    // ownership of samples by workgroups needs whole rounds of the grid (one workgroup per CU) over each minibatch
    const int G1 = curla_cu_count();
    if (B % G1 != 0 || B2 % G1 != 0) return CURLA_ERR_UNSUPPORTED;
    CURLA_REQUIRE(aligned16(in) && (!B2 || aligned16(in2)));
    for (int l = 0; l < nlayers; ++l) {
      CURLA_REQUIRE(w[l] && bias[l] && out[l] && aligned16(w[l]) && aligned16(bias[l]) && aligned16(out[l]));
      CURLA_REQUIRE(!B2 || (w2[l] && bias2[l] && out2[l] && aligned16(w2[l]) && aligned16(bias2[l]) && aligned16(out2[l])));
    }
    if (Hi - 2 * nlayers < 1 || Wi - 2 * nlayers < 1) return CURLA_ERR_UNSUPPORTED;
    return launch_rw_fwd(nlayers, in, w, bias, out, B, in2, w2, bias2, out2, B2, Hi, Wi, true,
                         static_cast<hipStream_t>(stream));
  }
  const int G = 2 * curla_cu_count();
  // ownership of samples by workgroups needs whole rounds of the grid over each minibatch
  if (B % G != 0 || B2 % G != 0) return CURLA_ERR_UNSUPPORTED;
  ConvS1StackArgs S;
  S.nlayers = nlayers, S.B = B, S.B2 = B2, S.Hs0 = Hi, S.Ws0 = Wi;
  S.in0 = in, S.in0_2 = in2;
  CURLA_REQUIRE(aligned16(in) && (!B2 || aligned16(in2)));
  size_t lds = (size_t)32 * kWStride * sizeof(float);
  for (int l = 0; l < kMaxStack; ++l) {
    const bool on = l < nlayers;
    S.w[l] = on ? w[l] : nullptr, S.bias[l] = on ? bias[l] : nullptr, S.out[l] = on ? out[l] : nullptr;
    S.w2[l] = on && B2 ? w2[l] : nullptr, S.bias2[l] = on && B2 ? bias2[l] : nullptr, S.out2[l] = on && B2 ? out2[l] : nullptr;
    S.th[l] = S.h1[l] = S.nbands[l] = 1;
    if (!on) continue;
    CURLA_REQUIRE(S.w[l] && S.bias[l] && S.out[l] && aligned16(S.w[l]) && aligned16(S.bias[l]) && aligned16(S.out[l]));
    CURLA_REQUIRE(!B2 || (S.w2[l] && S.bias2[l] && S.out2[l] && aligned16(S.w2[l]) && aligned16(S.bias2[l]) && aligned16(S.out2[l])));
    const int Ho = Hi - 2 * l - 2, Wo = Wi - 2 * l - 2;
    if (Ho <= 0 || Wo <= 0 || (Wo + 2) * 3 > kBandPx) return CURLA_ERR_UNSUPPORTED;
    plan_bands_conv_s1(Ho, Wo, kBandPx, &S.th[l], &S.h1[l], &S.nbands[l]);
    const size_t need = ((size_t)(S.h1[l] + 2) * (Wo + 2) + 1) * kLdsPix * sizeof(float);
    if (need > lds) lds = need;
  }
  int rc = set_lds(conv_s1_stack_kernel, lds);
  if (rc != CURLA_OK) return rc;
  hipLaunchKernelGGL(conv_s1_stack_kernel, dim3(G), dim3(256), lds, static_cast<hipStream_t>(stream), S);
  return curla_launch_status();
}

int curla_conv3x3_s1_stack_granule(void) { return use_rw() ? curla_cu_count() : 2 * curla_cu_count(); }

int curla_conv3x3_s1_dgrad(const float* g, const float* w, const float* act_below, float* gin, int B, int Ho, int Wo,
                           int channels, void* stream) {
  CURLA_REQUIRE(g && w && act_below && gin && B > 0 && Ho >= 1 && Wo >= 1);
  if (channels != 32) return CURLA_ERR_UNSUPPORTED;
  CURLA_REQUIRE(aligned16(g) && aligned16(gin) && aligned16(act_below) && aligned16(w));
  return launch_conv_s1(MODE_DGRAD, g, w, act_below, gin, B, Ho, Wo, static_cast<hipStream_t>(stream));
}

static int conv1_common_check(const void* src, int src_is_u8, int B, int C, int Hs, int Ws, int Hc, int Wc,
                              const int32_t* h1, const int32_t* w1) {
  CURLA_REQUIRE(src && B > 0 && Hc >= 3 && Wc >= 3 && src_is_u8 >= 0 && src_is_u8 <= 2);
  if (C != 9 && C != 12 && C != 6 && C != 3) return CURLA_ERR_UNSUPPORTED;  // 3 x frame_stack of 1..4
  if (src_is_u8 == 1) {
    CURLA_REQUIRE(Hs >= Hc && Ws >= Wc);
    // the loader rebuilds every 16-byte run from aligned dwords whatever its byte address, so frames of any size
    // work; only the ring's base must be dword-aligned (the first run would otherwise start before the buffer)
    CURLA_REQUIRE((reinterpret_cast<uintptr_t>(src) & 3) == 0);
    (void)h1, (void)w1;
  }
  return CURLA_OK;
}

#define CURLA_DISPATCH_SRC(CC, KIND, KERNEL, ...)                  \
  do {                                                             \
    if ((KIND) == 1) {                                             \
      KERNEL(SRC_U8, CC, __VA_ARGS__);                             \
    } else if ((KIND) == 2) {                                      \
      KERNEL(SRC_NHWC, CC, __VA_ARGS__);                           \
    } else {                                                       \
      KERNEL(SRC_F32, CC, __VA_ARGS__);                            \
    }                                                              \
  } while (0)
#define CURLA_DISPATCH_C(C, KIND, KERNEL, ...)                     \
  do {                                                             \
    if ((C) == 9) {                                                \
      CURLA_DISPATCH_SRC(9, KIND, KERNEL, __VA_ARGS__);            \
    } else if ((C) == 12) {                                        \
      CURLA_DISPATCH_SRC(12, KIND, KERNEL, __VA_ARGS__);           \
    } else if ((C) == 6) {                                         \
      CURLA_DISPATCH_SRC(6, KIND, KERNEL, __VA_ARGS__);            \
    } else {                                                       \
      CURLA_DISPATCH_SRC(3, KIND, KERNEL, __VA_ARGS__);            \
    }                                                              \
  } while (0)

#define CONV1_FWD_LAUNCH(SRC, CC, grid, lds, st, a)                                       \
  {                                                                                       \
    rc = set_lds(conv1_fwd_kernel<SRC, CC>, lds);                                         \
    if (rc == CURLA_OK) hipLaunchKernelGGL((conv1_fwd_kernel<SRC, CC>), dim3(grid), dim3(512), lds, st, a); \
  }

struct Conv1Second {
  const int64_t* idx;
  const int32_t* h1;
  const int32_t* w1;
  const float* w;
  const float* bias;
  float* out;
  int B;
};

static int conv1_fwd_impl(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                          const float* w, const float* bias, float* out, int B, int C, int Hs, int Ws, int Hc, int Wc,
                          int channels, float scale, void* stream, const Conv1Second* second) {
  CURLA_REQUIRE(w && bias && out);
  if (channels != 32) return CURLA_ERR_UNSUPPORTED;
  int rc = conv1_common_check(src, src_kind, B, C, Hs, Ws, Hc, Wc, h1, w1);
  if (rc != CURLA_OK) return rc;
  Conv1Args a;
  a.src = src, a.idx = idx, a.h1 = h1, a.w1 = w1, a.w = w, a.bias = bias, a.out = out;
  a.B = B, a.C = C, a.Hs = Hs, a.Ws = Ws, a.Hc = Hc, a.Wc = Wc;
  a.Ho = (Hc - 3) / 2 + 1, a.Wo = (Wc - 3) / 2 + 1;
  a.scale = scale;
  a.dbg = ABL_HOST;
  a.idx2 = second ? second->idx : nullptr, a.h1_2 = second ? second->h1 : nullptr, a.w1_2 = second ? second->w1 : nullptr;
  a.w2 = second ? second->w : nullptr, a.bias2 = second ? second->bias : nullptr, a.out2 = second ? second->out : nullptr;
  a.B2 = second ? second->B : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t wl = (size_t)32 * C * 9 * sizeof(float);
  if (second && src_kind != 1) return CURLA_ERR_UNSUPPORTED;
  if (src_kind == 1 && use_rw_u8() && (long long)Hs * Ws * C < (1LL << 30) &&
      (long long)(B + a.B2) * a.Ho * ((a.Wo + 15) / 16 + 1) < (1LL << 28)) {
    // uint8 ring: row walk, nothing staged (conv1_u8_rw.h)
    rw::Conv1U8Args ra;
    ra.src = static_cast<const uint8_t*>(src);
    ra.p[0] = rw::Conv1U8Problem{idx, h1, w1, w, bias, out, B};
    ra.p[1] = rw::Conv1U8Problem{a.idx2, a.h1_2, a.w1_2, a.w2, a.bias2, a.out2, a.B2};
    ra.Hs = Hs, ra.Ws = Ws, ra.Ho = a.Ho, ra.Wo = a.Wo, ra.scale = scale;
    ra.g.Hi = Hc, ra.g.Wi = Wc, ra.g.Ho = a.Ho, ra.g.Wo = a.Wo;
    rw::plan_units(ra.g, a.Ho, a.Wo, 16);
    // kC1U8PerCU workgroups per CU; fewer when a workgroup's share of the pool would drop below a few steps per wave
    const int cap = kC1U8PerCU * curla_cu_count();
    const long long pool = (long long)(B + a.B2) * ra.g.steps;
    const int want = (int)((pool + 63) / 64);
    const int grid_rw = want < cap ? (want < 1 ? 1 : want) : cap;
#define CONV1_U8_RW_LAUNCH(CC) hipLaunchKernelGGL((conv1_u8_rw_fwd_kernel<CC>), dim3(grid_rw), dim3(kC1U8Threads), 0, st, ra)
    if (C == 9) CONV1_U8_RW_LAUNCH(9); else if (C == 12) CONV1_U8_RW_LAUNCH(12); else if (C == 6) CONV1_U8_RW_LAUNCH(6); else CONV1_U8_RW_LAUNCH(3);
#undef CONV1_U8_RW_LAUNCH
    return curla_launch_status();
  }
  if (src_kind == 1 && !(ABL_HOST & 128)) {
    // uint8 ring: the band stays bytes in LDS; the tallest band that leaves room for two workgroups per CU
    const int RSb = ((Wc * C + 15) & ~15) + 16;
    int th = a.Ho;
    while (th > 1 && (size_t)(2 * th + 1) * RSb > 76 * 1024) --th;
    a.nbands = (a.Ho + th - 1) / th;
    a.th = (a.Ho + a.nbands - 1) / a.nbands;  // near-equal bands
    size_t lds = (size_t)(2 * a.th + 1) * RSb + 32;
    const size_t wl8 = (size_t)3 * ((3 * C + 3) & ~3) * 32 * sizeof(float);  // the kernel's k-major weight image
    if (lds < wl8) lds = wl8;
    const int grid = (B + a.B2) * a.nbands;
#define CONV1_U8_LAUNCH(CC)                                                                                 \
  {                                                                                                         \
    rc = set_lds(conv1_fwd_u8_kernel<CC>, lds);                                                             \
    if (rc == CURLA_OK) hipLaunchKernelGGL((conv1_fwd_u8_kernel<CC>), dim3(grid), dim3(512), lds, st, a);   \
  }
    // one band = the whole crop in LDS: the hybrid form (row walk out of LDS) unless CURLA_C1_U8=band asks for the old loop
    const char* impl = getenv("CURLA_C1_U8");
    // (its staging gives a lane one 16-byte run of a crop row: rows of at most 64 runs)
    if (a.nbands == 1 && Wc * C <= 64 * 16 && use_rw() && !(impl && !strcmp(impl, "band"))) {
      rw::Geom G;
      G.Hi = Hc, G.Wi = Wc, G.Ho = a.Ho, G.Wo = a.Wo;
      rw::plan_units(G, a.Ho, a.Wo, 16);
#define CONV1_U8_WALK(CC)                                                                                     \
  {                                                                                                           \
    rc = set_lds(conv1_u8_walk_kernel<CC>, lds);                                                              \
    if (rc == CURLA_OK) hipLaunchKernelGGL((conv1_u8_walk_kernel<CC>), dim3(grid), dim3(512), lds, st, a, G); \
  }
      if (C == 9) CONV1_U8_WALK(9) else if (C == 12) CONV1_U8_WALK(12) else if (C == 6) CONV1_U8_WALK(6) else CONV1_U8_WALK(3)
#undef CONV1_U8_WALK
      if (rc != CURLA_OK) return rc;
      return curla_launch_status();
    }
    if (C == 9) CONV1_U8_LAUNCH(9) else if (C == 12) CONV1_U8_LAUNCH(12) else if (C == 6) CONV1_U8_LAUNCH(6) else CONV1_U8_LAUNCH(3)
#undef CONV1_U8_LAUNCH
    if (rc != CURLA_OK) return rc;
    return curla_launch_status();
  }
  if (src_kind == 2 && use_rw() && (long long)Hc * Wc * C * 4 < (1LL << 30)) {
    // float NHWC minibatch: row walk, nothing staged (conv1_rw.h)
    rw::Conv1Args ra;
    ra.src = static_cast<const float*>(src), ra.w = w, ra.bias = bias, ra.out = out;
    ra.B = B, ra.Hc = Hc, ra.Wc = Wc, ra.Ho = a.Ho, ra.Wo = a.Wo, ra.scale = scale;
    ra.g.Hi = Hc, ra.g.Wi = Wc, ra.g.Ho = a.Ho, ra.g.Wo = a.Wo;
    rw::plan_units(ra.g, a.Ho, a.Wo, 16);
    const int cap = curla_cu_count();
    const int grid_rw = B < cap ? B : cap;
#define CONV1_RW_LAUNCH(CC) hipLaunchKernelGGL((conv1_rw_fwd_kernel<CC>), dim3(grid_rw), dim3(512), 0, st, ra)
    if (C == 12) CONV1_RW_LAUNCH(12); else if (C == 9) CONV1_RW_LAUNCH(9); else if (C == 6) CONV1_RW_LAUNCH(6); else CONV1_RW_LAUNCH(3);
#undef CONV1_RW_LAUNCH
    return curla_launch_status();
  }
  // two workgroups per CU so one stages while the other computes
  a.th = plan_band_conv1(a.Ho, a.Wo, Wc, C, 0, 76 * 1024);
  a.nbands = (a.Ho + a.th - 1) / a.th;
  const int RS = ((Wc * C + 3) & ~3) + 4;
  size_t lds = ((size_t)(2 * a.th + 1) * RS + 8) * sizeof(float);
  if (lds < wl) lds = wl;
  const int nitems = B * a.nbands;
  const int grid = nitems < 2 * curla_cu_count() ? nitems : 2 * curla_cu_count();
  CURLA_DISPATCH_C(C, src_kind, CONV1_FWD_LAUNCH, grid, lds, st, a);
  if (rc != CURLA_OK) return rc;
  return curla_launch_status();
}

int curla_conv1_fwd(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                    const float* w, const float* bias, float* out, int B, int C, int Hs, int Ws, int Hc, int Wc,
                    int channels, float scale, void* stream) {
  return conv1_fwd_impl(src, src_kind, idx, h1, w1, w, bias, out, B, C, Hs, Ws, Hc, Wc, channels, scale, stream, nullptr);
}

int curla_conv1_fwd2(const uint8_t* ring, const int64_t* idx, const int32_t* h1, const int32_t* w1, const float* w,
                     const float* bias, float* out, int B, const int64_t* idx2, const int32_t* h1_2,
                     const int32_t* w1_2, const float* w2, const float* bias2, float* out2, int B2, int C, int Hs, int Ws,
                     int Hc, int Wc, int channels, float scale, void* stream) {
  CURLA_REQUIRE(w2 && bias2 && out2 && B2 > 0);
  Conv1Second sec{idx2, h1_2, w1_2, w2, bias2, out2, B2};
  return conv1_fwd_impl(ring, 1, idx, h1, w1, w, bias, out, B, C, Hs, Ws, Hc, Wc, channels, scale, stream, &sec);
}

// workspace (floats) the weight-gradient kernels need for their per-workgroup slabs
size_t curla_conv_wgrad_workspace_floats(int cin) {
  return (size_t)4 * curla_cu_count() * ((size_t)32 * cin * 9 + 32);  // at most four workgroups (slabs) per CU
}

// which pair walk the weight-gradient body uses (see wgrad_s1_body): scalar for rows of at least 8 pairs
static int wgrad_walk(int Wo) {
  static const bool off = getenv("CURLA_WGRAD_WALK") && atoi(getenv("CURLA_WGRAD_WALK")) == 0;  // tuning aid
  if (off || (Wo + 1) / 2 < 8) return 0;
  return (Wo & 1) ? 2 : 1;
}

static int launch_wgrad_s1(const float* in, const float* g, float* workspace, int B, int Hi, int Wi, int channels,
                           hipStream_t st, int* nslabs) {
  CURLA_REQUIRE(in && g && workspace && B > 0 && Hi >= 3 && Wi >= 3);
  if (channels != 32) return CURLA_ERR_UNSUPPORTED;
  CURLA_REQUIRE(aligned16(in) && aligned16(g));
  if (use_rw_wgrad() && rw_supported(Hi, Wi)) {
    rw::WgradArgs ra{in, g, workspace, B, Hi, Wi, Hi - 2, Wi - 2, rw::plan4(Hi, Wi, Hi - 2, Wi - 2)};
    const int cap = 2 * curla_cu_count();
    const int grid = B < cap ? B : cap;
    const size_t lds = kPartialS1 * sizeof(float);
    int rc = set_lds(wgrad_rw_kernel, lds);
    if (rc != CURLA_OK) return rc;
    hipLaunchKernelGGL(wgrad_rw_kernel, dim3(grid), dim3(256), lds, st, ra);
    *nslabs = grid;
    return curla_launch_status();
  }
  WgradS1Args a;
  a.in = in, a.g = g, a.partial = workspace;
  a.B = B, a.Hi = Hi, a.Wi = Wi, a.Ho = Hi - 2, a.Wo = Wi - 2;
  if ((a.Wo + 2) * 3 > kBandPx) return CURLA_ERR_UNSUPPORTED;
  a.th = plan_band_s1(a.Ho, a.Wo, 0, kBandPx, 8, 1, /*pairs=*/true);
  a.nbands = (a.Ho + a.th - 1) / a.th;
  size_t lds = (size_t)((a.th + 2) * Wi + 1) * kLdsPix * sizeof(float);  // +1 pixel: 4th window pixel of the last pair
  if (lds < kPartialS1 * sizeof(float)) lds = kPartialS1 * sizeof(float);
  const int nitems = B * a.nbands;
  const int grid = nitems < 2 * curla_cu_count() ? nitems : 2 * curla_cu_count();
  int rc;
  switch (wgrad_walk(a.Wo)) {
    case 1:
      if ((rc = set_lds(wgrad_s1_kernel<1>, lds)) != CURLA_OK) return rc;
      hipLaunchKernelGGL(wgrad_s1_kernel<1>, dim3(grid), dim3(256), lds, st, a);
      break;
    case 2:
      if ((rc = set_lds(wgrad_s1_kernel<2>, lds)) != CURLA_OK) return rc;
      hipLaunchKernelGGL(wgrad_s1_kernel<2>, dim3(grid), dim3(256), lds, st, a);
      break;
    default:
      if ((rc = set_lds(wgrad_s1_kernel<0>, lds)) != CURLA_OK) return rc;
      hipLaunchKernelGGL(wgrad_s1_kernel<0>, dim3(grid), dim3(256), lds, st, a);
  }
  *nslabs = grid;
  return curla_launch_status();
}

int curla_conv3x3_s1_wgrad(const float* in, const float* g, float* dw, float* db, float* workspace, int B, int Hi,
                           int Wi, int channels, void* stream) {
  CURLA_REQUIRE(dw && db);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int grid = 0;
  int rc = launch_wgrad_s1(in, g, workspace, B, Hi, Wi, channels, st, &grid);
  if (rc != CURLA_OK) return rc;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((kPartialS1 + 31) / 32), dim3(1024), 0, st, workspace, grid, 32 * 288,
                     dw, db);
  return curla_launch_status();
}

int curla_conv3x3_s1_wgrad_slabs(const float* in, const float* g, float* workspace, int B, int Hi, int Wi, int channels,
                                 int* nslabs, void* stream) {
  CURLA_REQUIRE(nslabs);
  return launch_wgrad_s1(in, g, workspace, B, Hi, Wi, channels, static_cast<hipStream_t>(stream), nslabs);
}

int curla_conv3x3_s1_bwd_slabs(const float* in, const float* g, const float* w, float* gin, float* workspace, int B, int Hi,
                               int Wi, int channels, int* nslabs, void* stream) {
  CURLA_REQUIRE(in && g && w && gin && workspace && nslabs && B > 0 && Hi >= 3 && Wi >= 3);
  if (channels != 32) return CURLA_ERR_UNSUPPORTED;
  CURLA_REQUIRE(aligned16(in) && aligned16(g) && aligned16(w) && aligned16(gin));
  const int Ho = Hi - 2, Wo = Wi - 2;
  if (use_rw_wgrad() && rw_supported(Hi, Wi)) {
    // both halves in their row-walk forms: equal numbers of workgroups of each kind (the two do about the same
    // number of MFMAs per sample), each owning samples k, k + n, ...
    static const int split_rw = getenv("CURLA_BWD_SPLIT") ? atoi(getenv("CURLA_BWD_SPLIT")) : -1;
    const bool split2 = split_rw >= 0 ? split_rw != 0 : (long long)B * Ho * Wo <= (1LL << 20);
    const int cap2 = split2 ? curla_cu_count() : 2 * curla_cu_count();
    const int n2 = B < cap2 ? B : cap2;
    rw::WgradArgs wr{in, g, workspace, B, Hi, Wi, Ho, Wo, rw::plan4(Hi, Wi, Ho, Wo)};
    const rw::Args dr = rw_dgrad_args(g, w, in, gin, B, Ho, Wo);
    size_t lds2 = rw::kWFloats * sizeof(float);
    if (lds2 < kPartialS1 * sizeof(float)) lds2 = kPartialS1 * sizeof(float);
    int rc2 = set_lds(bwd_rw2_kernel, lds2);
    if (rc2 != CURLA_OK) return rc2;
    hipLaunchKernelGGL(bwd_rw2_kernel, dim3(2 * n2), dim3(256), lds2, static_cast<hipStream_t>(stream), wr, dr, n2);
    *nslabs = n2;
    return curla_launch_status();
  }
  // weight-gradient part (as launch_wgrad_s1)
  WgradS1Args wa;
  wa.in = in, wa.g = g, wa.partial = workspace;
  wa.B = B, wa.Hi = Hi, wa.Wi = Wi, wa.Ho = Ho, wa.Wo = Wo;
  if ((Wo + 2) * 3 > kBandPx) return CURLA_ERR_UNSUPPORTED;
  wa.th = plan_band_s1(Ho, Wo, 0, kBandPx, 8, 1, /*pairs=*/true);
  wa.nbands = (Ho + wa.th - 1) / wa.th;
  size_t lds_w = (size_t)((wa.th + 2) * Wi + 1) * kLdsPix * sizeof(float);
  if (lds_w < kPartialS1 * sizeof(float)) lds_w = kPartialS1 * sizeof(float);
  // grid: 2 x CUs workgroups of each kind, run one kind after the other -- or, for short launches, ONE workgroup of
  // each kind per CU side by side (blocks 0..CUs-1 and CUs..2 CUs-1; both persistent over their kind's items).  Side
  // by side the layer takes as long (143 / 161 / 182 us at B = 512: a CU shared between the two kinds is no busier
  // than one shared by two of a kind), but the weight gradient leaves half as many slabs for the reduction to read
  // (14 -> 10 us per backward pass).  Long launches keep the 2 + 2 form: there a few percent of imbalance between the
  // kinds (the tail runs at one workgroup per CU) costs more than the slabs (B = 1024 at 81 x 81: +0.3 ms per update).
  static const int split_env = getenv("CURLA_BWD_SPLIT") ? atoi(getenv("CURLA_BWD_SPLIT")) : -1;  // tuning aid: 0 / 1
  const bool split = split_env >= 0 ? split_env != 0 : (long long)B * Ho * Wo <= (1LL << 20);
  const int cap = split ? curla_cu_count() : 2 * curla_cu_count();
  const int items_w = B * wa.nbands;
  const int nw = items_w < cap ? items_w : cap;
  if (use_rw() && rw_supported(Hi, Wi)) {
    // data gradient in its row-walk form: workgroup k of its block range owns samples k, k + nd, ...
    const rw::Args ra = rw_dgrad_args(g, w, in, gin, B, Ho, Wo);
    const int nd_rw = B < cap ? B : cap;
    size_t lds_rw = rw::kWFloats * sizeof(float);
    if (lds_rw < lds_w) lds_rw = lds_w;
    int rc_rw;
    hipStream_t st_rw = static_cast<hipStream_t>(stream);
    switch (wgrad_walk(Wo)) {
      case 1:
        if ((rc_rw = set_lds(bwd_rw_kernel<1>, lds_rw)) != CURLA_OK) return rc_rw;
        hipLaunchKernelGGL(bwd_rw_kernel<1>, dim3(nw + nd_rw), dim3(256), lds_rw, st_rw, wa, ra, nw);
        break;
      case 2:
        if ((rc_rw = set_lds(bwd_rw_kernel<2>, lds_rw)) != CURLA_OK) return rc_rw;
        hipLaunchKernelGGL(bwd_rw_kernel<2>, dim3(nw + nd_rw), dim3(256), lds_rw, st_rw, wa, ra, nw);
        break;
      default:
        if ((rc_rw = set_lds(bwd_rw_kernel<0>, lds_rw)) != CURLA_OK) return rc_rw;
        hipLaunchKernelGGL(bwd_rw_kernel<0>, dim3(nw + nd_rw), dim3(256), lds_rw, st_rw, wa, ra, nw);
    }
    *nslabs = nw;
    return curla_launch_status();
  }
  // data-gradient part (as launch_conv_s1 in MODE_DGRAD: input = the output gradient [B][Ho][Wo], output [B][Hi][Wi])
  ConvS1Args da;
  da.in = g, da.w = w, da.aux = in, da.out = gin;
  da.in2 = nullptr, da.w2 = nullptr, da.aux2 = nullptr, da.out2 = nullptr, da.B2 = 0;
  da.B = B, da.Hs = Ho, da.Ws = Wo, da.pad = 2, da.Ho = Hi, da.Wo = Wi;
  if ((da.Wo + 2) * 3 > kBandPx) return CURLA_ERR_UNSUPPORTED;
  plan_bands_conv_s1(da.Ho, da.Wo, kBandPx, &da.th, &da.h1, &da.nbands);
  const int PW = (da.Wo + 1) / 2;
  da.qstep = 32 / PW, da.rstep = 32 - da.qstep * PW;
  da.dbg = 0;
  size_t lds_d = ((size_t)(da.h1 + 2) * (da.Wo + 2) + 1) * kLdsPix * sizeof(float);
  const size_t wl = (size_t)32 * kWStride * sizeof(float);
  if (lds_d < wl) lds_d = wl;
  const int items_d = B * da.nbands;
  const int nd = items_d < cap ? items_d : cap;
  const size_t lds = lds_w > lds_d ? lds_w : lds_d;
  int rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (wgrad_walk(Wo)) {
    case 1:
      if ((rc = set_lds(bwd_s1_kernel<1>, lds)) != CURLA_OK) return rc;
      hipLaunchKernelGGL(bwd_s1_kernel<1>, dim3(nw + nd), dim3(256), lds, st, wa, da, nw);
      break;
    case 2:
      if ((rc = set_lds(bwd_s1_kernel<2>, lds)) != CURLA_OK) return rc;
      hipLaunchKernelGGL(bwd_s1_kernel<2>, dim3(nw + nd), dim3(256), lds, st, wa, da, nw);
      break;
    default:
      if ((rc = set_lds(bwd_s1_kernel<0>, lds)) != CURLA_OK) return rc;
      hipLaunchKernelGGL(bwd_s1_kernel<0>, dim3(nw + nd), dim3(256), lds, st, wa, da, nw);
  }
  *nslabs = nw;
  return curla_launch_status();
}

int curla_wgrad_reduce_multi(int njobs, const float* const* slabs, const int* nslabs, const int* nw, float* const* dw,
                             float* const* db, void* stream) {
  CURLA_REQUIRE(njobs > 0 && njobs <= kMaxReduceJobs && slabs && nslabs && nw && dw && db);
  ReduceJobs J;
  int blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    CURLA_REQUIRE(slabs[j] && dw[j] && db[j] && nslabs[j] > 0 && nw[j] > 0);
    J.partial[j] = slabs[j], J.dw[j] = dw[j], J.db[j] = db[j], J.nslabs[j] = nslabs[j], J.nw[j] = nw[j];
    J.first_block[j] = blocks;
    blocks += (nw[j] + 32 + 31) / 32;
  }
  J.first_block[njobs] = blocks;
  J.njobs = njobs;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(1024), 0, static_cast<hipStream_t>(stream), J);
  return curla_launch_status();
}

#define WGRAD1_LAUNCH(SRC, CC, grid, lds, st, a)                                        \
  {                                                                                     \
    rc = set_lds(wgrad1_kernel<SRC, CC>, lds);                                          \
    if (rc == CURLA_OK) hipLaunchKernelGGL((wgrad1_kernel<SRC, CC>), dim3(grid), dim3(512), lds, st, a); \
  }

static int launch_wgrad1(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                         const float* g, float* workspace, int B, int C, int Hs, int Ws, int Hc, int Wc, int channels,
                         float scale, void* stream, int* nslabs) {
  CURLA_REQUIRE(g && workspace);
  if (channels != 32) return CURLA_ERR_UNSUPPORTED;
  int rc = conv1_common_check(src, src_kind, B, C, Hs, Ws, Hc, Wc, h1, w1);
  if (rc != CURLA_OK) return rc;
  CURLA_REQUIRE(aligned16(g));
  Wgrad1Args a;
  a.src = src, a.idx = idx, a.h1 = h1, a.w1 = w1, a.g = g, a.partial = workspace;
  a.B = B, a.C = C, a.Hs = Hs, a.Ws = Ws, a.Hc = Hc, a.Wc = Wc;
  a.Ho = (Hc - 3) / 2 + 1, a.Wo = (Wc - 3) / 2 + 1;
  a.scale = scale;
  a.lds_bytes = 0;
  a.dbg = ABL_HOST;
  const int nw = 32 * C * 9;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int grid;
  if (src_kind == 2 && use_rw() && (long long)Hc * Wc * C * 4 < (1LL << 30)) {
    // float NHWC minibatch: row walk, nothing staged (conv1_rw.h)
    rw::Wgrad1Args ra;
    ra.src = static_cast<const float*>(src), ra.g = g, ra.partial = workspace;
    ra.B = B, ra.Hc = Hc, ra.Wc = Wc, ra.Ho = a.Ho, ra.Wo = a.Wo, ra.scale = scale;
    ra.gg.Hi = Hc, ra.gg.Wi = Wc, ra.gg.Ho = a.Ho, ra.gg.Wo = a.Wo;
    rw::plan_units(ra.gg, a.Ho, a.Wo, 4);
    const int cap = curla_cu_count();
    grid = B < cap ? B : cap;
    const size_t lds = (size_t)(nw + 32) * sizeof(float);
#define WGRAD1_RW_LAUNCH(CC)                                                                     \
  {                                                                                              \
    rc = set_lds(wgrad1_rw_kernel<CC>, lds);                                                     \
    if (rc == CURLA_OK) hipLaunchKernelGGL((wgrad1_rw_kernel<CC>), dim3(grid), dim3(512), lds, st, ra); \
  }
    if (C == 12) WGRAD1_RW_LAUNCH(12) else if (C == 9) WGRAD1_RW_LAUNCH(9) else if (C == 6) WGRAD1_RW_LAUNCH(6) else WGRAD1_RW_LAUNCH(3)
#undef WGRAD1_RW_LAUNCH
    if (rc != CURLA_OK) return rc;
    *nslabs = grid;
    return curla_launch_status();
  }
  if (src_kind == 1 && !(ABL_HOST & 256)) {
    // uint8 ring, input band kept as bytes: two 512-thread workgroups per CU (<= 76 KB of LDS each).
    const int RSb = ((Wc * C + 15) & ~15) + 16;
    auto band_bytes = [&](int th) { return (size_t)(2 * th + 1) * RSb; };
    // (measured at 84x84x9, B = 512: 71.9 us with two 512-thread workgroups per CU, 77.4 us with four 256-thread ones:
    // the shorter bands' extra halo rows and slabs cost more than the finer interleaving buys)
    int nwaves = 8;
    size_t budget = 76 * 1024;
    if ((ABL_HOST & 1024) && band_bytes(4) <= 38 * 1024) nwaves = 4, budget = 38 * 1024;
    int th = a.Ho;
    while (th > 1 && band_bytes(th) > budget) --th;
    a.nbands = (a.Ho + th - 1) / th;
    a.th = (a.Ho + a.nbands - 1) / a.nbands;
    size_t lds = (((size_t)(2 * a.th + 1) * RSb + 15) & ~(size_t)15) + 32;
    if (lds < (size_t)(nw + 32) * sizeof(float)) lds = (size_t)(nw + 32) * sizeof(float);
    if (lds < (size_t)nwaves * 1024) lds = (size_t)nwaves * 1024;  // the final cross-wave sum: one tile of every wave
    {  // ... and all of a wave's tiles at once (ONE pass of the sum: 48.3 -> 44.8 us at 84x84x9) where two workgroups
       // of that size still share a CU
      const int k9 = 9 * C, ntiles = 2 * ((k9 % 16 == 1) ? k9 / 16 : (k9 + 15) / 16);
      const size_t one_pass = (size_t)ntiles * nwaves * 1024;
      if (nwaves == 8 && one_pass <= 80 * 1024 && lds < one_pass) lds = one_pass;
    }
    a.lds_bytes = (unsigned)lds;
    const int nitems = B * a.nbands;
    const int per_cu = nwaves == 4 ? 4 : 2;
    grid = nitems < per_cu * curla_cu_count() ? nitems : per_cu * curla_cu_count();
#define WGRAD1_U8_LAUNCH(CC)                                                                                       \
  {                                                                                                                \
    if (nwaves == 4) {                                                                                             \
      rc = set_lds(wgrad1_u8_kernel<CC, 4>, lds);                                                                  \
      if (rc == CURLA_OK) hipLaunchKernelGGL((wgrad1_u8_kernel<CC, 4>), dim3(grid), dim3(256), lds, st, a);        \
    } else {                                                                                                       \
      rc = set_lds(wgrad1_u8_kernel<CC, 8>, lds);                                                                  \
      if (rc == CURLA_OK) hipLaunchKernelGGL((wgrad1_u8_kernel<CC, 8>), dim3(grid), dim3(512), lds, st, a);        \
    }                                                                                                              \
  }
    if (C == 9) WGRAD1_U8_LAUNCH(9) else if (C == 12) WGRAD1_U8_LAUNCH(12) else if (C == 6) WGRAD1_U8_LAUNCH(6) else WGRAD1_U8_LAUNCH(3)
#undef WGRAD1_U8_LAUNCH
  } else {
    a.th = plan_band_conv1(a.Ho, a.Wo, Wc, C, 0, 150 * 1024);  // input rows only, one workgroup per CU
    a.nbands = (a.Ho + a.th - 1) / a.th;
    const int RS = ((Wc * C + 3) & ~3) + 4;
    size_t lds = ((size_t)(2 * a.th + 1) * RS + 8) * sizeof(float);
    if (lds < (size_t)(nw + 32) * sizeof(float)) lds = (size_t)(nw + 32) * sizeof(float);
    const int nitems = B * a.nbands;
    grid = nitems < curla_cu_count() ? nitems : curla_cu_count();
    CURLA_DISPATCH_C(C, src_kind, WGRAD1_LAUNCH, grid, lds, st, a);
  }
  if (rc != CURLA_OK) return rc;
  *nslabs = grid;
  return curla_launch_status();
}

int curla_conv1_wgrad(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                      const float* g, float* dw, float* db, float* workspace, int B, int C, int Hs, int Ws, int Hc,
                      int Wc, int channels, float scale, void* stream) {
  CURLA_REQUIRE(dw && db);
  int grid = 0;
  int rc = launch_wgrad1(src, src_kind, idx, h1, w1, g, workspace, B, C, Hs, Ws, Hc, Wc, channels, scale, stream, &grid);
  if (rc != CURLA_OK) return rc;
  const int nw = 32 * C * 9;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((nw + 32 + 31) / 32), dim3(1024), 0, static_cast<hipStream_t>(stream),
                     workspace, grid, nw, dw, db);
  return curla_launch_status();
}

int curla_conv1_wgrad_slabs(const void* src, int src_kind, const int64_t* idx, const int32_t* h1, const int32_t* w1,
                            const float* g, float* workspace, int B, int C, int Hs, int Ws, int Hc, int Wc, int channels,
                            float scale, int* nslabs, void* stream) {
  CURLA_REQUIRE(nslabs);
  return launch_wgrad1(src, src_kind, idx, h1, w1, g, workspace, B, C, Hs, Ws, Hc, Wc, channels, scale, stream, nslabs);
}



}  // extern "C"
