// Weight gradient of the stride-1 32->32 3x3 convolution, "row walk" form, gfx950.  Included by conv.hip (after
// conv_rw.h and the slab layout kPartialS1).
//
//   dW[co][ci][dy][dx] = sum over (sample, y, x) of g[y][x][co] * in[y + dy][x + dx][ci]        (encoder.py:59-63, autograd)
//
// GEMM view as in wgrad_s1_body: D[co][ci] per (row tap, Winograd position), K = pixel pairs, 4 per MFMA; Winograd
// F(3,2) along x: for a pair of gradient pixels (g0, g1) and its 4-pixel input window (d0..d3)
//   M0 += g0 (d0-d2), M1 += (g0+g1)(d1+d2), M2 += (g0-g1)(d2-d1), M3 += g1 (d1-d3)
// and dW(dx = 0, 1, 2) = M0 + (M1+M2)/2, (M1-M2)/2, (M1+M2)/2 - M3 once at the end.  A wave owns 16 output channels
// (mt) and both 16-channel input tiles: 3 x 4 x 2 accumulator tiles = 96 VGPRs, alive for the whole kernel.
//
// What changes against the banded kernel is the walk.  There a k-step (4 pairs) meets each of its three input rows
// once and transforms that window three times (once per row tap); with two waves per pixel set that is 12 packed adds
// + ~15 address / bookkeeping instructions per 24 MFMAs -- and on this chip a VALU cycle is a cycle the f32 matrix
// pipe idles (conv_rw.h).  Here the four lane groups of a wave own four pair COLUMNS and walk DOWN: at step t they load
// input row t (window of their pair, two adjacent input channels = their column of both input tiles: 4 x 8 bytes)
// and gradient row t (2 dwords),
// transform the window ONCE (4 packed adds) and multiply it with the gradient rows t (tap 0), t-1 (tap 1) and t-2
// (tap 2), whose transformed values (4 registers per row) stay in registers for their three steps.  10 VALU
// instructions per 24 MFMAs, no LDS, no barrier until the final sum; loads are issued three steps ahead, rows and
// columns outside the image read zeros through the buffer range check.
#pragma once

namespace rw {

struct WgradArgs {
  const float* in;  // [B][Hi][Wi][32]
  const float* g;   // [B][Ho][Wo][32]
  float* partial;   // [grid][kPartialS1]
  int B, Hi, Wi, Ho, Wo;
  Geom gg;          // strips of 4 pair columns over the GRADIENT image (plan4; plan4p when two_d)
  int two_d;        // Winograd in both directions (conv_rw_wgrad2.h)
};

// strips of 4 pair columns (one per lane group)
inline Geom plan4(int Hi, int Wi, int Ho, int Wo) {
  Geom g;
  g.Hi = Hi, g.Wi = Wi, g.Ho = Ho, g.Wo = Wo;
  plan_units(g, Ho, (Wo + 1) / 2, 4);
  return g;
}

// gradient pair -> A operands of the four Winograd positions (g0, g0+g1, g0-g1, g1) and the bias-gradient sum, as ONE
// opaque block: written as plain C++ the compiler's SLP vectoriser pairs these adds with the NEXT step's (packed
// ops over two steps), which makes a step wait for loads issued for the step after it.  `valid` = 1.0f / 0.0f scales the
// pair's contribution to the bias sum (a row outside the piece is loaded but not summed).
__device__ __forceinline__ void g_transform(float (&A)[4], float& bsum, const float g0, const float g1, const float valid) {
  asm("v_mov_b32 %0, %5\n\t"
      "v_add_f32 %1, %5, %6\n\t"
      "v_sub_f32 %2, %5, %6\n\t"
      "v_mov_b32 %3, %6\n\t"
      "v_fmac_f32 %4, %7, %1\n\t"
      "s_nop 1"
      : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]), "=&v"(A[3]), "+v"(bsum)
      : "v"(g0), "v"(g1), "v"(valid));
}

struct WgLoads {
  float d[4][2];  // window pixel c, input tile ct
  float g[2];     // the pair's two gradient pixels
};

template <int NW>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int mt = wave & 1, uslot = wave >> 1;
  constexpr int NU = NW / 2;  // wave pairs: each takes an equal share of the workgroup's steps
  const Geom& G = a.gg;

  f32x4 acc[3][4][2];  // [dy][Winograd position][cin tile]
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[dy][k][ct] = f32x4{0, 0, 0, 0};
  float bsum = 0.f;

  const int cnt = bid < a.B ? (a.B - bid + nblk - 1) / nblk : 0;
  const int nstrips = G.nfull + G.ntr;
  const int T = cnt * G.steps;
  const int lo = (int)((long)uslot * T / NU), hi = (int)((long)(uslot + 1) * T / NU);
  const int in_row = a.Wi * 128, g_row = a.Wo * 128;

  int before = 0;
  for (int si = 0; si < cnt; ++si) {
    const int b = bid + si * nblk;
    for (int k = 0; k < nstrips; ++k) {
      const int n_strip = k < G.nfull ? G.Ho : G.nr;
      const int a0 = lo > before ? lo : before;
      const int a1 = hi < before + n_strip ? hi : before + n_strip;
      const int sb = a0 - before, n = a1 - a0;  // gradient rows [sb, sb + n) of the strip
      before += n_strip;
      if (n <= 0) continue;

      int j, y0;
      bool lane_on;
      if (k < G.nfull) {
        j = 4 * k + kq, y0 = 0, lane_on = true;
      } else {
        const int u = (k - G.nfull) * 4 + kq;
        const int col = u / G.nseg, sg = u - col * G.nseg;
        lane_on = col < G.brem;
        j = 4 * G.nfull + col, y0 = sg * G.nr;
      }
      const int Y = y0 + sb, x0 = 2 * j;
      const __amdgpu_buffer_rsrc_t rin = uniform_rsrc(a.in + (size_t)b * a.Hi * a.Wi * 32, a.Hi * in_row);
      const __amdgpu_buffer_rsrc_t rg = uniform_rsrc(a.g + (size_t)b * a.Ho * a.Wo * 32, a.Ho * g_row);
      // window pixel c of input row Y + t, channels 2 li (tile 0) and 2 li + 1 (tile 1); gradient pixels x0 / x0 + 1 of row Y + t,
      // channel mt*16 + li.  A lane without a column, and the second pixel of an odd row's last pair, point far out
      // of range (zeros); rows advance through the scalar offset, rows past the image are out of range by themselves.
      unsigned vd[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        vd[c] = (lane_on && x0 + c < a.Wi) ? (unsigned)((Y * a.Wi + x0 + c) * 128 + li * 8) : 0x80000000u;
      const unsigned vg0 = lane_on ? (unsigned)((Y * a.Wo + x0) * 128 + (mt * 16 + li) * 4) : 0x80000000u;
      const unsigned vg1 = (lane_on && x0 + 1 < a.Wo) ? vg0 + 128u : 0x80000000u;

      auto issue = [&](WgLoads& L, int t) {
        const unsigned sd = __builtin_amdgcn_readfirstlane((unsigned)(t * in_row));  // (an SGPR: a scalar offset the compiler
        const unsigned sg_ = __builtin_amdgcn_readfirstlane((unsigned)(t * g_row));  //  cannot prove uniform costs a waterfall loop)
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // input channels 2 li and 2 li + 1 (the lane's column of the two cin tiles): 8 bytes
          const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, vd[c], sd, 0));
          L.d[c][0] = v[0], L.d[c][1] = v[1];
        }
        L.g[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, vg0, sg_, 0));
        L.g[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, vg1, sg_, 0));
      };

      WgLoads L0, L1, L2;       // loads of steps t = 0, 1, 2 (mod 3): issued three steps ahead
      float A0[4], A1[4], A2[4];  // transformed gradient rows t (mod 3): g0, g0+g1, g0-g1, g1

      // step t: input row t against gradient rows t (tap 0), t-1 (tap 1), t-2 (tap 2).  Loads are issued
      // unconditionally, three steps ahead (rows past the piece are ordinary rows of the image or out of range: read
      // and dropped), so the number of loads in flight is the same on every path and the wait in front of a step's
      // transform is a counted one (a conditional issue makes the compiler fall back to vmcnt(0): prefetch depth 1).
      auto mma = [&](const int dy, const float (&A)[4], const f32x2 (&v)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) acc[dy][p][ct] = mfma16(A[p], v[p][ct], acc[dy][p][ct]);
      };
      // FULL: all three taps (2 <= t < n), no branch inside
      auto full_step = [&](WgLoads& L, float (&At)[4], const float (&Am1)[4], const float (&Am2)[4], const int t) {
        f32x2 v[4];
        bt_pk(v[0], v[1], v[2], v[3], f32x2{L.d[0][0], L.d[0][1]}, f32x2{L.d[1][0], L.d[1][1]},
              f32x2{L.d[2][0], L.d[2][1]}, f32x2{L.d[3][0], L.d[3][1]});
        g_transform(At, bsum, L.g[0], L.g[1], 1.0f);
        issue(L, t + 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, At, v), mma(1, Am1, v), mma(2, Am2, v);
        __builtin_amdgcn_sched_barrier(0);
      };
      // the first two and last (up to four) steps of a piece: taps whose gradient row lies outside it are skipped
      auto edge_step = [&](WgLoads& L, float (&At)[4], const float (&Am1)[4], const float (&Am2)[4], const int t) {
        const bool do0 = t < n, do1 = t >= 1 && t <= n, do2 = t >= 2;
        f32x2 v[4];
        bt_pk(v[0], v[1], v[2], v[3], f32x2{L.d[0][0], L.d[0][1]}, f32x2{L.d[1][0], L.d[1][1]},
              f32x2{L.d[2][0], L.d[2][1]}, f32x2{L.d[3][0], L.d[3][1]});
        g_transform(At, bsum, L.g[0], L.g[1], do0 ? 1.0f : 0.0f);  // (At is never multiplied when row t is outside the piece)
        issue(L, t + 3);
        __builtin_amdgcn_sched_barrier(0);
        if (do0) mma(0, At, v);
        if (do1) mma(1, Am1, v);
        if (do2) mma(2, Am2, v);
        __builtin_amdgcn_sched_barrier(0);
      };

      issue(L0, 0), issue(L1, 1), issue(L2, 2);
      edge_step(L0, A0, A2, A1, 0);
      edge_step(L1, A1, A0, A2, 1);
      int t = 2;
      for (; t + 2 < n; t += 3) {
        full_step(L2, A2, A1, A0, t);
        full_step(L0, A0, A2, A1, t + 1);
        full_step(L1, A1, A0, A2, t + 2);
      }
      for (;;) {  // (t = 2 mod 3 here; n >= 1, so t <= n + 1)
        edge_step(L2, A2, A1, A0, t);
        if (++t > n + 1) break;
        edge_step(L0, A0, A2, A1, t);
        if (++t > n + 1) break;
        edge_step(L1, A1, A0, A2, t);
        if (++t > n + 1) break;
      }
    }
  }

  // output transform (linear: applied once to the accumulated products), then the cross-wave sum in a fixed order
  // (deterministic) and one slab per workgroup -- the layout wgrad_reduce_multi_kernel reads
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  for (int w = 0; w < NU; ++w) {
    if (uslot == w) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 hs = 0.5f * (acc[dy][1][ct] + acc[dy][2][ct]);
          const f32x4 hd = 0.5f * (acc[dy][1][ct] - acc[dy][2][ct]);
          const f32x4 dw[3] = {acc[dy][0][ct] + hs, hd, hs - acc[dy][3][ct]};
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = mt * 16 + 4 * kq + r, ci = 2 * li + ct;  // (tile ct holds the input channels of parity ct)
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
  for (int i = tid; i < kPartialS1; i += 64 * NW) slab[i] = lds[i];
}

}  // namespace rw
