// Stride-1 32->32 3x3 convolution, row walk with 1-D Winograd F(4,3) along x: forward (bias + ReLU) and data gradient
// (ReLU mask of the layer below), gfx950.  Included by conv.hip after conv_rw.h, whose geometry (strips of 16 column
// units cut into (column, segment) lanes), argument structs and buffer descriptors it shares.
//
// Reference semantics: encoder.py:59-63,84-87 (Conv2d(32, 32, 3, stride=1) + ReLU) and their autograd.
//
// Why.  conv_rw.h (F(2,3)) runs the matrix pipe at 0.875 busy: what is left to gain on the stride-1 layers -- 1.5 ms of
// a 2.4 ms update -- is FEWER MFMAs, not better issue.  F(4,3) computes 4 adjacent outputs of a row from a 6-pixel
// window with 6 products per (row tap, cin) instead of F(2,3)'s 2 x 4 = 8 (direct: 12): 0.75 x the MFMAs.  The price:
//   * registers: a lane owns a pixel QUAD and all 32 output channels: 3 rotating accumulator sets x 2 channel halves x 6
//     positions x 4 = 144 VGPRs, the raw window of the next row 48, the transformed window 48 -- more than the 256 a
//     wave has at two waves per SIMD, so this kernel runs ONE wave per SIMD (256-thread workgroups, one per CU; the
//     matrix pipe does not care whose MFMAs it runs, and the F(2,3) loop measured the same at one wave with both
//     channel halves as at two waves with one half each: tools/micro/wino_loop2.hip);
//   * VALU: the input transform is 12 packed FMAs / adds per channel pair (F(2,3): 4 adds), the output transform 10
//     per 4 outputs (F(2,3): 2 x 2) -- ~150 VALU per 288 MFMAs against 64 per 192;
//   * LDS: the transformed filter has 6 x-positions: 72 KB per problem, 144 KB for the two problems of a launch;
//   * a variant with TWO waves per SIMD, each computing one output-channel half of the same pieces (72 accumulator
//     registers in plain VGPRs, no v_accvgpr_read, a second wave to issue from) was built and measured in round 4: it does
//     the input transform and the loads twice and is no faster than F(2,3) (505 against 504 us on configs[1]'s critic-phase
//     stack, 8.42 against 8.81 ms on configs[4]'s) -- not kept;
//   * rounding: the transforms' coefficients (4, 5, 2, 8; 1/4, 1/6, 1/12, 1/24) cost accuracy: rms error of a
//     pre-activation 3.2e-7 against 1.3e-7 for F(2,3) and 1.0e-7 for the direct sum (tools/micro/wino_error.py) --
//     2.5 x as many activations within rounding of zero land on the other side of a ReLU (tests/test_gpu_fullsize.py
//     counts them against float64).
//
// Transforms (Lavin & Gray 2016, points 0, +-1, +-2, inf):
//   filter   u = G g:   u0 = g0/4, u1 = -(g0+g1+g2)/6, u2 = -(g0-g1+g2)/6, u3 = g0/24+g1/12+g2/6, u4 = g0/24-g1/12+g2/6, u5 = g2
//   input    v = B^T d: v0 = 4 d0 - 5 d2 + d4,  v1 = (d4 - 4 d2) + (d3 - 4 d1),  v2 = (d4 - 4 d2) - (d3 - 4 d1),
//                       v3 = (d4 - d2) + 2 (d3 - d1),  v4 = (d4 - d2) - 2 (d3 - d1),  v5 = 4 d1 - 5 d3 + d5
//   output   y = A^T m: y0 = m0+m1+m2+m3+m4, y1 = (m1-m2) + 2 (m3-m4), y2 = (m1+m2) + 4 (m3+m4), y3 = (m1-m2) + 8 (m3-m4) + m5
// (every output contains +m1: the bias rides in position 1's initial accumulator, as in conv_rw.h)
#pragma once

namespace rw43 {

using rw::Args;
using rw::Geom;
using rw::Problem;
using rw::uniform_rsrc;

constexpr int kWFloats = 3 * 6 * 2 * 2 * 256;  // transformed filter of one problem in LDS: [dy][pos][q][mt][kq][li][e]

// strip plan: 16 pixel-QUAD columns per wave
inline Geom plan(int Hi, int Wi, int Ho, int Wo) {
  Geom g;
  g.Hi = Hi, g.Wi = Wi, g.Ho = Ho, g.Wo = Wo;
  rw::plan_units(g, Ho, (Wo + 3) / 4, 16);
  return g;
}

// (explicit fmaf / no contraction: the same weights must give the same bits whichever code path transforms them -- the
// first or the second problem of a launch -- or a sample's result would depend on the launch it is computed in)
__device__ __forceinline__ void filter_transform(float g0, float g1, float g2, float (&u)[6]) {
#pragma clang fp contract(off)
  const float s = g0 + g2;
  u[0] = 0.25f * g0;
  u[1] = -(s + g1) * (1.f / 6.f);
  u[2] = -(s - g1) * (1.f / 6.f);
  const float a = __builtin_fmaf(g0, 1.f / 24.f, g2 * (1.f / 6.f)), b = g1 * (1.f / 12.f);
  u[3] = a + b, u[4] = a - b, u[5] = g2;
}

template <int MODE>
__device__ __forceinline__ void put_filter(float* lds_w, const float (&t)[9], int pr) {
  const int o = pr >> 5, i = pr & 31;
  // forward: cout = o, cin = i, taps as stored.  data gradient: cout = i, cin = o, taps flipped in both directions.
  const int co = MODE == MODE_FWD ? o : i, ci = MODE == MODE_FWD ? i : o;
  const int mt = co >> 4, li = co & 15, q = ci >> 4, kq = (ci >> 2) & 3, e = ci & 3;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    float u[6];
    if (MODE == MODE_FWD)
      filter_transform(t[dy * 3 + 0], t[dy * 3 + 1], t[dy * 3 + 2], u);
    else
      filter_transform(t[(2 - dy) * 3 + 2], t[(2 - dy) * 3 + 1], t[(2 - dy) * 3 + 0], u);
#pragma unroll
    for (int pos = 0; pos < 6; ++pos) lds_w[((((dy * 6 + pos) * 2 + q) * 2 + mt) * 64 + kq * 16 + li) * 4 + e] = u[pos];
  }
}

// (w1 may be null: one problem.  All loads of a pass are issued before the first LDS write.)
template <int MODE, int NT>
__device__ __forceinline__ void build_filter(float* lds_w, const float* __restrict__ w0, const float* __restrict__ w1,
                                             int tid) {
  constexpr int NP = 1024 / NT;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {  // (two passes of NP / 2 pairs: 72 staging registers instead of 144)
    float t0[NP / 2][9], t1[NP / 2][9];
#pragma unroll
    for (int u = 0; u < NP / 2; ++u) {
      const int pr = tid + (half * (NP / 2) + u) * NT;
#pragma unroll
      for (int k = 0; k < 9; ++k) t0[u][k] = w0[pr * 9 + k];
      if (w1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) t1[u][k] = w1[pr * 9 + k];
      }
    }
#pragma unroll
    for (int u = 0; u < NP / 2; ++u) {
      const int pr = tid + (half * (NP / 2) + u) * NT;
      put_filter<MODE>(lds_w, t0[u], pr);
      if (w1) put_filter<MODE>(lds_w + kWFloats, t1[u], pr);
    }
  }
}

// B^T d for two channels (see the header): 12 packed VALU issues for 12 results.  Inline asm, invisible to the compiler's
// hazard recogniser: a VALU write needs 2 wait states before an MFMA reads it (the trailing s_nop); none of the registers
// written here is ever an MFMA accumulator.  c4 / c2 / c5 hold the constants (4, 4), (2, 2), (5, 5) in SGPR pairs (one
// scalar operand per instruction: within the constant-bus limit).
__device__ __forceinline__ void bt_pk(f32x2& v0, f32x2& v1, f32x2& v2, f32x2& v3, f32x2& v4, f32x2& v5, const f32x2 d0,
                                      const f32x2 d1, const f32x2 d2, const f32x2 d3, const f32x2 d4, const f32x2 d5,
                                      const f32x2 c4, const f32x2 c2, const f32x2 c5) {
  f32x2 a, b, c, e;
  asm("v_pk_fma_f32 %6, %12, %16, %14 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // a = d4 - 4 d2
      "v_pk_fma_f32 %7, %11, %16, %13 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // b = d3 - 4 d1
      "v_pk_add_f32 %8, %14, %12 neg_lo:[0,1] neg_hi:[0,1]\n\t"            // c = d4 - d2
      "v_pk_add_f32 %9, %13, %11 neg_lo:[0,1] neg_hi:[0,1]\n\t"            // e = d3 - d1
      "v_pk_fma_f32 %0, %12, %18, %14 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // v0 = d4 - 5 d2
      "v_pk_fma_f32 %5, %13, %18, %15 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // v5 = d5 - 5 d3
      "v_pk_add_f32 %1, %6, %7\n\t"                                        // v1 = a + b
      "v_pk_add_f32 %2, %6, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"              // v2 = a - b
      "v_pk_fma_f32 %3, %9, %17, %8\n\t"                                   // v3 = c + 2 e
      "v_pk_fma_f32 %4, %9, %17, %8 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"     // v4 = c - 2 e
      "v_pk_fma_f32 %0, %10, %16, %0\n\t"                                  // v0 += 4 d0
      "v_pk_fma_f32 %5, %11, %16, %5\n\t"                                  // v5 += 4 d1
      "s_nop 1"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(e)
      : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "s"(c4), "s"(c2), "s"(c5));
}

struct Acc {
  f32x4 m[2][6];  // [channel half][Winograd position]: 16 output channels x 16 quads per entry
};

// One layer for the samples this workgroup owns (b = bid, bid + nblk, ...).  NW waves (one per SIMD); lds_w holds the
// transformed filters of the layer's (up to two) problems, already built and visible.
template <int MODE, int NW>
__device__ __forceinline__ void run_layer(const Geom& G, const Problem& P0, const Problem& P1, const float* lds_w,
                                          int bid, int nblk) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int pad = MODE == MODE_FWD ? 0 : 2;
  const int cnt0 = bid < P0.B ? (P0.B - bid + nblk - 1) / nblk : 0;
  const int cnt1 = bid < P1.B ? (P1.B - bid + nblk - 1) / nblk : 0;
  const int nstrips = G.nfull + G.ntr;
  const int T = (cnt0 + cnt1) * G.steps;
  const int lo = (int)((long)wave * T / NW), hi = (int)((long)(wave + 1) * T / NW);
  const int in_row = G.Wi * 128, out_row = G.Wo * 128;  // bytes per row
  const f32x2 c4 = {4.f, 4.f}, c2 = {2.f, 2.f}, c5 = {5.f, 5.f};

  int before = 0;  // steps of the instances before the current one
  for (int si = 0; si < cnt0 + cnt1; ++si) {
    const bool second = si >= cnt0;
    const Problem& P = second ? P1 : P0;
    const int b = bid + (second ? si - cnt0 : si) * nblk;
    const float* lw = lds_w + (second ? kWFloats : 0);
    for (int k = 0; k < nstrips; ++k) {
      const int n_strip = k < G.nfull ? G.Ho : G.nr;
      const int a0 = lo > before ? lo : before;
      const int a1 = hi < before + n_strip ? hi : before + n_strip;
      const int sb = a0 - before, n = a1 - a0;  // this wave runs steps [sb, sb + n) of the strip
      before += n_strip;
      if (n <= 0) continue;

      // ---- lane geometry: quad column j, first output row Y of this piece
      int j, y0;
      bool lane_on;
      if (k < G.nfull) {
        j = 16 * k + li, y0 = 0, lane_on = true;
      } else {
        const int u = (k - G.nfull) * 16 + li;
        const int col = u / G.nseg, sg = u - col * G.nseg;
        lane_on = col < G.brem;
        j = 16 * G.nfull + col, y0 = sg * G.nr;
      }
      const int Y = y0 + sb;
      const int x0 = 4 * j;
      const __amdgpu_buffer_rsrc_t rin = uniform_rsrc(P.in + (size_t)b * G.Hi * G.Wi * 32, G.Hi * in_row);
      const __amdgpu_buffer_rsrc_t rout = uniform_rsrc(P.out + (size_t)b * G.Ho * G.Wo * 32, G.Ho * out_row);
      const __amdgpu_buffer_rsrc_t raux = uniform_rsrc(MODE == MODE_DGRAD ? P.aux + (size_t)b * G.Ho * G.Wo * 32 : P.aux,
                                                       MODE == MODE_DGRAD ? G.Ho * out_row : 128);
      // window pixel c of input row (Y - pad + t): byte offset inside the sample, or far out of range (a column
      // outside the image; a row outside it is out of range by itself: negative offsets are huge unsigned ones)
      unsigned voff[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int col = x0 + c - pad;
        const bool ok = lane_on && (unsigned)col < (unsigned)G.Wi;
        voff[c] = ok ? (unsigned)(((Y - pad) * G.Wi + col) * 128 + kq * 16) : 0x80000000u;
      }
      // output pixels (Y + r, x0 + p), p = 0..3, channels 4 kq .. 4 kq + 3 of each half
      unsigned oo[4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
        oo[p] = (lane_on && x0 + p < G.Wo) ? (unsigned)((Y * G.Wo + x0 + p) * 128 + kq * 16) : 0x80000000u;

      f32x4 bias[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      if (MODE == MODE_FWD) {
        bias[0] = *reinterpret_cast<const f32x4*>(P.aux + 4 * kq);
        bias[1] = *reinterpret_cast<const f32x4*>(P.aux + 16 + 4 * kq);
      }

      f32x4 raw[6][2];  // [window pixel][channel half]
      auto load_row = [&]() {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          raw[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, voff[c], 0, 0));
          raw[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, voff[c] + 64u, 0, 0));
        }
      };
      // Weights.  An ITERATION is (row tap dy, channel half q, position pair pp): four 16-byte reads per lane (two
      // positions x two output-channel halves) feeding 16 MFMAs.  With one wave per SIMD nothing else covers the LDS
      // latency, so the reads run TWO iterations (32 MFMAs = 1024 cycles) ahead in a rotating set of three buffers; a row
      // tap has six iterations, so every block starts at the same phase of the rotation.
      const float* lw_lane = lw + lane * 4;
      f32x4 Wt[3][4];
      auto wread4 = [&](const int dy, const int it, f32x4 (&w)[4]) {
        const int q = it / 3, pos = 2 * (it % 3);
        const int g0 = (dy * 6 + pos) * 2 + q, g1 = (dy * 6 + pos + 1) * 2 + q;
        w[0] = *reinterpret_cast<const f32x4*>(lw_lane + (g0 * 2 + 0) * 256);
        w[1] = *reinterpret_cast<const f32x4*>(lw_lane + (g0 * 2 + 1) * 256);
        w[2] = *reinterpret_cast<const f32x4*>(lw_lane + (g1 * 2 + 0) * 256);
        w[3] = *reinterpret_cast<const f32x4*>(lw_lane + (g1 * 2 + 1) * 256);
      };

      Acc S0, S1, S2;
      f32x2 V[6][2][2];  // [position][channel half][component pair]
      f32x4 mk[4][2];    // data gradient: activation below at (pixel p, channel half)

      // 96 MFMAs of one row tap into one accumulator set; FIRST: the set starts here (zeros, the bias in position 1).
      // On entry Wt[0] / Wt[1] hold iterations 0 / 1 of this row tap; on exit those of row tap `next_dy`.  Two positions
      // at a time: four independent accumulator chains (2 channel halves x 2 positions) in rotation.
      auto block = [&](Acc& S, const int dy, const bool first, const int next_dy) {
#pragma unroll
        for (int it = 0; it < 6; ++it) {
          const int q = it / 3, pos = 2 * (it % 3);
          if (it + 2 < 6)
            wread4(dy, it + 2, Wt[(it + 2) % 3]);
          else
            wread4(next_dy, it + 2 - 6, Wt[(it + 2) % 3]);
          __builtin_amdgcn_sched_barrier(0);
          const f32x4(&w)[4] = Wt[it % 3];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = V[pos][q][e >> 1][e & 1], v1 = V[pos + 1][q][e >> 1][e & 1];
            if (first && q == 0 && e == 0) {
              const f32x4 z = {0, 0, 0, 0};
              S.m[0][pos] = mfma16(w[0][e], v, z);
              S.m[1][pos] = mfma16(w[1][e], v, z);
              S.m[0][pos + 1] = mfma16(w[2][e], v1, pos == 0 ? bias[0] : z);
              S.m[1][pos + 1] = mfma16(w[3][e], v1, pos == 0 ? bias[1] : z);
            } else {
              S.m[0][pos] = mfma16(w[0][e], v, S.m[0][pos]);
              S.m[1][pos] = mfma16(w[1][e], v, S.m[1][pos]);
              S.m[0][pos + 1] = mfma16(w[2][e], v1, S.m[0][pos + 1]);
              S.m[1][pos + 1] = mfma16(w[3][e], v1, S.m[1][pos + 1]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // a completed output row: A^T m, bias already inside, ReLU / ReLU mask, four pixels x two channel halves
      auto finish = [&](const Acc& S) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const f32x4 s12 = S.m[mt][1] + S.m[mt][2], d12 = S.m[mt][1] - S.m[mt][2];
          const f32x4 s34 = S.m[mt][3] + S.m[mt][4], d34 = S.m[mt][3] - S.m[mt][4];
          f32x4 y[4];
          y[0] = S.m[mt][0] + s12 + s34;
          y[1] = d12 + 2.f * d34;
          y[2] = s12 + 4.f * s34;
          y[3] = d12 + 8.f * d34 + S.m[mt][5];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            if (MODE == MODE_FWD) {
#pragma unroll
              for (int r = 0; r < 4; ++r) y[p][r] = fmaxf(y[p][r], 0.f);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) y[p][r] = mk[p][mt][r] > 0.f ? y[p][r] : 0.f;
            }
            // (store policy: common.h, CURLA_ACT_STORE_POLICY)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, y[p]),
                                                   rout, oo[p] + mt * 64u, 0, CURLA_ACT_STORE_POLICY);
          }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) oo[p] += out_row;
      };
      // step t: input row t of the piece -> output rows t-2 (completed), t-1, t (started)
      auto step = [&](Acc& Sdy2, Acc& Sdy1, Acc& Sdy0, const int t) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x2 d[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) d[c] = f32x2{raw[c][q][2 * h], raw[c][q][2 * h + 1]};
            bt_pk(V[0][q][h], V[1][q][h], V[2][q][h], V[3][q][h], V[4][q][h], V[5][q][h], d[0], d[1], d[2], d[3], d[4], d[5],
                  c4, c2, c5);
          }
        if (t < n + 1) {  // next input row: in flight for the whole step
#pragma unroll
          for (int c = 0; c < 6; ++c) voff[c] += in_row;
          load_row();
        }
        const bool do2 = t >= 2, do1 = t >= 1 && t <= n, do0 = t < n;
        if (MODE == MODE_DGRAD && do2) {
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              mk[p][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(raux, oo[p] + mt * 64u, 0, 0));
        }
        // (Wt[0] / Wt[1] hold iterations 0 / 1 of the first block that runs in this step)
        if (do2) block(Sdy2, 2, false, do1 ? 1 : 0);
        if (do1) block(Sdy1, 1, false, do0 ? 0 : 2);
        if (do2) finish(Sdy2);
        if (do0) block(Sdy0, 0, true, t + 1 >= 2 ? 2 : 1);
      };

      load_row();
      wread4(0, 0, Wt[0]), wread4(0, 1, Wt[1]);  // step 0 runs row tap 0 only
      for (int t = 0;;) {
        step(S1, S2, S0, t);
        if (++t > n + 1) break;
        step(S2, S0, S1, t);
        if (++t > n + 1) break;
        step(S0, S1, S2, t);
        if (++t > n + 1) break;
      }
    }
  }
}

}  // namespace rw43
