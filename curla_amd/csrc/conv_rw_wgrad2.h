// Weight gradient of the stride-1 32->32 3x3 convolution with Winograd F(3,2) in BOTH directions, gfx950.  Included by
// conv.hip after conv_rw_wgrad.h (whose argument block, lane roles, slab layout and final cross-wave sum it shares).
//
//   dW[co][ci][dy][dx] = sum over (sample, y, x) of g[y][x][co] * in[y + dy][x + dx][ci]        (encoder.py:59-63, autograd)
//
// conv_rw_wgrad.h transforms along x only: a gradient pixel PAIR and its 4-pixel input window meet in 4 products per row
// tap instead of 6 -- 12 matrix instructions per pair and tile over the three row taps.  The same identity holds along
// y: a 2 x 2 block of gradient pixels (rows R0, R1) and its 4 x 4 input window (rows E0 .. E3) meet in
//   M[yp][xp] += Gy[yp][xp] * Vy[yp][xp],   Gy = (R0, R0 + R1, R0 - R1, R1),  Vy = (E0 - E2, E1 + E2, E2 - E1, E1 - E3)
// of the x-transformed rows (R = (g0, g0 + g1, g0 - g1, g1), E = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)): 16 products for 36
// multiply-adds, 8 matrix instructions per pair and tile instead of 12.  The weight gradient shares its CUs with the
// bf16x3 data gradient (bwd_rwb2_kernel) and the chip holds ~1.5 GHz under that pair: the launch is matrix-pipe-bound,
// and a third fewer f32 matrix instructions is what shortens it.
//
// Walk: a step is a gradient ROW PAIR of the wave's four pair columns (one per lane group, as conv_rw_wgrad.h).  It loads
// input rows 2s + 2, 2s + 3 (rows 2s, 2s + 1 are the previous step's, kept x-transformed in registers) and gradient rows
// 2s, 2s + 1, forms the 16 operand pairs (~40 VALU instructions) and issues 32 matrix instructions on 4 x 4 x 2
// accumulator tiles (128 registers).  Every step is complete in itself: no taps that reach into a neighbouring step, no
// edge steps; a piece costs two halo rows.  Rows and columns outside the image read zeros through the buffer range
// check, which is also what the odd last row of a pair is.  The final output transform applies A^T . A in both directions.
#pragma once

namespace rw {

// strips of 4 pair columns over the gradient image in units of ROW PAIRS (Geom::Ho = pair rows)
inline Geom plan4p(int Hi, int Wi, int Ho, int Wo) {
  Geom g;
  g.Hi = Hi, g.Wi = Wi, g.Wo = Wo;
  g.Ho = (Ho + 1) / 2;
  plan_units(g, g.Ho, (Wo + 1) / 2, 4);
  return g;
}

struct Wg2Loads {
  f32x2 d[2][4];  // input rows 2s + 2, 2s + 3: window pixel c, (cin tile 0, cin tile 1)
  f32x2 g[2];     // gradient rows 2s, 2s + 1: the pair's two pixels
};

// (g0, g1) -> (g0 + g1, g0 - g1) in one packed instruction (both results read the first operand's low and the second's
// high half; the second is negated for the high result).  Inline asm: the trailing s_nop is the two wait states a VALU
// result needs before a matrix instruction reads it (conv_rw.h, bt_pk).
__device__ __forceinline__ f32x2 sum_diff(const f32x2 g) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 1"
      : "=&v"(r)
      : "v"(g));
  return r;
}

// the y-direction combinations of two x positions' rows E0 .. E3 (each a packed pair over the two cin tiles):
// (E0 - E2, E1 + E2, E2 - E1, E1 - E3).  One opaque block: written as vector arithmetic the compiler splits the packed
// operations into scalar ones (24 instead of 12 per step)
__device__ __forceinline__ void vy_pk(f32x2 (&Va)[4], f32x2 (&Vb)[4], const f32x2 a0, const f32x2 a1, const f32x2 a2,
                                      const f32x2 a3, const f32x2 b0, const f32x2 b1, const f32x2 b2, const f32x2 b3) {
  asm("v_pk_add_f32 %0, %8, %10 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %1, %9, %10\n\t"
      "v_pk_add_f32 %2, %10, %9 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %3, %9, %11 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %4, %12, %14 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %5, %13, %14\n\t"
      "v_pk_add_f32 %6, %14, %13 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %7, %13, %15 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 1"
      : "=&v"(Va[0]), "=&v"(Va[1]), "=&v"(Va[2]), "=&v"(Va[3]), "=&v"(Vb[0]), "=&v"(Vb[1]), "=&v"(Vb[2]), "=&v"(Vb[3])
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}

template <int NW>
__device__ __forceinline__ void wgrad2_body(const WgradArgs& a, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int mt = wave & 1, uslot = wave >> 1;
  constexpr int NU = NW / 2;  // wave pairs: each takes an equal share of the workgroup's steps
  const Geom& G = a.gg;       // (plan4p: rows are row pairs)

  f32x4 acc[4][4][2];  // [y position][x position][cin tile]
#pragma unroll
  for (int yp = 0; yp < 4; ++yp)
#pragma unroll
    for (int xp = 0; xp < 4; ++xp)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[yp][xp][ct] = f32x4{0, 0, 0, 0};
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
      const int sb = a0 - before, n = a1 - a0;  // row pairs [sb, sb + n) of the strip
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
      const int Y = 2 * (y0 + sb), x0 = 2 * j;  // first gradient (= first input) row of the piece
      const __amdgpu_buffer_rsrc_t rin = uniform_rsrc(a.in + (size_t)b * a.Hi * a.Wi * 32, a.Hi * in_row);
      const __amdgpu_buffer_rsrc_t rg = uniform_rsrc(a.g + (size_t)b * a.Ho * a.Wo * 32, a.Ho * g_row);
      // (as conv_rw_wgrad.h: window pixel c of input row Y + t, channels 2 li / 2 li + 1; gradient pixels x0 / x0 + 1, channel
      // 16 mt + li; lanes without a column and pixels past a row's end point far out of range; rows advance through the
      // scalar offset and are out of range by themselves past the image)
      unsigned vd[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        vd[c] = (lane_on && x0 + c < a.Wi) ? (unsigned)((Y * a.Wi + x0 + c) * 128 + li * 8) : 0x80000000u;
      const unsigned vg0 = lane_on ? (unsigned)((Y * a.Wo + x0) * 128 + (mt * 16 + li) * 4) : 0x80000000u;
      const unsigned vg1 = (lane_on && x0 + 1 < a.Wo) ? vg0 + 128u : 0x80000000u;

      auto load_in_row = [&](f32x2 (&d)[4], int row) {
        const unsigned sd = __builtin_amdgcn_readfirstlane((unsigned)(row * in_row));
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, vd[c], sd, 0));
      };
      auto issue = [&](Wg2Loads& L, int s) {  // step s: input rows 2s + 2, 2s + 3, gradient rows 2s, 2s + 1
        load_in_row(L.d[0], 2 * s + 2);
        load_in_row(L.d[1], 2 * s + 3);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const unsigned sg_ = __builtin_amdgcn_readfirstlane((unsigned)((2 * s + r) * g_row));
          L.g[r][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, vg0, sg_, 0));
          L.g[r][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, vg1, sg_, 0));  // (two halves of a register pair)
        }
      };

      // x-transformed input rows 2s, 2s + 1 of the coming step (P) and 2s + 2, 2s + 3 (Q): the two sets swap roles from
      // step to step
      f32x2 P0[4], P1[4], Q0[4], Q1[4], h0[4], h1[4];
      load_in_row(h0, 0);
      load_in_row(h1, 1);
      Wg2Loads L0, L1;  // loads of even / odd steps, issued two steps ahead (unconditionally: rows past the piece are
                        // ordinary rows of the image or out of range -- read and dropped; conv_rw_wgrad.h)
      auto step = [&](Wg2Loads& L, const f32x2 (&E0)[4], const f32x2 (&E1)[4], f32x2 (&E2)[4], f32x2 (&E3)[4], const int s) {
        bt_pk(E2[0], E2[1], E2[2], E2[3], L.d[0][0], L.d[0][1], L.d[0][2], L.d[0][3]);
        bt_pk(E3[0], E3[1], E3[2], E3[3], L.d[1][0], L.d[1][1], L.d[1][2], L.d[1][3]);
        // the 2 x 2 gradient block -> its 16 operands: rows R = (g0, g0 + g1, g0 - g1, g1), then (R0, R0 + R1, R0 - R1, R1):
        // six packed instructions (and the bias gradient's sum of the four pixels falls out of them)
        const f32x2 G0 = L.g[0], G1 = L.g[1];
        const f32x2 S0 = sum_diff(G0), S1 = sum_diff(G1);
        const f32x2 H1 = G0 + G1, H2 = G0 - G1, F1 = S0 + S1, F2 = S0 - S1;
        bsum += F1[0];
        const float A[4][4] = {{G0[0], S0[0], S0[1], G0[1]},
                               {H1[0], F1[0], F1[1], H1[1]},
                               {H2[0], F2[0], F2[1], H2[1]},
                               {G1[0], S1[0], S1[1], G1[1]}};
        issue(L, s + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xq = 0; xq < 4; xq += 2) {
          f32x2 Va[4], Vb[4];
          vy_pk(Va, Vb, E0[xq], E1[xq], E2[xq], E3[xq], E0[xq + 1], E1[xq + 1], E2[xq + 1], E3[xq + 1]);
#pragma unroll
          for (int yp = 0; yp < 4; ++yp)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
              acc[yp][xq][ct] = mfma16(A[yp][xq], Va[yp][ct], acc[yp][xq][ct]);
              acc[yp][xq + 1][ct] = mfma16(A[yp][xq + 1], Vb[yp][ct], acc[yp][xq + 1][ct]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      };

      issue(L0, 0), issue(L1, 1);
      bt_pk(P0[0], P0[1], P0[2], P0[3], h0[0], h0[1], h0[2], h0[3]);
      bt_pk(P1[0], P1[1], P1[2], P1[3], h1[0], h1[1], h1[2], h1[3]);
      for (int s = 0;;) {
        step(L0, P0, P1, Q0, Q1, s);
        if (++s >= n) break;
        step(L1, Q0, Q1, P0, P1, s);
        if (++s >= n) break;
      }
    }
  }

  // output transform in both directions (linear: applied once to the accumulated products), then the cross-wave sum in
  // a fixed order (deterministic) and one slab per workgroup -- the layout wgrad_reduce_multi_kernel reads
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  for (int w = 0; w < NU; ++w) {
    if (uslot == w) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x4 tx[4][3];  // [y position][dx]
#pragma unroll
        for (int yp = 0; yp < 4; ++yp) {
          const f32x4 hs = 0.5f * (acc[yp][1][ct] + acc[yp][2][ct]);
          const f32x4 hd = 0.5f * (acc[yp][1][ct] - acc[yp][2][ct]);
          tx[yp][0] = acc[yp][0][ct] + hs, tx[yp][1] = hd, tx[yp][2] = hs - acc[yp][3][ct];
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const f32x4 hs = 0.5f * (tx[1][dx] + tx[2][dx]);
          const f32x4 hd = 0.5f * (tx[1][dx] - tx[2][dx]);
          const f32x4 dw[3] = {tx[0][dx] + hs, hd, hs - tx[3][dx]};
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = mt * 16 + 4 * kq + r, ci = 2 * li + ct;  // (tile ct holds the input channels of parity ct)
              float* d = lds + co * 288 + ci * 9 + dy * 3 + dx;
              *d = (w == 0) ? dw[dy][r] : *d + dw[dy][r];
            }
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
