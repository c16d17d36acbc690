// Stride-1 32->32 3x3 convolution, "row walk" form: forward (bias + ReLU) and data gradient (ReLU mask of the layer
// below), gfx950.  Included by conv.hip.
//
// Reference semantics: encoder.py:59-63,84-87 (Conv2d(32, 32, 3, stride=1) + ReLU) and their autograd.
//
// The f32-input MFMA runs on the SIMD's own FMA lanes (MI355X_MICROARCH.md: 64 FLOP/clk/SIMD = the vector rate), so a
// VALU instruction does not overlap with it -- every VALU cycle in the loop is a cycle the matrix pipe idles.  The
// banded kernel (conv_s1_body) spends 94 VALU instructions per 96 MFMAs: half of them are the Winograd input
// transform, computed for every (row tap, output-channel half) that uses a window -- six times per window -- and
// most of the rest are per-tile coordinates.  This form computes each window's transform ONCE:
//
//   * a wave owns 16 pixel-pair COLUMNS and all 32 output channels and walks DOWN the image: at step t it loads input
//     row t (per lane: the 4-pixel window of its pair, its 4 + 4 channels), transforms it (16 packed adds) and feeds
//     it to the three output rows that use it -- row tap 2 completes output row t-2, tap 1 continues row t-1, tap 0
//     starts row t -- 192 MFMAs per step on three rotating accumulator sets (3 rows x 2 channel halves x 4 Winograd
//     positions x 4 registers = 96 VGPRs);
//   * the operand that changes from MFMA to MFMA is the WEIGHT: the Winograd-transformed filter (4 x-positions
//     instead of 3 taps: 12288 floats = 48 KB) sits in LDS in lane order, one conflict-free ds_read_b128 per four
//     MFMAs, read one group ahead; the transformed window stays in 32 registers for the whole step;
//   * inputs come straight from HBM/L2 into registers (every window is needed by exactly one wave, so there is
//     nothing to share through LDS): 8 buffer loads per step issued a whole step (~6000 cycles) ahead; rows and
//     columns outside the image read zeros through the buffer range check, so the data gradient's zero padding costs
//     nothing and no lane ever branches;
//   * no barrier inside a layer: waves are independent, a workgroup's waves split its samples' steps evenly.
//
// VALU per step (192 MFMAs): 16 packed adds (input transform), ~50 for the output transform / ReLU / stores of the
// completed row, 4 address increments -- against ~190 in the banded form.
//
// Strips.  An image row has PW = ceil(Wo / 2) pairs.  Columns 16k .. 16k+15 form full strip k (Ho steps).  The b =
// PW mod 16 remaining columns are cut into vertical segments of nr rows so that 16 lanes are (column, segment)
// units: a remainder strip takes nr steps and its lanes start at different rows.  (Ho, Wo) = (35, 35): one full
// strip (35 steps) + 2 columns x 7 segments of 5 rows (5 steps) = 40 steps against 39.4 ideal.
#pragma once

namespace rw {

constexpr int kWFloats = 3 * 4 * 2 * 2 * 256;  // transformed filter of one problem in LDS: [dy][pos][q][mt][kq][li][e]

// A raw buffer descriptor over [p, p + bytes) whose words are PROVABLY wave-uniform: built from readfirstlane'd address
// halves.  hipcc wraps every buffer load / store whose descriptor (or scalar offset) it cannot prove uniform in a
// waterfall loop (cdna_hip_programming.md T20) -- ~10 instructions per access and the accesses serialise; the first
// versions of the float first-layer kernels and the weight gradient ran with 20-40 such loops.  All descriptors of the
// row-walk kernels are made here, and their scalar offsets go through readfirstlane.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, int bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi32 << 32) | lo32), (short)0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

struct Geom {
  int Hi, Wi;    // input rows / columns
  int Ho, Wo;    // output rows / columns (forward: Hi-2, Wi-2; data gradient: Hi+2, Wi+2)
  int nfull;     // full strips
  int brem;      // remaining pair columns (0..15)
  int nr;        // rows per segment of the remainder strips
  int nseg;      // segments per remaining column = ceil(Ho / nr)
  int ntr;       // remainder strips = ceil(brem * nseg / 16)
  int steps;     // steps per sample = nfull * Ho + ntr * nr
};

struct Problem {
  const float* in;
  const float* w;    // OIHW [32][32][3][3]
  const float* aux;  // forward: bias[32]; data gradient: the activation below [B][Ho][Wo][32] (ReLU mask)
  float* out;
  int B;
};

constexpr int kMaxLayers = 6;
struct Args {
  int nlayers;  // > 1: layer l+1 of problem p reads layer l's output (p.out of layer l); every workgroup owns its samples
  Geom g[kMaxLayers];
  Problem p[kMaxLayers][2];  // second problem: B = 0 when absent
};

// host: strips of `per` columns over `cols` column units and Ho rows (see "Strips" above): full strips of Ho steps,
// and the remaining columns cut into vertical segments of nr rows so that `per` lane groups are (column, segment) units
inline void plan_units(Geom& g, int Ho, int cols, int per) {
  g.nfull = cols / per, g.brem = cols % per;
  g.nr = g.nseg = g.ntr = 0;
  if (g.brem) {
    // segments of nr rows: brem * ceil(Ho / nr) units on `per` lane groups per strip; cost = strips x nr steps
    long best = -1;
    for (int nr = 1; nr <= Ho; ++nr) {
      const int nseg = (Ho + nr - 1) / nr;
      const int ntr = (g.brem * nseg + per - 1) / per;
      // a step costs the same whatever it computes; short segments also spend 2 of their nr + 2 row loads on halo
      const long cost = (long)ntr * nr * 64 + (long)ntr * 2 * 8;
      if (best < 0 || cost <= best) best = cost, g.nr = nr, g.nseg = nseg, g.ntr = ntr;
    }
  }
  g.steps = g.nfull * Ho + g.ntr * g.nr;
}

// strip plan of the stride-1 forward / data gradient: 16 pixel-pair columns per wave
inline Geom plan(int Hi, int Wi, int Ho, int Wo) {
  Geom g;
  g.Hi = Hi, g.Wi = Wi, g.Ho = Ho, g.Wo = Wo;
  plan_units(g, Ho, (Wo + 1) / 2, 16);
  return g;
}

// Winograd F(2,3) filter transform of one (row tap, cout, cin): x-taps g0 g1 g2 -> the four position weights
__device__ __forceinline__ void filter_transform(float g0, float g1, float g2, float (&u)[4]) {
  u[0] = g0, u[1] = 0.5f * (g0 + g1 + g2), u[2] = 0.5f * (g0 - g1 + g2), u[3] = g2;
}

// OIHW weights -> transformed filter in LDS, laid out so that the A operands of four consecutive MFMAs (cin = 16 q +
// 4 kq + e, e = 0..3, of output channel 16 mt + li) are ONE 16-byte read per lane and a wave's read is 1 KB contiguous.
template <int MODE>
__device__ __forceinline__ void put_filter(float* lds_w, const float (&t)[9], int pr) {
  const int o = pr >> 5, i = pr & 31;
  // forward: cout = o, cin = i, taps as stored.  data gradient: cout = i, cin = o, taps flipped in both directions.
  const int co = MODE == MODE_FWD ? o : i, ci = MODE == MODE_FWD ? i : o;
  const int mt = co >> 4, li = co & 15, q = ci >> 4, kq = (ci >> 2) & 3, e = ci & 3;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    float u[4];
    if (MODE == MODE_FWD)
      filter_transform(t[dy * 3 + 0], t[dy * 3 + 1], t[dy * 3 + 2], u);
    else
      filter_transform(t[(2 - dy) * 3 + 2], t[(2 - dy) * 3 + 1], t[(2 - dy) * 3 + 0], u);
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) lds_w[((((dy * 4 + pos) * 2 + q) * 2 + mt) * 64 + kq * 16 + li) * 4 + e] = u[pos];
  }
}

// (w1 may be null: one problem.  All loads of a pass are issued before the first LDS write: the filters of both
// problems cost one memory latency, not two)
template <int MODE, int NT>
__device__ __forceinline__ void build_filter(float* lds_w, const float* __restrict__ w0, const float* __restrict__ w1,
                                             int tid) {
  // thread <- (o, i) pairs of the OIHW tensor: 9 contiguous floats each, consecutive threads consecutive pairs
  constexpr int NP = 1024 / NT;
  float t0[NP][9], t1[NP][9];
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int pr = tid + u * NT;
#pragma unroll
    for (int k = 0; k < 9; ++k) t0[u][k] = w0[pr * 9 + k];
    if (w1) {
#pragma unroll
      for (int k = 0; k < 9; ++k) t1[u][k] = w1[pr * 9 + k];
    }
  }
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    put_filter<MODE>(lds_w, t0[u], tid + u * NT);
    if (w1) put_filter<MODE>(lds_w + kWFloats, t1[u], tid + u * NT);
  }
}

// B^T d for two channels: v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3 (packed fp32 adds: 4 VALU issues
// for 8 results).  Inline asm, invisible to the compiler's hazard recogniser: a VALU write needs 2 wait states before
// an MFMA reads it (the trailing s_nop); none of the registers written here is ever an MFMA accumulator.
__device__ __forceinline__ void bt_pk(f32x2& v0, f32x2& v1, f32x2& v2, f32x2& v3, const f32x2 d0, const f32x2 d1,
                                      const f32x2 d2, const f32x2 d3) {
  asm("v_pk_add_f32 %0, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %1, %5, %6\n\t"
      "v_pk_add_f32 %2, %6, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %3, %5, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_nop 1"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
      : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
}

struct Acc {
  f32x4 m[2][4];  // [channel half][Winograd position]: 16 output channels x 16 pairs per entry
};

// One layer for the samples this workgroup owns (b = bid, bid + nblk, ...).  NW waves; lds_w holds the transformed
// filters of the layer's (up to two) problems, already built and visible.
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
  // this wave's share of the workgroup's steps: [lo, hi) of the concatenation over (sample, strip)
  const int lo = (int)((long)wave * T / NW), hi = (int)((long)(wave + 1) * T / NW);
  const int in_row = G.Wi * 128, out_row = G.Wo * 128;  // bytes per row

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

      // ---- lane geometry: pair column j, first output row Y of this piece
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
      const int x0 = 2 * j;
      // per-sample descriptors: everything outside the sample's image reads zeros / is not stored
      const __amdgpu_buffer_rsrc_t rin = uniform_rsrc(P.in + (size_t)b * G.Hi * G.Wi * 32, G.Hi * in_row);
      const __amdgpu_buffer_rsrc_t rout = uniform_rsrc(P.out + (size_t)b * G.Ho * G.Wo * 32, G.Ho * out_row);
      const __amdgpu_buffer_rsrc_t raux = uniform_rsrc(MODE == MODE_DGRAD ? P.aux + (size_t)b * G.Ho * G.Wo * 32 : P.aux,
                                                       MODE == MODE_DGRAD ? G.Ho * out_row : 128);
      // window pixel c of input row (Y - pad + t): byte offset inside the sample, or far out of range (a column
      // outside the image; a row outside it is out of range by itself: negative offsets are huge unsigned ones)
      unsigned voff[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = x0 + c - pad;
        const bool ok = lane_on && (unsigned)col < (unsigned)G.Wi;
        voff[c] = ok ? (unsigned)(((Y - pad) * G.Wi + col) * 128 + kq * 16) : 0x80000000u;
      }
      // output pixels (Y + r, x0) and (Y + r, x0 + 1), channels 4 kq .. 4 kq + 3 of each half
      unsigned oa = lane_on ? (unsigned)((Y * G.Wo + x0) * 128 + kq * 16) : 0x80000000u;
      unsigned ob = (lane_on && x0 + 1 < G.Wo) ? oa + 128u : 0x80000000u;

      f32x4 bias[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      if (MODE == MODE_FWD) {
        bias[0] = *reinterpret_cast<const f32x4*>(P.aux + 4 * kq);
        bias[1] = *reinterpret_cast<const f32x4*>(P.aux + 16 + 4 * kq);
      }

      f32x4 raw[4][2];  // [window pixel][channel half]
      auto load_row = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          raw[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, voff[c], 0, 0));
          raw[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, voff[c] + 64u, 0, 0));
        }
      };
      // the weight group (dy, pos, q): both channel halves, one 16-byte read each
      const float* lw_lane = lw + lane * 4;
      auto wread = [&](int g, f32x4& a, f32x4& bb) {
        a = *reinterpret_cast<const f32x4*>(lw_lane + (g * 2 + 0) * 256);
        bb = *reinterpret_cast<const f32x4*>(lw_lane + (g * 2 + 1) * 256);
      };

      Acc S0, S1, S2;
      f32x2 V[4][2][2];  // [position][channel half][component pair]
      f32x4 wa, wb;      // weights of the group about to be multiplied
      f32x4 mk[2][2];    // data gradient: activation below at (pixel a / b, channel half)

      // 64 MFMAs of one row tap into one accumulator set; FIRST: the set starts here (zeros, the bias in position 1:
      // y(x0) = m0 + m1 + m2 and y(x0 + 1) = m1 - m2 - m3 both contain +m1).  The weights of group 0 are in wa / wb on
      // entry; on exit they hold group 0 of row tap `next_dy`.
      auto block = [&](Acc& S, const int dy, const bool first, const int next_dy) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int pos = 0; pos < 4; ++pos) {
            const int g = (dy * 4 + pos) * 2 + q;
            f32x4 na, nb;
            const bool last = q == 1 && pos == 3;
            const int gn = last ? (next_dy * 4 + 0) * 2 + 0 : (pos == 3 ? (dy * 4 + 0) * 2 + 1 : g + 2);
            wread(gn, na, nb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = V[pos][q][e >> 1][e & 1];
              if (first && q == 0 && e == 0) {
                const f32x4 z = {0, 0, 0, 0};
                S.m[0][pos] = mfma16(wa[e], v, pos == 1 ? bias[0] : z);
                S.m[1][pos] = mfma16(wb[e], v, pos == 1 ? bias[1] : z);
              } else {
                S.m[0][pos] = mfma16(wa[e], v, S.m[0][pos]);
                S.m[1][pos] = mfma16(wb[e], v, S.m[1][pos]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            wa = na, wb = nb;
          }
      };
      // a completed output row: A^T m, bias already inside, ReLU / ReLU mask, two pixels x two channel halves
      auto finish = [&](const Acc& S) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4 ya = S.m[mt][0] + S.m[mt][1] + S.m[mt][2];
          f32x4 yb = S.m[mt][1] - S.m[mt][2] - S.m[mt][3];
          if (MODE == MODE_FWD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ya[r] = fmaxf(ya[r], 0.f), yb[r] = fmaxf(yb[r], 0.f);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) ya[r] = mk[0][mt][r] > 0.f ? ya[r] : 0.f, yb[r] = mk[1][mt][r] > 0.f ? yb[r] : 0.f;
          }
          // (store policy: common.h, CURLA_ACT_STORE_POLICY)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, ya), rout,
                                                 oa + mt * 64u, 0, CURLA_ACT_STORE_POLICY);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, yb), rout,
                                                 ob + mt * 64u, 0, CURLA_ACT_STORE_POLICY);
        }
        oa += out_row, ob += out_row;
      };
      // step t: input row t of the piece -> output rows t-2 (completed), t-1, t (started)
      auto step = [&](Acc& Sdy2, Acc& Sdy1, Acc& Sdy0, const int t) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 d0 = {raw[0][q][2 * h], raw[0][q][2 * h + 1]}, d1 = {raw[1][q][2 * h], raw[1][q][2 * h + 1]};
            const f32x2 d2 = {raw[2][q][2 * h], raw[2][q][2 * h + 1]}, d3 = {raw[3][q][2 * h], raw[3][q][2 * h + 1]};
            bt_pk(V[0][q][h], V[1][q][h], V[2][q][h], V[3][q][h], d0, d1, d2, d3);
          }
        if (t < n + 1) {  // next input row: in flight for the whole step
#pragma unroll
          for (int c = 0; c < 4; ++c) voff[c] += in_row;
          load_row();
        }
        const bool do2 = t >= 2, do1 = t >= 1 && t <= n, do0 = t < n;
        if (MODE == MODE_DGRAD && do2) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            mk[0][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(raux, oa + mt * 64u, 0, 0));
            mk[1][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(raux, ob + mt * 64u, 0, 0));
          }
        }
        // (wa / wb hold group 0 of the first block that runs in this step)
        if (do2) block(Sdy2, 2, false, do1 ? 1 : 0);
        if (do1) block(Sdy1, 1, false, do0 ? 0 : 2);
        if (do2) finish(Sdy2);
        if (do0) block(Sdy0, 0, true, t + 1 >= 2 ? 2 : 1);
      };

      load_row();
      wread((0 * 4 + 0) * 2 + 0, wa, wb);  // step 0 runs row tap 0 only
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

}  // namespace rw
