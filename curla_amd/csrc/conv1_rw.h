// First conv layer (3x3, stride 2, C -> 32, + bias + ReLU) from a FLOAT NHWC minibatch, "row walk" form, gfx950.
// Included by conv.hip.  Reference semantics: encoder.py:54-57,78-81 (obs / 255, Conv2d(C, 32, 3, stride=2), ReLU) on the
// float observations the torch/kornia augmentations produce (utils.py:168-182) -- BASELINE configs[4]: 168x168x12.
//
// The banded kernel (conv1_fwd_kernel) stages input rows in LDS; an input row of 168 x 12 floats is 8 KB, so a band is
// 4 output rows and staging + barriers are a third of the kernel.  Here nothing is staged: the 3 x C operand values of
// an output pixel and one input row are 3 x C CONTIGUOUS floats of the NHWC tensor (pixels 2x .. 2x+2), so the four
// lane groups of a wave take a quarter each (E = ceil(3C/4) values, 16-byte loads at 4-byte alignment).  A wave owns 16
// output columns and all 32 output channels and walks down: output row y needs input rows 2y, 2y+1, 2y+2, the last
// of which is the next row's first -- two new input rows per step, loaded a whole step ahead into a rotating set of
// five row buffers.  Weights (3 x E x 2 registers per lane, 1/255 folded in) never change.  Per step: 6 E MFMAs, two
// stores, no conversion, no transform, no LDS, no barrier.
#pragma once

namespace rw {

// ReLU as one integer max on the bit pattern (positive floats order like their bits, everything with the sign bit set
// is a negative integer): fmaxf(x, 0) costs two instructions on an MFMA result, which is not known to be canonical --
// IEEE mode makes the compiler quiet it first (v_max_f32 x, x, x)
__device__ __forceinline__ float relu_bits(float x) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

struct Conv1Args {
  const float* src;   // [B][Hc][Wc][C] float NHWC in [0, 255]
  const float* w;     // OIHW [32][C][3][3]
  const float* bias;  // [32]
  float* out;         // [B][Ho][Wo][32]
  int B, Hc, Wc, Ho, Wo;
  float scale;
  Geom g;  // strips of 16 OUTPUT COLUMNS (plan_units over Wo)
};

template <int C, int NW>
__device__ __forceinline__ void conv1_body(const Conv1Args& a, const int bid, const int nblk) {
  constexpr int E = (3 * C + 3) / 4;  // operand values per lane group and input row
  constexpr int NL = (E + 3) / 4;     // 16-byte loads per lane and input row (the last one may over-read: dropped)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const Geom& G = a.g;

  // weights: lane (li = cout, kq) holds W[cout][dy][rr = E kq + e] * scale, (dx, c) = (rr / C, rr % C); zero past 3C
  float wr[3][E][2];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int rr = E * kq + e;
      const bool ok = rr < 3 * C;
      const int dx = ok ? rr / C : 0, c = ok ? rr - dx * C : 0;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) wr[dy][e][mt] = ok ? a.w[((mt * 16 + li) * C + c) * 9 + dy * 3 + dx] * a.scale : 0.f;
    }
  f32x4 bias4[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) bias4[mt] = *reinterpret_cast<const f32x4*>(a.bias + mt * 16 + 4 * kq);

  const int cnt = bid < a.B ? (a.B - bid + nblk - 1) / nblk : 0;
  const int nstrips = G.nfull + G.ntr;
  const int T = cnt * G.steps;
  const int lo = (int)((long)wave * T / NW), hi = (int)((long)(wave + 1) * T / NW);
  const int in_row = a.Wc * C * 4, out_row = a.Wo * 128;  // bytes per row

  int before = 0;
  for (int si = 0; si < cnt; ++si) {
    const int b = bid + si * nblk;
    for (int k = 0; k < nstrips; ++k) {
      const int n_strip = k < G.nfull ? G.Ho : G.nr;
      const int a0 = lo > before ? lo : before;
      const int a1 = hi < before + n_strip ? hi : before + n_strip;
      const int sb = a0 - before, n = a1 - a0;  // this wave runs output rows [sb, sb + n) of the strip
      before += n_strip;
      if (n <= 0) continue;

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
      const int Y = y0 + sb;
      const __amdgpu_buffer_rsrc_t rin = uniform_rsrc(a.src + (size_t)b * a.Hc * a.Wc * C, a.Hc * in_row);
      const __amdgpu_buffer_rsrc_t rout = uniform_rsrc(a.out + (size_t)b * a.Ho * a.Wo * 32, a.Ho * out_row);
      // the lane group's E values of input row 2 Y + r: ((2 Y + r) Wc + 2x) C + E kq floats into the sample; the lane's own
      // first row is part of its offset (lanes of a remainder strip start at different rows), r advances through the
      // (wave-uniform) scalar offset; a lane without a column and rows past the image are out of range (zeros, nothing stored)
      const unsigned vin = lane_on ? (unsigned)(((2 * Y * a.Wc + 2 * x) * C + E * kq) * 4) : 0x80000000u;
      unsigned vo = lane_on ? (unsigned)((Y * a.Wo + x) * 128 + kq * 16) : 0x80000000u;

      struct Row {
        f32x4 v[NL];
      };
      auto load_row = [&](Row& R, int r) {  // input row 2 Y + r of the sample
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(r * in_row));
#pragma unroll
        for (int u = 0; u < NL; ++u)
          R.v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, vin + 16u * u, so, 0));
      };
      // output row t of the piece from the three input rows r0 (= 2 (Y + t)), r1, r2; the next step's two new rows
      // (2 (Y + t) + 3, + 4: scalar offsets 2 t + 3, 2 t + 4) are requested first and land during this step's MFMAs
      auto step = [&](const Row& r0, const Row& r1, const Row& r2, Row& p0, Row& p1, const int t) {
        load_row(p0, 2 * t + 3);
        load_row(p1, 2 * t + 4);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const Row& R = dy == 0 ? r0 : dy == 1 ? r1 : r2;
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const float v = R.v[e >> 2][e & 3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              acc[mt] = mfma16(wr[dy][e][mt], v, (dy == 0 && e == 0) ? bias4[mt] : acc[mt]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4 v = acc[mt];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = relu_bits(v[r]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rout,
                                                 vo + mt * 64u, 0, CURLA_ACT_STORE_POLICY);
        }
        vo += out_row;
      };

      Row S0, S1, S2, S3, S4;
      load_row(S0, 0), load_row(S1, 1), load_row(S2, 2);
      for (int t = 0;;) {  // rows of step t sit in sets (2t, 2t+1, 2t+2) mod 5
        step(S0, S1, S2, S3, S4, t);
        if (++t >= n) break;
        step(S2, S3, S4, S0, S1, t);
        if (++t >= n) break;
        step(S4, S0, S1, S2, S3, t);
        if (++t >= n) break;
        step(S1, S2, S3, S4, S0, t);
        if (++t >= n) break;
        step(S3, S4, S0, S1, S2, t);
        if (++t >= n) break;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the first layer from a float NHWC minibatch, "row walk" form:
//   dW[co][c][dy][dx] = scale * sum over (sample, y, x) of g[y][x][co] * src[2y + dy][2x + dx][c],   db[co] = sum g
// GEMM view: D[co][k'] with k' = (dy, dx C + c), K = pixels (4 per MFMA: one per lane group).  The operand indexed
// [k'][pixel] wants ONE k' per lane and MFMA -- and for a fixed tap row dy the 3 C patch values of a pixel are contiguous
// floats, so a lane takes FOUR consecutive ones with one 16-byte load and spends them on four MFMAs (four accumulator
// tiles); the 16 lanes of a group are (dy, group-of-four) units, 3 ceil(3C/4) <= 32 of them = two loads.  The gradient
// operand is indexed [co][pixel]: a lane takes output channels 2 li and 2 li + 1 (the two channel halves, by parity)
// with one 8-byte load.  So a step (4 pixels, one per lane group, walking DOWN four pixel columns) is 2 + 1 loads and
// 16 MFMAs into 16 accumulator tiles (64 VGPRs) -- against 7 LDS reads with computed addresses, a (row, column)
// division and 14 MFMAs per 4 pixels in the banded kernel (wgrad1_kernel).  No LDS until the final cross-wave sum, no
// barrier; loads three steps ahead, unconditional.
// ---------------------------------------------------------------------------------------------------------------------
struct Wgrad1Args {
  const float* src;  // [B][Hc][Wc][C] float NHWC
  const float* g;    // [B][Ho][Wo][32]
  float* partial;    // [grid][32 C 9 + 32]
  int B, Hc, Wc, Ho, Wo;
  float scale;
  Geom gg;  // strips of 4 OUTPUT COLUMNS (plan_units over Wo, 4)
};

template <int C, int NW>
__device__ __forceinline__ void wgrad1_body(const Wgrad1Args& a, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NG = (3 * C + 3) / 4;  // groups of four patch values per tap row
  constexpr int NU = 3 * NG;           // (dy, group) units: <= 32
  static_assert(NU <= 32, "two 16-lane loads cover the patch");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const Geom& G = a.gg;

  f32x4 acc[2][4][2];  // [unit half h][e][output-channel parity]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int par = 0; par < 2; ++par) acc[h][e][par] = f32x4{0, 0, 0, 0};
  float bsum[2] = {0.f, 0.f};

  const int cnt = bid < a.B ? (a.B - bid + nblk - 1) / nblk : 0;
  const int nstrips = G.nfull + G.ntr;
  const int T = cnt * G.steps;
  const int lo = (int)((long)wave * T / NW), hi = (int)((long)(wave + 1) * T / NW);
  const int in_row = a.Wc * C * 4, g_row = a.Wo * 128;

  int before = 0;
  for (int si = 0; si < cnt; ++si) {
    const int b = bid + si * nblk;
    for (int k = 0; k < nstrips; ++k) {
      const int n_strip = k < G.nfull ? G.Ho : G.nr;
      const int a0 = lo > before ? lo : before;
      const int a1 = hi < before + n_strip ? hi : before + n_strip;
      const int sb = a0 - before, n = a1 - a0;  // output rows [sb, sb + n) of the strip
      before += n_strip;
      if (n <= 0) continue;

      int x, y0;
      bool lane_on;
      if (k < G.nfull) {
        x = 4 * k + kq, y0 = 0, lane_on = true;
      } else {
        const int u = (k - G.nfull) * 4 + kq;
        const int col = u / G.nseg, sg = u - col * G.nseg;
        lane_on = col < G.brem;
        x = 4 * G.nfull + col, y0 = sg * G.nr;
      }
      const int Y = y0 + sb;
      const __amdgpu_buffer_rsrc_t rin = uniform_rsrc(a.src + (size_t)b * a.Hc * a.Wc * C, a.Hc * in_row);
      const __amdgpu_buffer_rsrc_t rg = uniform_rsrc(a.g + (size_t)b * a.Ho * a.Wo * 32, a.Ho * g_row);
      // unit u = li + 16 h -> (dy, group): four patch values of input row 2 (Y + t) + dy at floats 2 x C + 4 group
      unsigned vin[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int u = li + 16 * h;
        const int dy = u / NG, grp = u - dy * NG;
        vin[h] = (lane_on && u < NU) ? (unsigned)(((2 * Y + dy) * a.Wc * C + 2 * x * C + 4 * grp) * 4) : 0x80000000u;
      }
      const unsigned vg = lane_on ? (unsigned)((Y * a.Wo + x) * 128 + li * 8) : 0x80000000u;

      struct Ld {
        f32x4 d[2];
        f32x2 g;
      };
      auto issue = [&](Ld& L, int t) {
        const unsigned sd = __builtin_amdgcn_readfirstlane((unsigned)(2 * t * in_row));
        const unsigned sg_ = __builtin_amdgcn_readfirstlane((unsigned)(t * g_row));
        L.d[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, vin[0], sd, 0));
        L.d[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, vin[1], sd, 0));
        L.g = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rg, vg, sg_, 0));
      };
      auto step = [&](Ld& L, int t) {
        const f32x4 d0 = L.d[0], d1 = L.d[1];
        const f32x2 gv = L.g;
        issue(L, t + 3);  // (past the piece: rows of the image that are dropped, or out of range)
        __builtin_amdgcn_sched_barrier(0);
        bsum[0] += gv[0], bsum[1] += gv[1];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int par = 0; par < 2; ++par) {
            acc[0][e][par] = mfma16(gv[par], d0[e], acc[0][e][par]);
            acc[1][e][par] = mfma16(gv[par], d1[e], acc[1][e][par]);
          }
        __builtin_amdgcn_sched_barrier(0);
      };
      Ld L0, L1, L2;
      issue(L0, 0), issue(L1, 1), issue(L2, 2);
      for (int t = 0;;) {
        step(L0, t);
        if (++t >= n) break;
        step(L1, t);
        if (++t >= n) break;
        step(L2, t);
        if (++t >= n) break;
      }
    }
  }

  // cross-wave sum in wave order (deterministic), one slab per workgroup in the layout wgrad_reduce_kernel reads:
  // [co][c][dy][dx] then the 32 bias gradients
  bsum[0] += __shfl_xor(bsum[0], 16), bsum[1] += __shfl_xor(bsum[1], 16);
  bsum[0] += __shfl_xor(bsum[0], 32), bsum[1] += __shfl_xor(bsum[1], 32);
  constexpr int nw = 32 * C * 9;
  for (int w = 0; w < NW; ++w) {
    if (wave == w) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int u = li + 16 * h;  // D column = the B operand's lane li
        const int dy = u / NG, grp = u - dy * NG;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int rr = 4 * grp + e;
          if (u < NU && rr < 3 * C) {
            const int dx = rr / C, c = rr - dx * C;
#pragma unroll
            for (int par = 0; par < 2; ++par)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int co = 2 * (4 * kq + r) + par;  // D row = the A operand's lane li' = 4 kq + r -> channels 2 li', 2 li' + 1
                float* d = lds + (co * C + c) * 9 + dy * 3 + dx;
                const float v = acc[h][e][par][r] * a.scale;
                *d = (w == 0) ? v : *d + v;
              }
          }
        }
      }
      if (kq == 0) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
          float* d = lds + nw + 2 * li + par;
          *d = (w == 0) ? bsum[par] : *d + bsum[par];
        }
      }
    }
    __syncthreads();
  }
  float* slab = a.partial + (size_t)bid * (nw + 32);
  for (int i = tid; i < nw + 32; i += 64 * NW) slab[i] = lds[i];
}

}  // namespace rw
