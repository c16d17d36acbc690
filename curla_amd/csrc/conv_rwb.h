// Stride-1 32->32 3x3 convolution, row walk on the BF16 matrix cores with fp32-accurate operands ("bf16x3") and 1-D
// Winograd F(2,3) along x: forward (bias + ReLU) and data gradient (ReLU mask of the layer below), gfx950.  Included by
// conv.hip after conv_rw.h, whose argument structs, strip planner and buffer descriptors it shares.
//
// Reference semantics: encoder.py:59-63,84-87 (Conv2d(32, 32, 3, stride=1) + ReLU) and their autograd, in float32.
//
// Why.  The f32-input MFMA (conv_rw.h, conv_rw43.h) runs on the SIMD's own FMA lanes at 64 FLOP/clk -- 1/16 of the rate
// of the bf16 matrix cores -- and VALU work does not overlap with it.  Round 4 left the F(2,3) forward at 0.877 of that
// pipe: nothing more to gain there.  An fp32 number is the EXACT sum of three bf16 numbers (8 + 8 + 8 significand
// bits, each with fp32's exponent range: no scaling, no overflow, gradients of 1e-9 split as well as activations of
// 10): x = xh + xm + xl.  A product of two such sums needs the six terms whose weight is >= 2^-16 of the leading one:
//     x w  ~=  xh wh + (xh wm + xm wh) + (xh wl + xm wm + xl wh)          (dropped: xm wl, xl wm, xl wl <= 2^-24 |x w|)
// each of which is an EXACT bf16 x bf16 product accumulated in fp32 by v_mfma_f32_16x16x32_bf16: six matrix
// instructions of 16 cycles replace the k = 32 slice of a Winograd position that costs 8 f32-input instructions of 32
// cycles -- and the splitting arithmetic issues from the VALU while the matrix cores run.
// Accuracy: the terms kept carry 24+ bits of every operand; tools/micro/bf16x3_error.py measures an rms error of a
// K = 288 dot product of 1.3e-7 of its scale against 2.2e-7 for a float32 fmaf chain (the reference's own arithmetic);
// the Winograd transforms are those of conv_rw.h (+-1, 1/2: computed in fp32 BEFORE the split).
// tests/test_gpu_fullsize.py holds the kernel to the float64 arbiter like the other forms.
//
// What bounds it.  Not the matrix cores alone: a SIMD issues one vector instruction (VALU or MFMA) per wave per 4
// cycles, a 16x16x32 MFMA holds the issue port for 8 of its 16 cycles, and splitting one fp32 value costs 5.5 VALU
// instructions.  The DIRECT form (3 x-taps = 3 shifted operand loads and splits per pixel: 108 MFMAs + ~175 VALU per 16
// pixels, 1584 issue cycles against 1728 matrix cycles) measured 467 us on configs[1]'s critic stack, no better than
// F(4,3)'s 458; timing-only ablations put 130 us on the split arithmetic and 130 us on the weight reads.  With the
// Winograd transform in front a window of 4 pixels feeds 2 outputs through 4 positions instead of 6 (tap, pixel)
// pairs: 2/3 of the matrix instructions AND 2/3 of the splits per output.
//
// Walk.  As conv_rw.h: a wave owns 16 pixel-PAIR columns and all 32 output channels and walks down the image.  At step t
// every lane loads the 4-pixel window of its pair, 8 channels (its quarter of k = 32: two 16-byte loads per pixel,
// zero outside the image through the buffer range check), transforms it (4 positions x 8 channels), splits the 32
// values into 4 x 3 packed operands and feeds each position to the three output rows that use the row: row tap 2
// completes output row t-2, tap 1 continues t-1, tap 0 starts t.  144 matrix instructions per step on 3 x 2 x 4
// accumulators (96 registers).  The operand that changes per instruction is the WEIGHT: each (row tap, position,
// channel half)'s three split parts sit in LDS in lane order (one conflict-free ds_read_b128 per part; 72 KB per
// problem), read one group ahead.
#pragma once

namespace rwb {

using rw::Args;
using rw::Geom;
using rw::Problem;
using rw::uniform_rsrc;

typedef short bf16x8 __attribute__((ext_vector_type(8)));      // the MFMA's A / B operand: 8 bf16 in 4 VGPRs
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

constexpr int kWBytes = 3 * 4 * 2 * 3 * 64 * 16;  // split filter of one problem in LDS: [dy][pos][mt][part][lane][8 bf16]

// strip plan: 16 pixel-pair columns per wave (conv_rw.h's)
inline Geom plan(int Hi, int Wi, int Ho, int Wo) { return rw::plan(Hi, Wi, Ho, Wo); }

struct B3 {
  u32x4 h, m, l;  // 8 bf16 each: element j in the low / high half of word j >> 1
};

// two fp32 values -> one word of two round-to-nearest-even bf16 (v_cvt_pk_bf16_f32 on gfx950: a plain cast, so that the
// compiler's hazard recogniser sees the instruction), first value in the low half
__device__ __forceinline__ unsigned cvt_pk_bf16(float x0, float x1) {
  const f32x2 v = {x0, x1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// 8 fp32 values (a lane's quarter of the 32 channels of one Winograd position) -> the three bf16 operands: per pair of
// values three packed conversions, four half-word extractions and four subtractions (5.5 VALU instructions per value)
__device__ __forceinline__ B3 split8(const float (&v)[8]) {
  B3 o;
#if defined(RWB_ABL) && (RWB_ABL & 1)  // timing-only ablation: no split arithmetic (results wrong)
#pragma unroll
  for (int p = 0; p < 4; ++p) o.h[p] = __builtin_bit_cast(unsigned, v[2 * p]), o.m[p] = __builtin_bit_cast(unsigned, v[2 * p + 1]);
  o.l = o.h;
  return o;
#endif
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = v[2 * p], x1 = v[2 * p + 1];
    const unsigned h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xFFFF0000u);
    const unsigned m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xFFFF0000u);
    o.h[p] = h, o.m[p] = m, o.l[p] = cvt_pk_bf16(s0, s1);
  }
  return o;
}

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// OIHW weights -> the Winograd-transformed (conv_rw.h: rw::filter_transform), split filter image.  A work item is one
// LANE SLOT of one (row tap, channel half): output channel co = 16 mt + li, input channels 8 kq .. 8 kq + 7 -- the 24
// taps it reads, the 4 x 8 transformed values, their split and twelve 16-byte LDS writes (lane slot kq * 16 + li of the
// (dy, pos, mt, part) blocks: a wave writes 1 KB contiguous, no bank conflicts).  (The first version gave a thread one
// (o, i) pair and 108 two-byte writes: 16 k cycles per layer, 8 % of a configs[1] launch; this one ~4 k.)
// (w1 may be null: one problem.)
template <int MODE, int NT>
__device__ __forceinline__ void build_filter(unsigned short* lds_w, const float* __restrict__ w0,
                                             const float* __restrict__ w1, int tid) {
  const int nitems = (w1 ? 2 : 1) * 6 * 64;
  for (int item = tid; item < nitems; item += NT) {
    const int slot = item & 63, dm = (item >> 6) % 6, prob = item / 384;
    const int li = slot & 15, kq = slot >> 4, dy = dm >> 1, mt = dm & 1;
    const float* __restrict__ w = prob ? w1 : w0;
    const int co = 16 * mt + li;
    float u[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = 8 * kq + j;
      // forward: cout = o, cin = i, taps as stored.  data gradient: cout = i, cin = o, taps flipped in both directions.
      const float* t = MODE == MODE_FWD ? w + (co * 32 + ci) * 9 + dy * 3 : w + (ci * 32 + co) * 9 + (2 - dy) * 3;
      if (MODE == MODE_FWD)
        rw::filter_transform(t[0], t[1], t[2], u[j]);
      else
        rw::filter_transform(t[2], t[1], t[0], u[j]);
    }
    u32x4* img = reinterpret_cast<u32x4*>(lds_w + prob * (kWBytes / 2));
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = u[j][pos];
      const B3 x = split8(v);
      u32x4* p = img + ((((dy * 4 + pos) * 2 + mt) * 3)) * 64 + slot;
      p[0] = x.h, p[64] = x.m, p[128] = x.l;
    }
  }
}

#ifdef RWB_STAMP
__device__ unsigned long long g_rwb_stamp[16];  // [0] split phase, [1] product phase, [2] finish, [3] steps, [4] whole pieces
#endif

struct Acc {
  f32x4 m[2][4];  // [channel half][Winograd position]: 16 output channels x 16 pairs per entry
};

// One layer for the samples this workgroup owns (b = bid, bid + nblk, ...).  NW waves; lds_w holds the split filters of
// the layer's (up to two) problems, already built and visible.
template <int MODE, int NW>
__device__ __forceinline__ void run_layer(const Geom& G, const Problem& P0, const Problem& P1, const unsigned short* lds_w,
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
    const u32x4* lw = reinterpret_cast<const u32x4*>(lds_w + (second ? kWBytes / 2 : 0)) + lane;
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
      // window pixel c of input row (Y - pad + t): byte offset of this lane's 8 channels inside the sample, or far out
      // of range (a column outside the image; a row outside it is out of range by itself)
      unsigned voff[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = x0 + c - pad;
        const bool ok = lane_on && (unsigned)col < (unsigned)G.Wi;
        voff[c] = ok ? (unsigned)(((Y - pad) * G.Wi + col) * 128 + kq * 32) : 0x80000000u;
      }
      // output pixels (Y + r, x0) and (Y + r, x0 + 1), channels 4 kq .. 4 kq + 3 of each half
      unsigned oa = lane_on ? (unsigned)((Y * G.Wo + x0) * 128 + kq * 16) : 0x80000000u;
      unsigned ob = (lane_on && x0 + 1 < G.Wo) ? oa + 128u : 0x80000000u;

      f32x4 bias[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      if (MODE == MODE_FWD) {
        bias[0] = *reinterpret_cast<const f32x4*>(P.aux + 4 * kq);
        bias[1] = *reinterpret_cast<const f32x4*>(P.aux + 16 + 4 * kq);
      }

      f32x4 raw[4][2];  // [window pixel][first / second four channels]
      auto load_row = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          raw[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, voff[c], 0, 0));
          raw[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, voff[c] + 16u, 0, 0));
        }
      };
      // the weights of group g = (position, row tap, channel half) -- the order the step multiplies them in: three
      // 16-byte reads, issued ONE GROUP AHEAD of their use (two register sets in rotation)
      struct W3 {
        u32x4 h, m, l;
      };
      auto wread = [&](const int g) {
        const int pos = g / 6, dy = 2 - (g / 2) % 3, mt = g & 1;
        const u32x4* p = lw + ((((dy * 4 + pos) * 2 + mt) * 3)) * 64;
        W3 w;
#if defined(RWB_ABL) && (RWB_ABL & 2)  // timing-only ablation: one weight read per group instead of three
        w.h = p[0], w.m = w.h, w.l = w.h;
        return w;
#endif
        w.h = p[0], w.m = p[64], w.l = p[128];
        return w;
      };

      Acc S0, S1, S2;
      f32x4 mk[2][2];  // data gradient: activation below at (pixel a / b, channel half)

      // six products of one (position, row tap, channel half) into one accumulator, smallest terms first (`acc` of a
      // row's first tap: zeros, or the bias for position 1 -- the C operand of the first instruction, no register moves)
      auto mul6 = [&](f32x4 acc, const W3& w, const B3& x) {
#if !(defined(RWB_ABL) && (RWB_ABL & 4))  // timing-only ablation: three of the six products (results wrong)
        acc = mfma_bf16(w.l, x.h, acc);
        acc = mfma_bf16(w.h, x.l, acc);
        acc = mfma_bf16(w.m, x.m, acc);
#endif
        acc = mfma_bf16(w.m, x.h, acc);
        acc = mfma_bf16(w.h, x.m, acc);
        acc = mfma_bf16(w.h, x.h, acc);
        return acc;
      };
      // a completed output row: A^T m (bias already inside position 1), ReLU / ReLU mask, two pixels x two channel halves
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
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ya), rout, oa + mt * 64u, 0, CURLA_ACT_STORE_POLICY);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yb), rout, ob + mt * 64u, 0, CURLA_ACT_STORE_POLICY);
        }
        oa += out_row, ob += out_row;
      };

      // step t: input row t of the piece -> output rows t-2 (completed), t-1, t (started).  On entry Wa holds the
      // weights of group 0; on exit again (of the next step).
      W3 Wa, Wb;
      auto step = [&](Acc& Sdy2, Acc& Sdy1, Acc& Sdy0, const int t) {
#ifdef RWB_STAMP
        const unsigned long long c0 = __builtin_readcyclecounter();
#endif
        // B^T d per channel (v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3), then the split
        B3 X[4];
        {
          float v[4][8];
#pragma unroll
          for (int c8 = 0; c8 < 8; ++c8) {
            const float d0 = raw[0][c8 >> 2][c8 & 3], d1 = raw[1][c8 >> 2][c8 & 3], d2 = raw[2][c8 >> 2][c8 & 3],
                        d3 = raw[3][c8 >> 2][c8 & 3];
            v[0][c8] = d0 - d2, v[1][c8] = d1 + d2, v[2][c8] = d2 - d1, v[3][c8] = d1 - d3;
          }
#pragma unroll
          for (int pos = 0; pos < 4; ++pos) X[pos] = split8(v[pos]);
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
#ifdef RWB_STAMP
        const unsigned long long c1 = __builtin_readcyclecounter();
#endif
#pragma unroll
        for (int g = 0; g < 24; ++g) {
          const int pos = g / 6, which = (g / 2) % 3, mt = g & 1;  // which: 0 = row tap 2, 1 = tap 1, 2 = tap 0
          W3& cur = (g & 1) ? Wb : Wa;
          W3& nxt = (g & 1) ? Wa : Wb;
          nxt = wread(g == 23 ? 0 : g + 1);
          __builtin_amdgcn_sched_barrier(0);
          if (which == 0) {
            if (do2) Sdy2.m[mt][pos] = mul6(Sdy2.m[mt][pos], cur, X[pos]);
          } else if (which == 1) {
            if (do1) Sdy1.m[mt][pos] = mul6(Sdy1.m[mt][pos], cur, X[pos]);
          } else {
            // (the row starts here: zeros, the bias in position 1 -- both outputs of a pair contain +m1)
            const f32x4 z = {0, 0, 0, 0};
            if (do0) Sdy0.m[mt][pos] = mul6(pos == 1 ? bias[mt] : z, cur, X[pos]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#ifdef RWB_STAMP
        const unsigned long long c2 = __builtin_readcyclecounter();
#endif
        if (do2) finish(Sdy2);
#ifdef RWB_STAMP
        const unsigned long long c3 = __builtin_readcyclecounter();
        if (threadIdx.x == 0 && blockIdx.x == 7 && t >= 2 && t < n) {
          g_rwb_stamp[0] += c1 - c0, g_rwb_stamp[1] += c2 - c1, g_rwb_stamp[2] += c3 - c2, g_rwb_stamp[3] += 1;
        }
        if (threadIdx.x == 0 && blockIdx.x == 7 && !(t >= 2 && t < n)) {
          const int e = t < 2 ? t : (t == n ? 2 : 3);
          g_rwb_stamp[11 + e] += c3 - c0;
        }
#endif
      };

#ifdef RWB_STAMP
      const unsigned long long p1 = __builtin_readcyclecounter();
#endif
      load_row();
      Wa = wread(0);
      for (int t = 0;;) {
        step(S1, S2, S0, t);
        if (++t > n + 1) break;
        step(S2, S0, S1, t);
        if (++t > n + 1) break;
        step(S0, S1, S2, t);
        if (++t > n + 1) break;
      }
#ifdef RWB_STAMP
      if (threadIdx.x == 0 && blockIdx.x == 7) {
        g_rwb_stamp[8] += __builtin_readcyclecounter() - p1, g_rwb_stamp[9] += 1, g_rwb_stamp[10] += n;
      }
#endif

    }
  }
}

}  // namespace rwb
