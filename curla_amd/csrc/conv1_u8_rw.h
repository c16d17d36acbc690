// First conv layer (3x3, stride 2, C -> 32, + bias + ReLU) and its weight gradient straight from the uint8 replay ring,
// "row walk" form, gfx950.  Included by conv.hip.  Reference semantics: utils.py:151-166 (gather the sampled stacks,
// crop), encoder.py:54-57,78-81 (obs / 255, Conv2d(C, 32, 3, stride=2), ReLU) and their autograd.
//
// The banded kernels (conv1_fwd_u8_kernel, wgrad1_u8_kernel) keep a sample's crop as bytes in LDS and read the MFMA
// operand from it one byte per lane and k-step: 21 ds_read_u8 + 21 conversions + the tile's coordinates per 42 MFMAs, on
// LDS banks that conflict on almost every second cycle, behind a staging pass and two barriers per sample.  Here
// nothing is staged.  The 3 C operand bytes of an output pixel and one input row are 3 C CONTIGUOUS bytes of the NHWC
// frame (pixels 2x .. 2x+2 of the crop), so the four lane groups of a wave take E = ceil(3C/4) bytes each with ONE
// buffer load, and v_cvt_f32_ubyte0..3 turn them into the operand values -- the gather (ring slot), the crop offset
// and the u8 -> f32 conversion all happen in that load + conversion.  The run starts at any byte.  Loads at byte
// alignment work (SH_MEM_CONFIG.alignment_mode is "unaligned" under ROCm) but keep the texture addresser busy about
// twice as long as aligned ones, and with two loads + two stores per 42 MFMAs on 16 waves per CU that shows (-8 %
// with aligned addresses); so when the row pitch is a multiple of 4 -- a lane's misalignment is then the same for
// every row of its piece -- the loads are aligned dwords and v_alignbyte shifts the run into place (AL = true).  A wave owns 16 output columns and all 32 output
// channels and walks down: output row y needs input rows 2y, 2y+1, 2y+2, the last of which is the next row's first, so
// a step loads and converts two new rows (2 loads, 2 E conversions) for 6 E MFMAs.  No LDS, no barrier.
//
// Status (rounds 3-4; since round 5 this walk is what option conv1_u8 = auto takes, see conv.hip use_rw_u8 and the bf16 form at the end
// of this file): opt-in.  Alone -- the same ring slots re-read out of the Infinity Cache from launch to
// launch -- it is the faster kernel (1024 samples of 76x76x9: 66-71 us against 85; 1536: 100-105 against 128).  On slots
// drawn afresh from a ring of gigabytes for every launch, which is what update() does, it is the slower one: 114 us on
// average against 104 under rocprofv3 on the same box (88 against 88 per 1024 samples in the microbenchmark).  More
// prefetch distance (1 / 2 / 4 steps: 90.6 / 86.4 / 88.4 us), five waves per SIMD instead of four, and pulling a
// workgroup's crops into the L2 in one coalesced burst first (92 against 82) did not change that; the banded kernel,
// which reads a crop once in one burst and then works out of LDS, does not see the difference -- and the default is now
// the hybrid of the two (conv1_u8_walk_kernel in conv.hip: that staging, this loop reading from LDS).  DESIGN.md section 6.
#pragma once

namespace rw {

struct Conv1U8Problem {
  const int64_t* idx;  // ring slot per sample (null: sample b is slot b)
  const int32_t* h1;   // crop origin per sample (null: 0)
  const int32_t* w1;
  const float* w;      // OIHW [32][C][3][3]
  const float* bias;   // [32]
  float* out;          // [B][Ho][Wo][32]
  int B;
};

struct Conv1U8Args {
  const uint8_t* src;  // [slots][Hs][Ws][C]
  Conv1U8Problem p[2];
  int Hs, Ws, Ho, Wo;
  float scale;
  Geom g;  // strips of 16 OUTPUT COLUMNS (plan_units over Wo)
};

// p[i] through the scalar data cache (constant address space): for wave-uniform indices into memory that no kernel
// writes while this one runs
template <class T>
__device__ __forceinline__ T const_load(const T* p, int i) {
  return ((const T __attribute__((address_space(4)))*)(unsigned long long)p)[i];
}

template <int NL>
struct RawBytes {
  unsigned d[NL];
};

template <int NL>
__device__ __forceinline__ void load_raw(RawBytes<NL>& R, const __amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned so) {
  static_assert(NL >= 1 && NL <= 4, "1..4 dwords");
  if constexpr (NL == 1) {
    R.d[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0);
  } else if constexpr (NL == 2) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);
    R.d[0] = v[0], R.d[1] = v[1];
  } else if constexpr (NL == 3) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b96(rs, vo, so, 0);
    R.d[0] = v[0], R.d[1] = v[1], R.d[2] = v[2];
  } else {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
    R.d[0] = v[0], R.d[1] = v[1], R.d[2] = v[2], R.d[3] = v[3];
  }
}

// byte e of the loaded run as a float (v_cvt_f32_ubyte<e & 3>: extraction and conversion in one instruction)
template <int NL>
__device__ __forceinline__ float byte_f32(const RawBytes<NL>& R, int e) {
  return (float)((R.d[e >> 2] >> (8 * (e & 3))) & 0xffu);
}

// OIHW weights -> k-major image [dy][rr (4 E, zero padded past 3 C)][cout 32] + the 32 biases in LDS, with the 1/255 of
// `obs / 255.` (encoder.py:78) folded in: the index arithmetic is paid once per weight, coalesced over the workgroup,
// and every lane then reads its 6 E values at compile-time offsets from one base
template <int C>
constexpr int conv1_u8_image_floats() {
  return 3 * 4 * ((3 * C + 3) / 4) * 32 + 32;
}

template <int C, int NT>
__device__ __forceinline__ void conv1_u8_stage_weights(float* img, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float scale, int tid) {
  constexpr int E = (3 * C + 3) / 4, KR = 4 * E;
  for (int i = tid; i < 3 * KR * 32; i += NT) {
    const int co = i & 31, k = i >> 5;
    const int dy = k / KR, rr = k - dy * KR;
    const int dx = rr / C, c = rr - dx * C;
    img[i] = rr < 3 * C ? w[(co * C + c) * 9 + dy * 3 + dx] * scale : 0.f;
  }
  if (tid < 32) img[3 * KR * 32 + tid] = bias[tid];
}

// Work distribution: the steps (one output row of one 16-column strip) of ALL samples of both minibatches form one
// pool; a workgroup takes an equal contiguous share of it and its waves equal shares of that, cut wherever they fall
// (pieces of strips).  Whole samples per workgroup would leave most of the chip waiting for the workgroups that got one
// sample more (1024 samples on 341 workgroups: one of them runs 4 instead of 3 -- a third longer than the rest).
// steps between a row's request and its use (1: 90.6 us, 2: 86.4, 4: 88.4 at 1024 samples from fresh ring slots; the
// unrolled rotation below is written for 2)
constexpr int kDepth = 2;

template <int C, int NW, bool AL>
__device__ __forceinline__ void conv1_u8_body(const Conv1U8Args& a, const float* lds_img, const int bid, const int nblk) {
  constexpr int E = (3 * C + 3) / 4;               // operand bytes per lane group and input row
  constexpr int NW_ = (E + 3) / 4;                 // dwords that hold them once they start at byte 0
  constexpr int NL = AL ? (E + 3 + 3) / 4 : NW_;   // dwords per load (aligned: the run starts at byte 0..3 of the first)
  constexpr int IMG = conv1_u8_image_floats<C>();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const Geom& G = a.g;

  // (32-bit throughout: the host checks that the pool has fewer than 2^31 / 8 steps)
  const int B0 = a.p[0].B;
  const int pool = (B0 + a.p[1].B) * G.steps;
  const int per = pool / nblk, rem = pool - per * nblk;
  const int g0 = bid * per + (bid < rem ? bid : rem), len = per + (bid < rem ? 1 : 0);
  const int lo = g0 + len * wave / NW, hi = g0 + len * (wave + 1) / NW;
  const int in_row = a.Ws * C, out_row = a.Wo * 128;  // bytes per row
  const int frame = a.Hs * in_row;

  // weights: lane (li = cout, kq) holds W[cout][dy][rr = E kq + e] * scale, (dx, c) = (rr / C, rr % C), zero past 3C --
  // of the minibatch its current piece belongs to (re-read from the LDS image when that changes)
  float wr[3][E][2];
  f32x4 bias4[2];
  int have = -1;

  for (int g = lo; g < hi;) {
    const int u = g / G.steps;      // sample (both minibatches counted through)
    const int r = g - u * G.steps;  // step inside the sample -> strip k, row sb of the strip
    int k, sb, n_strip;
    if (r < G.nfull * G.Ho) {
      k = r / G.Ho, sb = r - k * G.Ho, n_strip = G.Ho;
    } else {
      const int q = (r - G.nfull * G.Ho) / G.nr;
      k = G.nfull + q, sb = r - G.nfull * G.Ho - q * G.nr, n_strip = G.nr;
    }
    const int n = hi - g < n_strip - sb ? hi - g : n_strip - sb;  // this wave runs output rows [sb, sb + n) of the strip
    g += n;
    const int prob = u >= B0 ? 1 : 0;
    const int b = u - (prob ? B0 : 0);
    const Conv1U8Problem& P = a.p[prob];
    if (prob != have) {
      have = prob;
      const float* wl = lds_img + prob * IMG + (E * kq) * 32 + li;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) wr[dy][e][mt] = wl[(dy * 4 * E + e) * 32 + mt * 16];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        bias4[mt] = *reinterpret_cast<const f32x4*>(lds_img + prob * IMG + 3 * 4 * E * 32 + mt * 16 + 4 * kq);
    }
    // the crop of the sampled slot: its first byte is the descriptor's base, the frame's end (+ 8 bytes: the last lane
    // group's load window ends up to 5 bytes past the 3 C bytes of a pixel run, and a load that straddles the range
    // reads zeros as a whole -- the ring is followed by 32 readable bytes, curla_hip.h) its range; rows past the crop
    // that a piece's last prefetch touches are in range or read zeros, either way they are never used
    // (scalar loads: the index block is constant for the launch, and a vector load here would make every piece wait for
    // ALL of the wave's outstanding memory operations -- the previous piece's stores included -- before it can start)
    const long long slot = P.idx ? const_load(P.idx, b) : b;
    const int origin = ((P.h1 ? const_load(P.h1, b) : 0) * a.Ws + (P.w1 ? const_load(P.w1, b) : 0)) * C;
    const uint8_t* crop = a.src + (size_t)slot * frame + origin;
    // (aligned form: the descriptor starts at the dword that holds the crop's first byte)
    const int mis = AL ? (int)(reinterpret_cast<uintptr_t>(crop) & 3) : 0;
    const __amdgpu_buffer_rsrc_t rin = uniform_rsrc(crop - mis, frame - origin + 8 + mis);
    const __amdgpu_buffer_rsrc_t rout = uniform_rsrc(P.out + (size_t)b * a.Ho * a.Wo * 32, a.Ho * out_row);
    {
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
      // the lane group's E bytes of crop row 2 Y + r: (2 Y + r) Ws C + 2 x C + E kq bytes from the crop's origin; the
      // lane's own first row is part of its offset, r advances through the (wave-uniform) scalar offset
      const unsigned run = (unsigned)((2 * Y * a.Ws + 2 * x) * C + E * kq + mis);
      const unsigned vin = lane_on ? (AL ? run & ~3u : run) : 0x80000000u;
      const unsigned sh = run & 3u;  // (aligned form) the run's first byte inside its dword, the same for every row
      unsigned vo = lane_on ? (unsigned)((Y * a.Wo + x) * 128 + kq * 16) : 0x80000000u;

      struct Row {
        float v[E];
      };
      using Raw = RawBytes<NL>;
      auto load_row = [&](Raw& R, int r) {  // crop row 2 Y + r
        load_raw<NL>(R, rin, vin, __builtin_amdgcn_readfirstlane((unsigned)(r * in_row)));
      };
      auto convert = [&](Row& F, const Raw& R) {
        RawBytes<NW_> Wd;
#pragma unroll
        for (int j = 0; j < NW_; ++j)
          Wd.d[j] = AL ? __builtin_amdgcn_alignbyte(j + 1 < NL ? R.d[j + 1 < NL ? j + 1 : j] : 0u, R.d[j], sh) : R.d[j];
#pragma unroll
        for (int e = 0; e < E; ++e) F.v[e] = byte_f32<NW_>(Wd, e);
      };
      auto mma_row = [&](f32x4 (&acc)[2], const Row& F, const int dy) {
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma16(wr[dy][e][mt], F.v[e], acc[mt]);
      };
      // output row t of the piece from crop rows 2 (Y + t) (converted by the previous step), + 1, + 2 (bytes in `cur`,
      // requested TWO steps ago: the sampled slots are scattered over a ring of gigabytes, so every crop comes from HBM,
      // and with one step of cover -- enough when a microbenchmark re-reads the same slots out of the Infinity Cache --
      // the kernel ran a quarter slower inside update() than alone).  The rows of step t + 2 are requested first, into
      // the pair of byte registers the previous step freed.  The tap row that is ready goes first and the two new rows
      // are converted under it.
      struct Pair {
        Raw a, b;
      };
      auto step = [&](const Row& r0, Row& r1, Row& r2, const Pair& cur, Pair& nxt, const int t) {
        load_row(nxt.a, 2 * t + 2 * kDepth + 1);
        load_row(nxt.b, 2 * t + 2 * kDepth + 2);
        // (the barriers pin the requests to the top of the step -- left alone the compiler sinks them below the
        // conversions to reuse the byte registers, and every step then waits out most of a memory latency -- and keep
        // the conversions together, one hazard gap for all of them instead of one per k-step)
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
          for (int r = 0; r < 4; ++r) v[r] = relu_bits(v[r]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rout,
                                                 vo + mt * 64u, 0, CURLA_ACT_STORE_POLICY);
        }
        vo += out_row;
      };

      Row S0, S1, S2, S3, S4;
      Pair P0, P1, P2;
      {
        Raw R0;
        load_row(R0, 0), load_row(P0.a, 1), load_row(P0.b, 2), load_row(P1.a, 3), load_row(P1.b, 4);
        convert(S0, R0);
      }
      for (int t = 0;;) {  // rows of step t sit in sets (2t, 2t+1, 2t+2) mod 5, its bytes in pair t mod 3
        step(S0, S1, S2, P0, P2, t);
        if (++t >= n) break;
        step(S2, S3, S4, P1, P0, t);
        if (++t >= n) break;
        step(S4, S0, S1, P2, P1, t);
        if (++t >= n) break;
        step(S1, S2, S3, P0, P2, t);
        if (++t >= n) break;
        step(S3, S4, S0, P1, P0, t);
        if (++t >= n) break;
        step(S0, S1, S2, P2, P1, t);
        if (++t >= n) break;
        step(S2, S3, S4, P0, P2, t);
        if (++t >= n) break;
        step(S4, S0, S1, P1, P0, t);
        if (++t >= n) break;
        step(S1, S2, S3, P2, P1, t);
        if (++t >= n) break;
        step(S3, S4, S0, P0, P2, t);
        if (++t >= n) break;
        step(S0, S1, S2, P1, P0, t);
        if (++t >= n) break;
        step(S2, S3, S4, P2, P1, t);
        if (++t >= n) break;
        step(S4, S0, S1, P0, P2, t);
        if (++t >= n) break;
        step(S1, S2, S3, P1, P0, t);
        if (++t >= n) break;
        step(S3, S4, S0, P2, P1, t);
        if (++t >= n) break;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same layer on the BF16 matrix cores (round 5; 3 C <= 32, i.e. C <= 10).  A uint8 pixel value is EXACT in bf16 (8
// significand bits), so the pixel operand needs ONE part; the weight (with obs / 255's 1 / 255 folded in, as above)
// is split into three bf16 parts (conv_rwb.h), and a product is three exact bf16 x bf16 products accumulated in fp32 --
// the same value an fp32 multiply-add chain gives, up to the order of the sum.  The 3 C <= 32 operand bytes of an output
// pixel and one input row are ONE k-step of v_mfma_f32_16x16x32_bf16: lane group kq takes bytes 8 kq .. 8 kq + 7 of
// the run (one aligned three-dword load + v_alignbyte, then 8 v_cvt_f32_ubyte + 4 packs: a float that holds an integer
// below 256 has nothing in its low half, so the bf16 is its high half) -- 18 matrix instructions of 16 cycles per
// output row of 16 pixels instead of 6 E = 42 of 32 cycles.  The three split parts of every (row tap, channel half)
// weight fragment stay in 72 registers (re-read from an LDS image when a piece changes minibatch).  Everything else --
// the pool of steps, the pieces, the two-steps-ahead row requests -- is conv1_u8_body's.
struct U8Frag {
  unsigned d[4];  // 8 bf16
};

// image of one problem: [dy][mt][part][lane slot (kq * 16 + li)][8 bf16] + 32 float biases
constexpr int kConv1U8B3ImageBytes = 3 * 2 * 3 * 64 * 16 + 32 * 4;

template <int C, int NT>
__device__ __forceinline__ void conv1_u8b_stage_weights(unsigned char* img, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float scale, int tid) {
  static_assert(3 * C <= 32, "one k-step per input row");
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  auto pk = [](float x0, float x1) {
    const f32x2 v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
  };
  for (int item = tid; item < 3 * 2 * 64; item += NT) {
    const int slot = item & 63, dm = item >> 6, dy = dm >> 1, mt = dm & 1;
    const int li = slot & 15, kq = slot >> 4, co = 16 * mt + li;
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int rr = 8 * kq + 2 * q + e, dx = rr / C, c = rr - dx * C;
        x[e] = rr < 3 * C ? w[(co * C + c) * 9 + dy * 3 + dx] * scale : 0.f;
      }
      h[q] = pk(x[0], x[1]);
      const float r0 = x[0] - __builtin_bit_cast(float, h[q] << 16), r1 = x[1] - __builtin_bit_cast(float, h[q] & 0xFFFF0000u);
      m[q] = pk(r0, r1);
      const float s0 = r0 - __builtin_bit_cast(float, m[q] << 16), s1 = r1 - __builtin_bit_cast(float, m[q] & 0xFFFF0000u);
      l[q] = pk(s0, s1);
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4* p = reinterpret_cast<u4*>(img) + (dm * 3) * 64 + slot;
    p[0] = u4{h[0], h[1], h[2], h[3]}, p[64] = u4{m[0], m[1], m[2], m[3]}, p[128] = u4{l[0], l[1], l[2], l[3]};
  }
  if (tid < 32) reinterpret_cast<float*>(img + 3 * 2 * 3 * 64 * 16)[tid] = bias[tid];
}

template <int C, int NW, bool AL>
__device__ __forceinline__ void conv1_u8b_body(const Conv1U8Args& a, const unsigned char* lds_img, const int bid,
                                               const int nblk) {
  constexpr int E = 8;                             // operand bytes per lane group and input row (one k-step of 32)
  constexpr int NW_ = 2;                           // dwords that hold them once they start at byte 0
  constexpr int NL = AL ? 3 : NW_;                 // dwords per load (aligned: the run starts at byte 0..3 of the first)
  typedef short bf16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const Geom& G = a.g;

  const int B0 = a.p[0].B;
  const int pool = (B0 + a.p[1].B) * G.steps;
  const int per = pool / nblk, rem = pool - per * nblk;
  const int g0 = bid * per + (bid < rem ? bid : rem), len = per + (bid < rem ? 1 : 0);
  const int lo = g0 + len * wave / NW, hi = g0 + len * (wave + 1) / NW;
  const int in_row = a.Ws * C, out_row = a.Wo * 128;  // bytes per row
  const int frame = a.Hs * in_row;

  u4 wf[3][2][3];  // [row tap][channel half][part h / m / l]: the lane's fragment of the split, scaled weights
  f32x4 bias4[2];
  int have = -1;

  for (int g = lo; g < hi;) {
    const int u = g / G.steps;      // sample (both minibatches counted through)
    const int r = g - u * G.steps;  // step inside the sample -> strip k, row sb of the strip
    int k, sb, n_strip;
    if (r < G.nfull * G.Ho) {
      k = r / G.Ho, sb = r - k * G.Ho, n_strip = G.Ho;
    } else {
      const int q = (r - G.nfull * G.Ho) / G.nr;
      k = G.nfull + q, sb = r - G.nfull * G.Ho - q * G.nr, n_strip = G.nr;
    }
    const int n = hi - g < n_strip - sb ? hi - g : n_strip - sb;  // this wave runs output rows [sb, sb + n) of the strip
    g += n;
    const int prob = u >= B0 ? 1 : 0;
    const int b = u - (prob ? B0 : 0);
    const Conv1U8Problem& P = a.p[prob];
    if (prob != have) {
      have = prob;
      const unsigned char* im = lds_img + prob * kConv1U8B3ImageBytes;
      const u4* wl = reinterpret_cast<const u4*>(im) + lane;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int p = 0; p < 3; ++p) wf[dy][mt][p] = wl[((dy * 2 + mt) * 3 + p) * 64];
      const float* bl = reinterpret_cast<const float*>(im + 3 * 2 * 3 * 64 * 16);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) bias4[mt] = *reinterpret_cast<const f32x4*>(bl + mt * 16 + 4 * kq);
    }
    // (descriptors, scalar index loads, lane geometry: as conv1_u8_body; the lane group's run is 8 bytes at 8 kq, of
    // which bytes >= 3 C -- the next pixels' -- meet zero weights; the last lane group's window ends up to 5 + 3 bytes
    // past the 3 C bytes of a pixel run: + 16 bytes of range, the ring's 32 readable bytes of slack cover them)
    const long long slot = P.idx ? const_load(P.idx, b) : b;
    const int origin = ((P.h1 ? const_load(P.h1, b) : 0) * a.Ws + (P.w1 ? const_load(P.w1, b) : 0)) * C;
    const uint8_t* crop = a.src + (size_t)slot * frame + origin;
    const int mis = AL ? (int)(reinterpret_cast<uintptr_t>(crop) & 3) : 0;
    const __amdgpu_buffer_rsrc_t rin = uniform_rsrc(crop - mis, frame - origin + 16 + mis);
    const __amdgpu_buffer_rsrc_t rout = uniform_rsrc(P.out + (size_t)b * a.Ho * a.Wo * 32, a.Ho * out_row);
    {
      int x, y0;
      bool lane_on;
      if (k < G.nfull) {
        x = 16 * k + li, y0 = 0, lane_on = true;
      } else {
        const int uu = (k - G.nfull) * 16 + li;
        const int col = uu / G.nseg, sg = uu - col * G.nseg;
        lane_on = col < G.brem;
        x = 16 * G.nfull + col, y0 = sg * G.nr;
      }
      const int Y = y0 + sb;
      const unsigned run = (unsigned)((2 * Y * a.Ws + 2 * x) * C + E * kq + mis);
      const unsigned vin = lane_on ? (AL ? run & ~3u : run) : 0x80000000u;
      const unsigned sh = run & 3u;
      unsigned vo = lane_on ? (unsigned)((Y * a.Wo + x) * 128 + kq * 16) : 0x80000000u;

      using Raw = RawBytes<NL>;
      auto load_row = [&](Raw& R, int rr) {  // crop row 2 Y + rr
        load_raw<NL>(R, rin, vin, __builtin_amdgcn_readfirstlane((unsigned)(rr * in_row)));
      };
      // 8 bytes -> 8 bf16: the float of an integer below 256 is exact and its low half is zero
      auto convert = [&](u4& F, const Raw& R) {
        RawBytes<NW_> Wd;
#pragma unroll
        for (int j = 0; j < NW_; ++j)
          Wd.d[j] = AL ? __builtin_amdgcn_alignbyte(R.d[j + 1], R.d[j], sh) : R.d[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float f0 = byte_f32<NW_>(Wd, 2 * q), f1 = byte_f32<NW_>(Wd, 2 * q + 1);
          F[q] = (__builtin_bit_cast(unsigned, f0) >> 16) | (__builtin_bit_cast(unsigned, f1) & 0xFFFF0000u);
        }
      };
      auto mma_row = [&](f32x4 (&acc)[2], const u4& F, const int dy) {
        const bf16x8 xb = __builtin_bit_cast(bf16x8, F);
#pragma unroll
        for (int p = 2; p >= 0; --p)  // smallest part first
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[dy][mt][p]), xb, acc[mt], 0, 0, 0);
      };
      struct Pair {
        Raw a, b;
      };
      auto step = [&](const u4& r0, u4& r1, u4& r2, const Pair& cur, Pair& nxt, const int t) {
        load_row(nxt.a, 2 * t + 2 * kDepth + 1);
        load_row(nxt.b, 2 * t + 2 * kDepth + 2);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2] = {bias4[0], bias4[1]};
        mma_row(acc, r0, 0);
        convert(r1, cur.a);
        convert(r2, cur.b);
        mma_row(acc, r1, 1);
        mma_row(acc, r2, 2);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4 v = acc[mt];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) v[rr] = relu_bits(v[rr]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rout, vo + mt * 64u, 0, CURLA_ACT_STORE_POLICY);
        }
        vo += out_row;
      };

      u4 S0, S1, S2, S3, S4;
      Pair P0, P1, P2;
      {
        Raw R0;
        load_row(R0, 0), load_row(P0.a, 1), load_row(P0.b, 2), load_row(P1.a, 3), load_row(P1.b, 4);
        convert(S0, R0);
      }
      for (int t = 0;;) {  // rows of step t sit in sets (2t, 2t+1, 2t+2) mod 5, its bytes in pair t mod 3
        step(S0, S1, S2, P0, P2, t);
        if (++t >= n) break;
        step(S2, S3, S4, P1, P0, t);
        if (++t >= n) break;
        step(S4, S0, S1, P2, P1, t);
        if (++t >= n) break;
        step(S1, S2, S3, P0, P2, t);
        if (++t >= n) break;
        step(S3, S4, S0, P1, P0, t);
        if (++t >= n) break;
        step(S0, S1, S2, P2, P1, t);
        if (++t >= n) break;
        step(S2, S3, S4, P0, P2, t);
        if (++t >= n) break;
        step(S4, S0, S1, P1, P0, t);
        if (++t >= n) break;
        step(S1, S2, S3, P2, P1, t);
        if (++t >= n) break;
        step(S3, S4, S0, P0, P2, t);
        if (++t >= n) break;
        step(S0, S1, S2, P1, P0, t);
        if (++t >= n) break;
        step(S2, S3, S4, P2, P1, t);
        if (++t >= n) break;
        step(S4, S0, S1, P0, P2, t);
        if (++t >= n) break;
        step(S1, S2, S3, P1, P0, t);
        if (++t >= n) break;
        step(S3, S4, S0, P2, P1, t);
        if (++t >= n) break;
      }
    }
  }
}

}  // namespace rw
