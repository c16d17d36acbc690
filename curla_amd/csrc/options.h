// Run-time kernel-selection options of the library (curla_set_option / curla_get_option in include/curla_hip.h).
// Every option has a default that is the measured-best path; the alternatives are kept because they are the fallback
// for shapes the default does not take, or the A/B partner a measurement needs -- and every value of every option runs
// under the whole-update parity test of tests/test_gpu_switches.py.  The environment variable of the same name
// (CURLA_<NAME>) only sets the INITIAL value, read once when the library is first used.
#pragma once

enum CurlaOpt {
  kOptConv1U8 = 0,  // first layer from the uint8 ring: 0 auto (= rwb where 3 C <= 32, else rw), 1 hybrid (crop in LDS, row walk
                    // out of LDS), 2 band, 3 rw (no LDS, f32-input MFMA), 4 rwb (no LDS, bf16 matrix cores: uint8 is exact in bf16)
  kOptConv1F32,     // first layer (and its weight gradient) from a float NHWC minibatch: 0 rw (conv1_rw.h), 1 band
  kOptS1Fwd,        // stride-1 forward / data gradient: 0 auto (= b3), 1 Winograd F(2,3) on the f32-input MFMA (conv_rw.h),
                    // 2 F(4,3) (conv_rw43.h), 3 bf16x3 on the bf16 matrix cores behind F(2,3) (conv_rwb.h)
  kOptBwdSplit,     // stride-1 backward launch: 0 auto, 1 two + two workgroups per CU, 2 one + one side by side
  kOptGemmTile,     // tile shape of the tiled GEMM: 0 auto, 1 64x64, 2 64x32, 3 32x32, 4 128x64 (bf16x3) wherever it applies
  kOptLinearBwd,    // backward of a linear layer: 0 one launch for dW and dx, 1 two launches
  kOptGemmMfma,     // arithmetic of the tiled GEMM and of the encoder fc forward: 0 auto (GEMM: bf16x3 on 128x64 tiles where those
                    // fill the chip, else f32; fc forward: bf16x3), 1 f32 (the f32-input MFMA everywhere), 2 b3 (bf16x3 on the
                    // bf16 matrix cores for every interior, aligned tile)
  kOptS1Wgrad,      // stride-1 weight gradient: 0 auto (= xy), 1 x (Winograd F(3,2) along x, conv_rw_wgrad.h), 2 xy (both directions,
                    // conv_rw_wgrad2.h)
  kOptWgrad1U8,     // first-layer weight gradient from the uint8 ring: 0 auto (= b16 where output rows hold >= 8 pixels), 1 f32 (the
                    // f32-input MFMA, wgrad1_u8_kernel), 2 b16 (bf16 matrix cores: a uint8 pixel is exact in one bf16, the gradient
                    // is split into three; wgrad1_u8b_kernel)
  kOptCount
};

int curla_opt(int id);  // current value (options.hip)
