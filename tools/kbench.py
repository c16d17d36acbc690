#!/usr/bin/env python3
"""Per-kernel timing at the BASELINE configs[1] shapes (B=512), with optional
timing-only ablations of the stride-1 conv kernel (needs the -DCURLA_ABLATE
build: tools/kbench.py --build-ablate).  Prints one line per kernel: avg us,
TFLOP/s, fraction of the 157.3 TF fp32 MFMA peak."""
import argparse
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 157.3e12


def build_ablate():
    out = os.path.join(ROOT, "tools", "_build")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libcurla_ablate.so")
    srcs = [os.path.join(ROOT, "curla_amd", "csrc", f) for f in ("conv.hip", "gemm.hip", "heads.hip", "augment.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
                           "-DCURLA_ABLATE", "-o", lib] + srcs)
    return lib


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-ablate", action="store_true")
    ap.add_argument("--ablate", type=str, default="")
    ap.add_argument("--B", type=int, default=512)
    ap.add_argument("--what", type=str, default="conv,gemm,misc")
    ap.add_argument("--slots", type=int, default=2048, help="c1only: ring slots (2048 = 130 MB: Infinity-Cache resident)")
    ap.add_argument("--outs", type=int, default=1, help="c1only: rotate over this many output buffers")
    ap.add_argument("--lib", type=str, default="", help="measure this build of the library instead (A/B against an older commit)")
    args = ap.parse_args()
    if args.build_ablate:
        print(build_ablate())
        return
    from curla_amd import _lib, ops
    flags = [int(x) for x in args.ablate.split(",") if x] or [0]
    if flags != [0]:
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "_build", "libcurla_ablate.so")
        _lib.SIGNATURES["curla_debug_ablate"] = [ctypes.c_int]
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
        os.environ["CURLA_SKIP_SRCHASH"] = "1"
    lib = _lib.load()
    B = args.B
    dev = "cuda"
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731

    def report(name, us, flop):
        print(f"{name:46s} {us:9.1f} us  {flop / us / 1e6:7.1f} TF  {flop / us * 1e6 / PEAK * 100:5.1f}%", flush=True)

    # the first ~0.5 s of MFMA work in a process runs ~8 % slow (clock ramp): burn it before measuring
    xw, ww, bw = torch.relu(r(B, 37, 37, 32)), r(32, 32, 3, 3) * 0.1, r(32) * 0.1
    ow = torch.empty(B, 35, 35, 32, device=dev)
    for _ in range(6000):
        ops.conv_s1_fwd(xw, ww, bw, ow)
    torch.cuda.synchronize()

    if "c1only" in args.what:  # just the two first-layer kernels (for rocprofv3 --pmc passes)
        NS = args.slots
        store = torch.randint(0, 256, (NS * 84 * 84 * 9 + 32,), dtype=torch.uint8, device=dev)
        ring = store[:NS * 84 * 84 * 9].view(NS, 84, 84, 9)
        idx = torch.randint(0, NS, (B,), device=dev)
        h1 = torch.randint(0, 8, (B,), device=dev, dtype=torch.int32)
        w1 = torch.randint(0, 8, (B,), device=dev, dtype=torch.int32)
        obs = ops.ObsRef.from_ring(ring, idx, h1, w1, B, (76, 76))
        w0, b = r(32, 9, 3, 3) * 0.1, r(32) * 0.1
        out = torch.empty(B, 37, 37, 32, device=dev)
        g = r(B, 37, 37, 32)
        dw0, db = torch.empty(32, 9, 3, 3, device=dev), torch.empty(32, device=dev)
        ws0 = torch.empty(ops.wgrad_workspace_floats(9), device=dev)
        fl_ = 2.0 * B * 37 * 37 * 32 * 9 * 9
        for fl in flags:
            if flags != [0]:
                lib.curla_debug_ablate(fl)
            outs = [out] + [torch.empty_like(out) for _ in range(args.outs - 1)]
            it = [0]

            def fwd():
                it[0] += 1
                ops.conv1_fwd(obs, w0, b, outs[it[0] % len(outs)])
            report(f"[abl {fl}] conv1_fwd u8", timeit(fwd), fl_)
            # two minibatches of one ring in one launch, as update() issues it: [obs | next_obs] online + next_obs target
            B2 = B // 2
            idx2 = torch.randint(0, NS, (B2,), device=dev)
            obs2 = ops.ObsRef.from_ring(ring, idx2, h1[:B2].contiguous(), w1[:B2].contiguous(), B2, (76, 76))
            w02, b2, out2 = r(32, 9, 3, 3) * 0.1, r(32) * 0.1, torch.empty(B2, 37, 37, 32, device=dev)
            report(f"[abl {fl}] conv1_fwd2 u8 {B}+{B2}", timeit(lambda: ops.conv1_fwd2(obs, w0, b, out, obs2, w02, b2, out2)), fl_ * 1.5)
            big = torch.empty(B + B2, 37, 37, 32, device=dev)
            report(f"[abl {fl}] conv1_fwd2 u8 {B}+{B2} one out buffer", timeit(lambda: ops.conv1_fwd2(obs, w0, b, big[:B], obs2, w02, b2, big[B:])), fl_ * 1.5)
            report(f"[abl {fl}] conv1_fwd2 u8 {B}+{B2} same weights", timeit(lambda: ops.conv1_fwd2(obs, w0, b, out, obs2, w0, b, out2)), fl_ * 1.5)
            # fresh ring slots every launch (a fixed index set stays in the Infinity Cache from one launch to the next)
            sets = [ops.ObsRef.from_ring(ring, torch.randint(0, NS, (B,), device=dev), h1, w1, B, (76, 76)) for _ in range(16)]
            it2 = [0]

            def fwd_fresh():
                it2[0] += 1
                ops.conv1_fwd(sets[it2[0] % 16], w0, b, out)
            report(f"[abl {fl}] conv1_fwd u8 fresh slots", timeit(fwd_fresh, iters=32), fl_)
            idx3 = torch.cat([idx, idx2])
            obs3 = ops.ObsRef.from_ring(ring, idx3, torch.cat([h1, h1[:B2]]), torch.cat([w1, w1[:B2]]), B + B2, (76, 76))
            report(f"[abl {fl}] conv1_fwd u8 {B + B2} one problem", timeit(lambda: ops.conv1_fwd(obs3, w0, b, big)), fl_ * 1.5)
            report(f"[abl {fl}] conv1_wgrad u8", timeit(lambda: ops.conv1_wgrad(obs, g, dw0, db, ws0)), fl_)
        if flags != [0]:
            lib.curla_debug_ablate(0)

    if "wg1" in args.what:  # the uint8 first-layer weight gradient alone (ablations: 1 no staging, 2 no multiply, 4 no slab)
        NS = args.slots
        store = torch.randint(0, 256, (NS * 84 * 84 * 9 + 32,), dtype=torch.uint8, device=dev)
        ring = store[:NS * 84 * 84 * 9].view(NS, 84, 84, 9)
        h1 = torch.randint(0, 8, (B,), device=dev, dtype=torch.int32)
        w1 = torch.randint(0, 8, (B,), device=dev, dtype=torch.int32)
        sets = [ops.ObsRef.from_ring(ring, torch.randint(0, NS, (B,), device=dev), h1, w1, B, (76, 76)) for _ in range(16)]
        g = r(B, 37, 37, 32)
        dw0, db = torch.empty(32, 9, 3, 3, device=dev), torch.empty(32, device=dev)
        ws0 = torch.empty(ops.wgrad_workspace_floats(9), device=dev)
        fl_ = 2.0 * B * 37 * 37 * 32 * 9 * 9
        for fl in flags:
            if flags != [0]:
                lib.curla_debug_ablate(fl)
            it3 = [0]

            def wg_fresh():
                it3[0] += 1
                ops.conv1_wgrad_slabs(sets[it3[0] % 16], g, ws0, 32)
            report(f"[abl {fl}] conv1_wgrad u8 slabs only, same slots", timeit(lambda: ops.conv1_wgrad_slabs(sets[0], g, ws0, 32)), fl_)
            report(f"[abl {fl}] conv1_wgrad u8 slabs only, fresh slots", timeit(wg_fresh, iters=32), fl_)
        if flags != [0]:
            lib.curla_debug_ablate(0)

    if "fcfwd" in args.what:  # the encoder fc forward of three encoders in one launch: tiled GEMM against the streaming kernel
        Kf, Fd = 30752, 50
        hs = [torch.relu(r(B, Kf)) for _ in range(3)]
        hb = [ops.to_blocked(h, B, Kf) for h in hs]
        Wf = [r(Fd, Kf) * 0.01 for _ in range(3)]
        fl_ = 2.0 * 3 * B * Kf * Fd
        for ks in (32,):
            po = [torch.empty(ks, B, Fd, device=dev) for _ in range(3)]
            report(f"fc fwd x3 gemm_multi ksplit {ks}", timeit(lambda: ops.gemm_multi(hs, Wf, po, B, Fd, Kf, ksplit=ks, split_stride=B * Fd)), fl_)
        for blocked, xs_ in ((False, hs), (True, hb)):
            for ns in (21, 32, 43, 64):
                po = [torch.empty(ns, B, Fd, device=dev) for _ in range(3)]
                report(f"fc fwd x3 fc_fwd_multi nsplit {ns} blocked={blocked}",
                       timeit(lambda: ops.fc_fwd_multi(xs_, Wf, po, B, Fd, Kf, ns, B * Fd, blocked=blocked)), fl_)
        for ns in (43, 64):
            po = [torch.empty(ns, B, Fd, device=dev) for _ in range(2)]
            report(f"fc fwd x2 fc_fwd_multi nsplit {ns} blocked", timeit(lambda: ops.fc_fwd_multi(hb[:2], Wf[:2], po, B, Fd, Kf, ns, B * Fd, blocked=True)), fl_ * 2 / 3)

    if "conv" in args.what:
        w, b = r(32, 32, 3, 3) * 0.1, r(32) * 0.1
        for fl in flags:
            if flags != [0]:
                lib.curla_debug_ablate(fl)
            for H in (37, 35, 33):
                x = torch.relu(r(B, H, H, 32))
                out = torch.empty(B, H - 2, H - 2, 32, device=dev)
                fl_ = 2.0 * B * (H - 2) ** 2 * 32 * 32 * 9
                report(f"[abl {fl}] conv_s1_fwd {H}->{H - 2}", timeit(lambda: ops.conv_s1_fwd(x, w, b, out)), fl_)
            for H in (35, 33, 31):
                g = r(B, H, H, 32)
                below = torch.relu(r(B, H + 2, H + 2, 32))
                gin = torch.empty_like(below)
                fl_ = 2.0 * B * H * H * 32 * 32 * 9
                report(f"[abl {fl}] conv_s1_dgrad {H}->{H + 2}", timeit(lambda: ops.conv_s1_dgrad(g, w, below, gin)), fl_)
        if flags != [0]:
            lib.curla_debug_ablate(0)
        # the whole stack of stride-1 layers of two minibatches in one launch, as update() issues it: critic phase
        # ([obs | next_obs] online + next_obs target) and actor / CURL phase (obs online + positives target)
        for B1, B2 in ((2 * B, B), (B, B)):
            x1, x2 = torch.relu(r(B1, 37, 37, 32)), torch.relu(r(B2, 37, 37, 32))
            ws_, bs_ = [r(32, 32, 3, 3) * 0.1 for _ in range(3)], [r(32) * 0.1 for _ in range(3)]
            o1 = [torch.empty(B1, 35 - 2 * i, 35 - 2 * i, 32, device=dev) for i in range(3)]
            o2 = [torch.empty(B2, 35 - 2 * i, 35 - 2 * i, 32, device=dev) for i in range(3)]
            fl_ = sum(2.0 * (B1 + B2) * (35 - 2 * i) ** 2 * 32 * 32 * 9 for i in range(3))
            if ops.conv_s1_fwd_stack(x1, ws_, bs_, o1, x2, ws_, bs_, o2):
                report(f"conv_s1_fwd_stack 37->31, {B1}+{B2}", timeit(lambda: ops.conv_s1_fwd_stack(x1, ws_, bs_, o1, x2, ws_, bs_, o2)), fl_)
        ws = torch.empty(ops.wgrad_workspace_floats(32), device=dev)
        dw, db = torch.empty(32, 32, 3, 3, device=dev), torch.empty(32, device=dev)
        for H in (37, 35, 33):
            x = torch.relu(r(B, H, H, 32))
            g = r(B, H - 2, H - 2, 32)
            gin = torch.empty_like(x)
            fl_ = 2.0 * 2 * B * (H - 2) ** 2 * 32 * 32 * 9
            report(f"conv_s1_bwd_slabs (wgrad + dgrad) {H}", timeit(lambda: ops.conv_s1_bwd_slabs(x, g, w, gin, ws)), fl_)
        for H in (37, 35, 33):
            x = torch.relu(r(B, H, H, 32))
            g = r(B, H - 2, H - 2, 32)
            fl_ = 2.0 * B * (H - 2) ** 2 * 32 * 32 * 9
            report(f"conv_s1_wgrad(+reduce) {H}", timeit(lambda: ops.conv_s1_wgrad(x, g, dw, db, ws)), fl_)
        store = torch.randint(0, 256, (2048 * 84 * 84 * 9 + 32,), dtype=torch.uint8, device=dev)
        ring = store[:2048 * 84 * 84 * 9].view(2048, 84, 84, 9)
        idx = torch.randint(0, 2048, (B,), device=dev)
        h1 = torch.randint(0, 8, (B,), device=dev, dtype=torch.int32)
        w1 = torch.randint(0, 8, (B,), device=dev, dtype=torch.int32)
        obs = ops.ObsRef.from_ring(ring, idx, h1, w1, B, (76, 76))
        w0 = r(32, 9, 3, 3) * 0.1
        out = torch.empty(B, 37, 37, 32, device=dev)
        fl_ = 2.0 * B * 37 * 37 * 32 * 9 * 9
        for fl in flags:
            if flags != [0]:
                lib.curla_debug_ablate(fl)
            report(f"[abl {fl}] conv1_fwd u8 84->76->37", timeit(lambda: ops.conv1_fwd(obs, w0, b, out)), fl_)
        if flags != [0]:
            lib.curla_debug_ablate(0)
        g = r(B, 37, 37, 32)
        dw0 = torch.empty(32, 9, 3, 3, device=dev)
        ws0 = torch.empty(ops.wgrad_workspace_floats(9), device=dev)
        for fl in flags:
            if flags != [0]:
                lib.curla_debug_ablate(fl)
            report(f"[abl {fl}] conv1_wgrad(+reduce) u8", timeit(lambda: ops.conv1_wgrad(obs, g, dw0, db, ws0)), fl_)
        if flags != [0]:
            lib.curla_debug_ablate(0)

    if "c5first" in args.what:  # the float NHWC first-layer kernels of BASELINE configs[4] (168x168x12 -> 83x83x32)
        Bc = 1024
        xin = torch.rand(Bc, 168, 168, 12, device=dev) * 255.0
        obs = ops.ObsRef.from_nhwc(xin)
        w0, b0 = r(32, 12, 3, 3) * 0.1, r(32) * 0.1
        out = torch.empty(Bc, 83, 83, 32, device=dev)
        g = r(Bc, 83, 83, 32)
        dw0, db0 = torch.empty(32, 12, 3, 3, device=dev), torch.empty(32, device=dev)
        ws0 = torch.empty(ops.wgrad_workspace_floats(12), device=dev)
        fl_ = 2.0 * Bc * 83 * 83 * 32 * 12 * 9
        import curla_amd
        aug = curla_amd.ColorJiggle((168, 168))
        ring = torch.randint(0, 256, (512 * 168 * 168 * 12 + 32,), dtype=torch.uint8, device=dev)[:512 * 168 * 168 * 12].view(512, 168, 168, 12)
        jidx = torch.randint(0, 512, (Bc,), device=dev)
        params, order = aug.draw_params(Bc * 4)
        params, order = params.to(dev), order.to(dev)
        us = timeit(lambda: ops.color_jiggle(ring, jidx, params, order, Bc, xin))
        print(f"{'color_jiggle 168x168x12 -> float NHWC, B=' + str(Bc):46s} {us:9.1f} us  {(Bc * 168 * 168 * 12 * 5) / us / 1e6:7.2f} TB/s", flush=True)
        report(f"conv1_fwd float NHWC 168x168x12, B={Bc}", timeit(lambda: ops.conv1_fwd(obs, w0, b0, out)), fl_)
        report(f"conv1_wgrad(+reduce) float NHWC, B={Bc}", timeit(lambda: ops.conv1_wgrad(obs, g, dw0, db0, ws0)), fl_)

    if "gemm" in args.what:
        H, F, K = 1024, 50, 30752
        for nb in (1, 2):
            x, W, bb, out = r(nb, B, H), r(nb, H, H), r(nb, H), torch.empty(nb, B, H, device=dev)
            fl_ = 2.0 * nb * B * H * H
            report(f"linear_fwd {B}x{H}x{H} nb{nb}", timeit(lambda: ops.linear_fwd(x, B * H, W, H * H, bb, H, out, B * H, B, H, H, nb, relu=1)), fl_)
            report(f"linear_dx  {B}x{H}x{H} nb{nb}", timeit(lambda: ops.linear_dx(x, B * H, W, H * H, out, B * H, B, H, H, nb, mask=x, smask=B * H)), fl_)
            report(f"linear_dw  {B}x{H}x{H} nb{nb}", timeit(lambda: ops.linear_dw(x, B * H, out, B * H, W, H * H, B, H, H, nb)), fl_)
            dWp, dxp = torch.empty(nb, H, H, device=dev), torch.empty(nb, B, H, device=dev)

            def two():
                ops.linear_dw(x, B * H, out, B * H, dWp, H * H, B, H, H, nb)
                ops.linear_dx(x, B * H, W, H * H, dxp, B * H, B, H, H, nb, mask=out, smask=B * H)
            report(f"linear_dw + linear_dx {B}x{H}x{H} nb{nb}", timeit(two), 2 * fl_)
            report(f"linear_bwd (pair)     {B}x{H}x{H} nb{nb}",
                   timeit(lambda: ops.linear_bwd(x, B * H, out, B * H, W, H * H, dWp, H * H, dxp, B * H, B, H, H, nb, mask=out, smask=B * H)), 2 * fl_)
            k1 = 54 if nb == 2 else 50
            x1, W1, dW1, dx1, db1 = r(nb, B, k1), r(nb, H, k1), torch.empty(nb, H, k1, device=dev), torch.empty(nb, B, k1, device=dev), torch.empty(nb, H, device=dev)

            def two1():
                ops.linear_dw(x, B * H, x1, B * k1, dW1, H * k1, B, H, k1, nb, colsum=db1, s_colsum=H)
                ops.linear_dx(x, B * H, W1, H * k1, dx1, B * k1, B, H, k1, nb)
            report(f"first layer dw + dx   {B}x{H}x{k1} nb{nb}", timeit(two1), 4.0 * nb * B * H * k1)
            report(f"first layer pair      {B}x{H}x{k1} nb{nb}",
                   timeit(lambda: ops.linear_bwd(x, B * H, x1, B * k1, W1, H * k1, dW1, H * k1, dx1, B * k1, B, H, k1, nb, db=db1, sdb=H)), 4.0 * nb * B * H * k1)
        h, Wf = r(B, K), r(F, K)
        for ks in (16, 32, 64):
            part = torch.empty(ks, B, F, device=dev)
            report(f"fc fwd split-K {ks} [{B}x{K}]x[{K}x{F}]",
                   timeit(lambda: ops.gemm(h, 0, K, 0, Wf, 0, K, 0, part, F, 0, B, F, K, 1, ksplit=ks, split_stride=B * F)), 2.0 * B * F * K)
        dz, dW, dh = r(B, F), torch.empty(F, K, device=dev), torch.empty(B, K, device=dev)
        report("fc dW (TN)", timeit(lambda: ops.linear_dw(dz, 0, h, 0, dW, 0, B, F, K)), 2.0 * B * F * K)
        report("fc dW streaming", timeit(lambda: ops.fc_dw(dz, h, dW, B, F, K)), 2.0 * B * F * K)
        WT = r(F, K)
        report("fc dh streaming (+ mask)", timeit(lambda: ops.fc_dx(dz, WT, dh, B, F, K, mask=h)), 2.0 * B * F * K)
        report("fc dh (NN + mask)", timeit(lambda: ops.linear_dx(dz, 0, Wf, 0, dh, 0, B, F, K, mask=h)), 2.0 * B * F * K)

    if "misc" in args.what:
        X = r(2, B, 1024)
        o = torch.empty(2, 1024, device=dev)
        report("colsum 2x512x1024", timeit(lambda: ops.colsum(X, B, 1024, 1024, B * 1024, o, 1024, 2)), 2.0 * B * 1024)
        part = r(64, B, 50)
        y = torch.empty(B, 50, device=dev)
        bi, ga, be = r(50), r(50), r(50)
        report("fc_ln_fwd ks64", timeit(lambda: ops.fc_ln_fwd(part, 64, B * 50, 50, bi, ga, be, B, 50, y)), 64.0 * B * 50)


if __name__ == "__main__":
    main()
