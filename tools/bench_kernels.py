#!/usr/bin/env python
"""Per-kernel micro-benchmarks at BASELINE cfg2 shapes (run on the MI355X):

    python tools/bench_kernels.py [attn] [conv] [temporal] [ln] [gn] [--iters N]

Prints achieved TFLOP/s (MFMA kernels, vs 2500 dense bf16) or GB/s (HBM kernels, vs 8000) per shape.
Usable under `rocprofv3 --pmc ...` for counters of one kernel class.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402

DT = torch.bfloat16
DEV = "cuda"


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def rn(*s):
    return torch.randn(*s, device=DEV, dtype=torch.float32).to(DT)


def bench_attn(iters):
    shapes = [("pano L0 self", 32, 5, 8192, 8192, 64, False), ("pers L0 self", 640, 5, 1024, 1024, 64, False),
              ("pano L1 self", 32, 10, 2048, 2048, 64, False), ("pers L1 self", 640, 10, 256, 256, 64, False),
              ("pano L2 self", 32, 20, 512, 512, 64, False), ("pers L0 cross text", 640, 5, 1024, 77, 64, False),
              ("warp L1 e2p", 32, 10, 2048, 5120, 32, True), ("warp L1 p2e", 32, 10, 5120, 2048, 32, True),
              ("warp L2 e2p", 32, 20, 512, 1280, 32, True)]
    for name, B, H, Nq, Nk, D, bias in shapes:
        q, k, v = rn(B, Nq, H * D), rn(B, Nk, H * D), rn(B, Nk, H * D)
        bb = K.pack_attn_bias((torch.rand(Nq, Nk, device=DEV) * 2 - 1).to(DT)) if bias else None      # the form WarpAttn caches
        t = timeit(lambda: K.attention(q, k, v, H, bias=bb, bias_packed=bias), iters)
        fl = 4.0 * B * H * Nq * Nk * D
        print(f"attn  {name:20s} B={B:4d} H={H:2d} Nq={Nq:5d} Nk={Nk:5d} d={D}: {t * 1e3:8.3f} ms  {fl / t / 1e12:7.1f} TF/s "
              f"({fl / t / 2.5e15 * 100:4.1f}% of MFMA peak)")


def bench_attn_variants(iters):
    """WarpAttn (d = 32 + shared bias): one vs two query blocks per wave; text + IP cross attention: two launches with
    accumulate vs the single two-set launch."""
    for name, B, H, Nq, Nk in [("warp L1 e2p", 32, 10, 2048, 5120), ("warp L1 p2e", 32, 10, 5120, 2048), ("warp L2 e2p", 32, 20, 512, 1280)]:
        D = 32
        q, k, v = rn(B, Nq, H * D), rn(B, Nk, H * D), rn(B, Nk, H * D)
        bb = (torch.rand(Nq, Nk, device=DEV) * 2 - 1).to(DT)
        fl = 4.0 * B * H * Nq * Nk * D
        row = []
        for qb in (1, 2):
            K.tuning_set("attn_qb", qb)
            t = timeit(lambda: K.attention(q, k, v, H, bias=bb), iters)
            row.append(f"QB={qb}: {t * 1e3:7.3f} ms {fl / t / 1e12:6.1f} TF/s")
        pbb = K.pack_attn_bias(bb)
        for qb in (1, 2):
            K.tuning_set("attn_qb", qb)
            t = timeit(lambda: K.attention(q, k, v, H, bias=pbb, bias_packed=True), iters)
            row.append(f"packed bias QB={qb}: {t * 1e3:7.3f} ms {fl / t / 1e12:6.1f} TF/s")
        K.tuning_set("attn_qb", 2)
        K.tuning_set("attn_hl", 2)          # A/B: mask rows through the wave's LDS patch instead of per-lane fragments from global memory
        t = timeit(lambda: K.attention(q, k, v, H, bias=pbb, bias_packed=True), iters)
        row.append(f"packed QB=2, mask rows through LDS: {t * 1e3:7.3f} ms {fl / t / 1e12:6.1f} TF/s")
        K.tuning_set("attn_hl", 0)
        K.tuning_set("attn_qb", 0)
        print(f"attn  {name:14s} " + " | ".join(row))
    for name, B, H, Nq, Nk in [("pano L0 self d64", 32, 5, 8192, 8192), ("pers L0 self d64", 640, 5, 1024, 1024)]:
        D = 64
        q, k, v = rn(B, Nq, H * D), rn(B, Nk, H * D), rn(B, Nk, H * D)
        fl = 4.0 * B * H * Nq * Nk * D
        row = []
        for qb in (1, 2):
            K.tuning_set("attn_qb", qb)
            t = timeit(lambda: K.attention(q, k, v, H), iters)
            row.append(f"QB={qb}: {t * 1e3:7.3f} ms {fl / t / 1e12:6.1f} TF/s")
        K.tuning_set("attn_qb", 0)
        print(f"attn  {name:14s} " + " | ".join(row))
    for name, B, H, Nq, grp in [("pers L0 cross", 640, 5, 1024, 16), ("pano L0 cross", 32, 5, 8192, 16), ("pers L1 cross", 640, 10, 256, 16)]:
        D = 64
        q = rn(B, Nq, H * D)
        k1, v1, k2, v2 = rn(B // grp, 77, H * D), rn(B // grp, 77, H * D), rn(B // grp, 64, H * D), rn(B // grp, 64, H * D)

        def two():
            o = K.attention(q, k1, v1, H, kv_group=grp)
            K.attention(q, k2, v2, H, kv_group=grp, out=o, accumulate=True)
            return o
        t2 = timeit(two, iters)
        t1 = timeit(lambda: K.attention2(q, k1, v1, k2, v2, H, kv_group=grp), iters)
        print(f"attn  {name:14s} two launches {t2 * 1e3:7.3f} ms | one launch (attn_fwd2) {t1 * 1e3:7.3f} ms")


def bench_attn_pipe(iters):
    """d = 64 self-attention: attn_fwd_kernel (knob attn_pipe 0) against the software-pipelined kernel, every schedule, four-
    and eight-wave workgroups; alternated on the same tensors."""
    shapes = [("pano L0 self", 32, 5, 8192, 8192), ("pers L0 self", 640, 5, 1024, 1024), ("pano L1 self", 32, 10, 2048, 2048),
              ("pers L1 self", 640, 10, 256, 256), ("pano L2 self", 32, 20, 512, 512), ("pano L3 self", 32, 20, 128, 128)]
    if os.environ.get("IM360_ABL"):
        shapes = shapes[:2]
    knobs = [0, 1, 9, 10, 25, 65, 66, 81] + ([] if not os.environ.get('IM360_ABL') else [1 + 256 * a for a in (1, 2, 3, 4, 5, 11)])
    for name, B, H, Nq, Nk in shapes:
        D = 64
        q, k, v = rn(B, Nq, H * D), rn(B, Nk, H * D), rn(B, Nk, H * D)
        fl = 4.0 * B * H * Nq * Nk * D
        best = {}
        for rnd_ in range(3):
            for kb in knobs:
                K.tuning_set("attn_pipe", kb)
                t = timeit(lambda: K.attention(q, k, v, H), iters)
                best[kb] = min(best.get(kb, 1e9), t)
        K.tuning_set("attn_pipe", K.ATTN_PIPE_DEFAULT)
        print(f"attn_pipe {name:14s} " + " | ".join(f"{kb}: {best[kb] * 1e3:6.3f} ms {fl / best[kb] / 2.5e15 * 100:4.1f}%" for kb in knobs), flush=True)


def bench_attn_self(iters):
    """The two big d = 64 self-attention shapes with whatever the IM360_ATTN_PIPE environment variable selected (for PMC passes)."""
    for name, B, H, Nq, Nk in [("pano L0 self", 32, 5, 8192, 8192), ("pers L0 self", 640, 5, 1024, 1024)]:
        q, k, v = rn(B, Nq, H * 64), rn(B, Nk, H * 64), rn(B, Nk, H * 64)
        t = timeit(lambda: K.attention(q, k, v, H), iters)
        print(f"attn_self {name}: {t * 1e3:.3f} ms  {4.0 * B * H * Nq * Nk * 64 / t / 2.5e15 * 100:.1f}% of MFMA peak")


def bench_conv(iters):
    shapes = [("pers L0 320->320", 640, 32, 32, 320, 320, False), ("pano L0 320->320 (W+4)", 32, 64, 132, 320, 320, False),
              ("pers L1 640->640", 640, 16, 16, 640, 640, False), ("pers L2 1280->1280", 640, 8, 8, 1280, 1280, False),
              ("pers L3 1280->1280", 640, 4, 4, 1280, 1280, False), ("pers up L0 960->320", 640, 32, 32, 960, 320, False),
              ("pers up L1 1920->640", 640, 16, 16, 1920, 640, False), ("pano L0 wrap s1", 32, 64, 128, 320, 320, True)]
    for name, N, H, W, Ci, Co, wrap in shapes:
        x = rn(N, H, W, Ci)
        w = K.pack_conv_weight(rn(Co, Ci, 3, 3) * (9 * Ci) ** -0.5)
        b = rn(Co)
        t = timeit(lambda: K.conv2d(x, w, Co, bias=b, wrap=wrap), iters)
        fl = 2.0 * N * H * W * Ci * Co * 9
        print(f"conv  {name:26s} N={N:3d} {H:3d}x{W:3d} {Ci:4d}->{Co:4d}: {t * 1e3:8.3f} ms  {fl / t / 1e12:7.1f} TF/s "
              f"({fl / t / 2.5e15 * 100:4.1f}% of MFMA peak)")


def bench_conv_small(iters):
    """cfg1-sized convolutions with Cout = 320 (too few tiles for the 256 x 320 tile): 256 x 64 tiles with 32- (knob conv_small 0)
    or 64-channel K steps (1: spills since round 3), or 128 x 128 tiles (2, default)."""
    shapes = [("cfg1 pers L0 320->320", 320, 16, 16, 320, 320), ("cfg1 pers up L0 960->320", 320, 16, 16, 960, 320),
              ("cfg1 pano L0 320->320 (W+4)", 16, 32, 68, 320, 320), ("cfg1 pers up L0 640->320", 320, 16, 16, 640, 320)]
    for name, N, H, W, Ci, Co in shapes:
        x = rn(N, H, W, Ci)
        w = K.pack_conv_weight(rn(Co, Ci, 3, 3) * (9 * Ci) ** -0.5)
        b = rn(Co)
        fl = 2.0 * N * H * W * Ci * Co * 9
        row = []
        for pol in (0, 1, 2, 0):
            K.tuning_set("conv_small", pol)
            t = timeit(lambda: K.conv2d(x, w, Co, bias=b), iters)
            row.append(f"conv_small={pol}: {t * 1e3:7.3f} ms {fl / t / 1e12:6.0f} TF/s")
        K.tuning_set("conv_small", 2)
        print(f"conv_small {name:28s}: " + " | ".join(row))


def bench_up2(iters):
    """Upsample3D convolutions: nearest-x2 folded into the 9-tap kernel's addressing vs four 2x2 convolutions of the
    low-resolution input (sub-pixel form, 4/9 of the MACs).  TF/s are quoted on the 9-tap flop count for both."""
    for name, N, H, W, C, wrap in [("pers L1->L0 640", 640, 16, 16, 640, False), ("pers L2->L1 1280", 640, 8, 8, 1280, False),
                                   ("pers L3->L2 1280", 640, 4, 4, 1280, False), ("pano L1->L0 640", 32, 32, 64, 640, True),
                                   ("pano L2->L1 1280", 32, 16, 32, 1280, True)]:
        x, w, b = rn(N, H, W, C), rn(C, C, 3, 3) * (9 * C) ** -0.5, rn(C)
        wp, w4 = K.pack_conv_weight(w), K.pack_conv_up2_weight(w)
        t9 = timeit(lambda: K.conv2d(x, wp, C, bias=b, up=True, wrap=wrap), iters)
        t4 = timeit(lambda: K.conv_up2(x, w4, C, bias=b, wrap=wrap), iters)
        fl = 2.0 * N * 4 * H * W * C * C * 9
        print(f"up2   {name:18s} N={N:3d} {H}x{W} -> {2 * H}x{2 * W} C={C:4d}: 9-tap {t9 * 1e3:7.3f} ms {fl / t9 / 1e12:6.0f} TF/s | sub-pixel {t4 * 1e3:7.3f} ms "
              f"({fl / t4 / 1e12:6.0f} TF/s equivalent, {t9 / t4:4.2f}x)")


def bench_temporal(iters):
    for name, B, Fr, P, C in [("pers L0", 40, 16, 1024, 320), ("pano L0", 2, 16, 8192, 320), ("pers L1", 40, 16, 256, 640),
                              ("pers L2", 40, 16, 64, 1280), ("cfg4 pers L0 F48", 40, 48, 256, 320), ("cfg4 pano L0 F48", 2, 48, 2048, 320),
                              ("cfg4 pers L1 F48", 40, 48, 64, 640)]:
        qkv = rn(B * Fr * P, 3 * C)
        t = timeit(lambda: K.temporal_attention(qkv, B, Fr, P, 8), iters)
        by = 4.0 * B * Fr * P * C * 2
        K.tuning_set("tattn_scalar", 1)
        ts = timeit(lambda: K.temporal_attention(qkv, B, Fr, P, 8), iters)
        K.tuning_set("tattn_scalar", 0)
        print(f"tattn {name:16s} B={B:3d} F={Fr} P={P:5d} C={C:4d}: {t * 1e3:8.3f} ms  {by / t / 1e9:7.0f} GB/s ({by / t / 8e12 * 100:4.1f}% of HBM peak)"
              f" | scalar kernel {ts * 1e3:8.3f} ms {by / ts / 1e9:7.0f} GB/s")


def bench_ln(iters):
    for name, rows, C in [("pers L0", 655360, 320), ("pano L0", 262144, 320), ("pers L1", 163840, 640), ("pers L2", 40960, 1280)]:
        x, g, b = rn(rows, C), rn(C), rn(C)
        t = timeit(lambda: K.layer_norm(x, g, b), iters)
        by = 2.0 * rows * C * 2
        K.tuning_set("ln_packed", 0)             # A/B: the row-per-wave kernel at C = 320
        t_old = timeit(lambda: K.layer_norm(x, g, b), iters)
        K.tuning_set("ln_packed", 1)
        print(f"ln    {name:10s} rows={rows:7d} C={C:4d}: {t * 1e3:8.3f} ms  {by / t / 1e9:7.0f} GB/s ({by / t / 8e12 * 100:4.1f}% of HBM peak) | row-per-wave kernel {t_old * 1e3:8.3f} ms")
        h = rn(rows, 8 * C) if rows * C * 16 < 8e9 else None
        if h is not None:
            t = timeit(lambda: K.geglu(h), iters)
            by = 3.0 * rows * 4 * C * 2
            print(f"geglu {name:10s} rows={rows:7d} I={4 * C:4d}: {t * 1e3:8.3f} ms  {by / t / 1e9:7.0f} GB/s ({by / t / 8e12 * 100:4.1f}% of HBM peak)")


def bench_gn(iters):
    for name, N, H, W, C, pad in [("pers L0", 640, 32, 32, 320, 0), ("pano L0 pad2", 32, 64, 128, 320, 2), ("pers L1", 640, 16, 16, 640, 0),
                                  ("pers up L0 960", 640, 32, 32, 960, 0)]:
        x, g, b = rn(N, H, W, C), rn(C), rn(C)
        t1 = timeit(lambda: K.group_norm_stats(x, g, b, 32, 1e-5, pad), iters)
        s, h = K.group_norm_stats(x, g, b, 32, 1e-5, pad)
        t2 = timeit(lambda: K.group_norm_apply(x, s, h, True, pad), iters)
        by = N * H * W * C * 2.0
        K.GN_FUSED = True
        t3 = timeit(lambda: K.group_norm(x, g, b, 32, 1e-5, silu=True, pad=pad), iters)
        K.GN_FUSED = False
        print(f"gn    {name:14s} N={N:3d} {H}x{W} C={C:4d}: stats {t1 * 1e3:7.3f} ms {by / t1 / 1e9:6.0f} GB/s | apply {t2 * 1e3:7.3f} ms "
              f"{2 * by / t2 / 1e9:6.0f} GB/s | one launch {t3 * 1e3:7.3f} ms (stats + apply {1e3 * (t1 + t2):7.3f}), {2 * by / t3 / 1e9:6.0f} GB/s algorithmic")


def bench_srpad(iters):
    """SR close-loop pad: algorithmic bytes = output read once + written once."""
    from imagine360_amd import sr_patch
    for name, shape, dt, latent in [("SR video 1024x2048 fp16", (1, 3, 16, 1024, 2048), torch.float16, False),
                                    ("SR video 1024x2048 fp32", (1, 3, 16, 1024, 2048), torch.float32, False),
                                    ("SR latent 128x256 fp16", (1, 4, 32, 128, 256), torch.float16, True)]:
        x = torch.randn(shape, device="cuda").to(dt)
        t = timeit(lambda: sr_patch.padding_pano(x, latent=latent), iters)
        y = sr_patch.padding_pano(x, latent=latent)
        by = 2.0 * y.numel() * y.element_size()
        pw = (y.shape[-1] - x.shape[-1]) // 2
        tt = timeit(lambda: torch.nn.functional.pad(x.flatten(0, 2), [pw, pw], mode="circular"), iters)      # what pad_pano does
        print(f"srpad {name:26s}: {t * 1e3:8.3f} ms  {by / t / 1e9:7.0f} GB/s ({by / t / 8e12 * 100:4.1f}% of HBM peak) | torch F.pad circular {tt * 1e3:8.3f} ms")


def bench_preproc(iters):
    """Preprocessing warps: process_equi of a 16-frame 512 x 1024 panorama into 20 views of 256 x 256 --
    one launch -- against the per-(frame, view) CPU restatement the oracle runs (and the reference's cv2 loop)."""
    import time
    import numpy as np
    from imagine360_amd import preprocess as PP, synthetic as S
    pano = torch.rand(16, 3, 512, 1024) * 2 - 1
    th, ph = S.icosahedron_angles() if hasattr(S, "icosahedron_angles") else (np.linspace(-180, 180, 20), np.zeros(20))
    th, ph = np.rad2deg(np.asarray(th, np.float64)), np.rad2deg(np.asarray(ph, np.float64))
    PP.process_equi(pano, th, ph)                                  # builds + caches the 20 maps (host, float64)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        out = PP.process_equi(pano, th, ph)
    torch.cuda.synchronize()
    t_all = (time.time() - t0) / 3
    frames = ((pano + 1) * 127.5).permute(0, 2, 3, 1).to(torch.uint8).cuda()
    eq = PP.Equirectangular(frames)
    t = timeit(lambda: eq.GetPerspectives(90, th, ph, 256, 256), iters)
    px = 16 * 20 * 256 * 256
    t0 = time.time()
    for _ in range(3):
        PP.process_equi(pano, th, ph, keep_on_device=True)
    torch.cuda.synchronize()
    t_dev = (time.time() - t0) / 3
    print(f"preproc process_equi, result left on the device: {t_dev * 1e3:7.1f} ms per call")
    print(f"preproc process_equi 16 x 512x1024 -> 16 x 20 x 256x256: warp kernel {t * 1e3:7.3f} ms ({px / t / 1e9:6.1f} Gpixel/s, "
          f"{px * 3 / t / 1e9:6.0f} GB/s written); whole call incl. uploads / download {t_all * 1e3:7.1f} ms; output {tuple(out.shape)}")


def bench_linear(iters):
    """torch F.linear (hipBLASLt) vs the implicit-GEMM kernel used as a 1x1 conv on the same shapes."""
    import torch.nn.functional as F
    shapes = [("pers L0 qkv", 655360, 320, 960), ("pers L0 ff-in", 655360, 320, 2560), ("pers L0 ff-out", 655360, 1280, 320),
              ("pers L0 proj", 655360, 320, 320), ("pers L1 qkv", 163840, 640, 1920), ("pers L1 ff-in", 163840, 640, 5120),
              ("pers L1 ff-out", 163840, 2560, 640), ("pers L2 ff-in", 40960, 1280, 10240), ("pers L2 ff-out", 40960, 5120, 1280),
              ("pano L0 ff-in", 262144, 320, 2560)]
    for name, M, Kd, N in shapes:
        x, w, b = rn(M, Kd), rn(N, Kd) * Kd ** -0.5, rn(N)
        t1 = timeit(lambda: F.linear(x, w, b), iters)
        wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
        x4 = x.reshape(M // 64, 8, 8, Kd)
        t2 = timeit(lambda: K.conv2d(x4, wp, N, bias=b), iters)
        fl = 2.0 * M * Kd * N
        print(f"linear {name:16s} M={M:6d} K={Kd:4d} N={N:5d}: torch {t1 * 1e3:7.3f} ms {fl / t1 / 1e12:6.0f} TF/s | im360 1x1 {t2 * 1e3:7.3f} ms {fl / t2 / 1e12:6.0f} TF/s")


def bench_geglu_fused(iters):
    """GEGLU feed-forward input: hipBLASLt projection + elementwise kernel vs the single fused GEMM launch."""
    import torch.nn.functional as F
    for name, M, C in [("pers L0", 655360, 320), ("pano L0", 262144, 320), ("pers L1", 163840, 640), ("pers L2", 40960, 1280)]:
        x, w, b = rn(M, C), rn(8 * C, C) * C ** -0.5, rn(8 * C)
        t1 = timeit(lambda: K.geglu(F.linear(x, w, b)), iters)
        wp, bp = K.pack_geglu(w, b)
        t2 = timeit(lambda: K.linear_geglu(x, wp, bp, 4 * C), iters)
        fl = 2.0 * M * C * 8 * C
        print(f"geglu_fused {name:8s} M={M:6d} C={C:4d}: torch+geglu {t1 * 1e3:7.3f} ms | fused {t2 * 1e3:7.3f} ms {fl / t2 / 1e12:6.0f} TF/s")


def bench_lnfold(iters):
    """Round 3: LayerNorm folded into the consuming GEMM (row statistics from the producer's epilogue) vs the LayerNorm
    kernel + GEMM; the producer with and without the statistics; the skip pair (x, skip) read in place vs torch.cat; the
    cout-grouped tile walk on the level-1 GEGLU projection."""
    import torch.nn as nn
    from imagine360_amd import layers
    for name, M, C in [("pers L0", 655360, 320), ("pano L0", 262144, 320), ("pers L1", 163840, 640), ("pano L1", 65536, 640)]:
        prod, norm = nn.Linear(C, C).to(DEV, DT), nn.LayerNorm(C).to(DEV, DT)
        x, res = rn(M, C), rn(M, C)
        cache = layers.DerivedCache()
        t_p0 = timeit(lambda: layers.gemm_linear(prod.weight, prod.bias, x, res=res, cache=cache), iters)
        t_p1 = timeit(lambda: layers.gemm_linear(prod.weight, prod.bias, x, res=res, cache=cache, row_stats=True), iters)
        y, st = layers.gemm_linear(prod.weight, prod.bias, x, res=res, cache=cache, row_stats=True)
        wq = rn(3 * C, C) * C ** -0.5
        c2 = layers.DerivedCache()
        t_ln = timeit(lambda: layers.layer_norm(norm, y), iters)
        t_u = timeit(lambda: layers.ln_linear(norm, wq, None, y, None, c2, "q"), iters)
        t_f = timeit(lambda: layers.ln_linear(norm, wq, None, y, st, c2, "q"), iters)
        ff = layers.GEGLU(C, 4 * C).to(DEV, DT)
        t_gu = timeit(lambda: ff(y, norm, None), iters)
        t_gf = timeit(lambda: ff(y, norm, st), iters)
        print(f"lnfold {name:8s} M={M:6d} C={C:4d}: out-proj+res {t_p0 * 1e3:6.3f} ms, with row stats {t_p1 * 1e3:6.3f} | LN {t_ln * 1e3:6.3f} | "
              f"LN+qkv {t_u * 1e3:6.3f} -> folded {t_f * 1e3:6.3f} | LN+GEGLU {t_gu * 1e3:6.3f} -> folded {t_gf * 1e3:6.3f}")
    for name, N, H, W, C1, C2, Cout in [("pers up L0", 640, 32, 32, 320, 320, 320), ("pers up L0", 640, 32, 32, 640, 320, 320),
                                         ("pers up L1", 640, 16, 16, 1280, 640, 640), ("pano up L0", 32, 64, 128, 320, 320, 320)]:
        xa, xb = rn(N, H, W, C1), rn(N, H, W, C2)
        g, b = rn(C1 + C2), rn(C1 + C2)
        wp = K.pack_conv_weight(rn(Cout, C1 + C2, 1, 1) * (C1 + C2) ** -0.5)

        def old():
            c = torch.cat([xa, xb], dim=-1)
            h = K.group_norm(c, g, b, 32, 1e-5, silu=True)
            return h, K.conv2d(c, wp, Cout)

        def new():
            h = K.group_norm((xa, xb), g, b, 32, 1e-5, silu=True)
            return h, K.conv1x1_cat(xa, xb, wp, Cout)

        t_o, t_n = timeit(old, iters), timeit(new, iters)
        print(f"skip   {name:10s} N={N:3d} {H}x{W} {C1}+{C2}->{Cout}: cat + GN + shortcut {t_o * 1e3:6.3f} ms | in place {t_n * 1e3:6.3f} ms")
    x, w, b = rn(163840, 640), rn(5120, 640) * 640 ** -0.5, rn(5120)
    wp, bp = K.pack_geglu(w, b)
    for ng in (1, 2, 4):
        K.tuning_set("ring_groups", ng)
        t = timeit(lambda: K.linear_geglu(x, wp, bp, 2560), iters)
        print(f"groups pers L1 GEGLU 640->5120, {ng} cout group(s): {t * 1e3:6.3f} ms {2.0 * 163840 * 640 * 5120 / t / 1e12:6.0f} TF/s")
    K.tuning_set("ring_groups", 0)


def bench_attn_ds(iters):
    """Knob attn_ds (row sums as dot2's of the packed weights, half-wave exchange through v_permlane32_swap) on the step's
    self-attention and WarpAttn shapes: plain vs DS."""
    shapes = [("pano L0 self", 32, 5, 8192, 8192, 64, False), ("pers L0 self", 640, 5, 1024, 1024, 64, False),
              ("pano L1 self", 32, 10, 2048, 2048, 64, False), ("pers L1 self", 640, 10, 256, 256, 64, False),
              ("pano L2 self", 32, 20, 512, 512, 64, False), ("pers L2 self", 640, 20, 64, 64, 64, False),
              ("warp L0 e2p", 32, 5, 8192, 20480, 32, True), ("warp L1 e2p", 32, 10, 2048, 5120, 32, True),
              ("warp L1 p2e", 32, 10, 5120, 2048, 32, True), ("warp L2 e2p", 32, 20, 512, 1280, 32, True)]
    for name, B, H, Nq, Nk, D, bias in shapes:
        q, k, v = rn(B, Nq, H * D), rn(B, Nk, H * D), rn(B, Nk, H * D)
        bb = K.pack_attn_bias((torch.rand(Nq, Nk, device=DEV) * 2 - 1).to(DT)) if bias else None
        fl = 4.0 * B * H * Nq * Nk * D
        row = []
        for ds in (0, 1, 0, 1):
            K.tuning_set("attn_ds", ds)
            t = timeit(lambda: K.attention(q, k, v, H, bias=bb, bias_packed=bias), iters)
            row.append(f"ds={ds}: {t * 1e3:7.3f} ms {fl / t / 2.5e15 * 100:4.1f}%")
        K.tuning_set("attn_ds", 0)
        print(f"attn_ds {name:14s} " + " | ".join(row))


def bench_attn_small(iters):
    """The step's small attention launches (levels 2 / 3): latency / occupancy bound, far from either roofline."""
    shapes = [("pers L2 self", 640, 20, 64, 64, 64, False), ("pers L3 self", 640, 20, 16, 16, 64, False),
              ("pano L2 self", 32, 20, 512, 512, 64, False), ("pano L3 self", 32, 20, 128, 128, 64, False),
              ("warp L2 e2p", 32, 20, 512, 1280, 32, True), ("warp L2 p2e", 32, 20, 1280, 512, 32, True),
              ("warp L3 e2p", 32, 40, 128, 320, 32, True), ("warp L3 p2e", 32, 40, 320, 128, 32, True)]
    for name, B, H, Nq, Nk, D, bias in shapes:
        q, k, v = rn(B, Nq, H * D), rn(B, Nk, H * D), rn(B, Nk, H * D)
        bb = K.pack_attn_bias((torch.rand(Nq, Nk, device=DEV) * 2 - 1).to(DT)) if bias else None
        row = []
        for one in (0, 1):
            K.tuning_set("attn_one", one)
            t = timeit(lambda: K.attention(q, k, v, H, bias=bb, bias_packed=bias), iters)
            fl = 4.0 * B * H * Nq * Nk * D
            by = 2.0 * (2 * B * Nq + 2 * B * Nk) * H * D
            row.append(f"attn_one={one}: {t * 1e3:8.3f} ms  {fl / t / 1e12:7.1f} TF/s  {by / t / 1e9:6.0f} GB/s")
        K.tuning_set("attn_one", 1)
        print(f"attn_small {name:14s} B={B:4d} H={H:2d} Nq={Nq:5d} Nk={Nk:5d} d={D}: " + " | ".join(row))
    for name, B, H, Nq, grp in [("pers L3 cross", 640, 20, 16, 16), ("pano L3 cross", 32, 20, 128, 16)]:
        q = rn(B, Nq, H * 64)
        k1, v1, k2, v2 = rn(B // grp, 77, H * 64), rn(B // grp, 77, H * 64), rn(B // grp, 64, H * 64), rn(B // grp, 64, H * 64)
        t = timeit(lambda: K.attention2(q, k1, v1, k2, v2, H, kv_group=grp), iters)
        print(f"attn_small {name:14s} B={B:4d} H={H:2d} Nq={Nq:5d}: {t * 1e3:8.3f} ms")


def bench_attn_ablate(iters):
    """Where the time of the d = 64 two-query-block flash kernel goes (knob attn_dbg; ablation builds, results are garbage):
    full | no exp2 | no QK^T MFMAs | no PV MFMAs | no MFMAs | no MFMAs, no exp2 | no K/V staging | skeleton only."""
    for name, B, H, Nq, Nk in [("pano L0 self", 32, 5, 8192, 8192), ("pers L0 self", 640, 5, 1024, 1024)]:
        q, k, v = rn(B, Nq, H * 64), rn(B, Nk, H * 64), rn(B, Nk, H * 64)
        row = []
        for dbg in (0, 1, 2, 4, 6, 7, 8, 15, 0):
            K.tuning_set("attn_dbg", dbg)
            t = timeit(lambda: K.attention(q, k, v, H), iters)
            row.append(f"dbg={dbg}: {t * 1e3:6.3f}")
        K.tuning_set("attn_dbg", 0)
        print(f"attn_ablate {name:14s} (ms) " + " | ".join(row))


def bench_attn_w3(iters):
    """d = 64 self-attention: the default rule (two query blocks per wave on large grids) vs one block per wave at two and at
    three waves per SIMD (knobs attn_qb 1, attn_w3 1)."""
    shapes = [("pano L0 self", 32, 5, 8192, 8192), ("pers L0 self", 640, 5, 1024, 1024), ("pano L1 self", 32, 10, 2048, 2048),
              ("pers L1 self", 640, 10, 256, 256), ("pano L2 self", 32, 20, 512, 512)]
    for name, B, H, Nq, Nk in shapes:
        q, k, v = rn(B, Nq, H * 64), rn(B, Nk, H * 64), rn(B, Nk, H * 64)
        fl = 4.0 * B * H * Nq * Nk * 64
        row = []
        for qb, w3 in ((0, 0), (1, 0), (1, 1), (0, 0), (1, 1)):
            K.tuning_set("attn_qb", qb)
            K.tuning_set("attn_w3", w3)
            t = timeit(lambda: K.attention(q, k, v, H), iters)
            row.append(f"qb={qb} w3={w3}: {t * 1e3:7.3f} ms {fl / t / 2.5e15 * 100:4.1f}%")
        K.tuning_set("attn_qb", 0)
        K.tuning_set("attn_w3", 0)
        print(f"attn_w3 {name:14s} " + " | ".join(row))


def bench_xattn(iters):
    """Text + IP cross attention (77 + 64 keys, one context per 16-frame video): generic two-pass kernel (knob attn_x 0) vs
    both key / value sets resident in LDS (1: 16-byte stores, 2: 8-byte stores).  Floor = Q read + O written at HBM speed."""
    for name, B, H, Nq, grp in [("pers L0", 640, 5, 1024, 16), ("pano L0", 32, 5, 8192, 16), ("pers L1", 640, 10, 256, 16),
                                ("pano L1", 32, 10, 2048, 16), ("pers L2", 640, 20, 64, 16), ("pano L2", 32, 20, 512, 16)]:
        D = 64
        q = rn(B, Nq, H * D)
        k1, v1, k2, v2 = rn(B // grp, 77, H * D), rn(B // grp, 77, H * D), rn(B // grp, 64, H * D), rn(B // grp, 64, H * D)
        fl = 4.0 * B * H * Nq * 141 * D
        by = 2.0 * 2 * B * Nq * H * D
        row, outs = [], []
        for x in (0, 1, 2, 3):
            K.tuning_set("attn_x", x)
            t = timeit(lambda: K.attention2(q, k1, v1, k2, v2, H, kv_group=grp), iters)
            outs.append(K.attention2(q, k1, v1, k2, v2, H, kv_group=grp).float())
            row.append(f"attn_x={x}: {t * 1e3:7.3f} ms {fl / t / 1e12:6.1f} TF/s {by / t / 1e9:6.0f} GB/s")
        K.tuning_set("attn_x", 3)
        d = ((outs[1] - outs[0]).norm() / outs[0].norm()).item()
        print(f"xattn {name:8s} M={B * Nq:7d} H={H:2d}: " + " | ".join(row) + f" | rel diff vs generic {d:.2e}, wide == narrow stores: {bool((outs[1] == outs[2]).all())}, ring == registers: {bool((outs[1] == outs[3]).all())}")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    iters = 10
    if "--iters" in sys.argv:
        iters = int(sys.argv[sys.argv.index("--iters") + 1])
        args = [a for a in args if a != str(iters)]
    which = args or ["attn", "conv", "up2", "temporal", "ln", "gn", "linear", "geglu_fused"]
    torch.set_grad_enabled(False)
    for w in which:
        globals()["bench_" + w](iters)
