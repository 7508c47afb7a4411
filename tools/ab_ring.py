#!/usr/bin/env python
"""(Round 2 / 3 tool; since round 4 knob conv_ring 1 selects the staggered 64-channel-stage loop for the token-major GEMMs and 8 the
ring with interleaved requests this tool was written around -- see tools/ab_stag.py for the current A/B.)
A/B of the conv / GEMM pipelines on the MI355X: two-stage kernel (knob conv_ring = 0) vs the persistent ring kernel
(1: asm LDS-DMA, 2: builtin LDS-DMA).  Checks that all variants return IDENTICAL tensors (same accumulation order) and
prints their times at cfg2 shapes.

    python tools/ab_ring.py [--iters N] [--no-check]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402

DT = torch.bfloat16
DEV = "cuda"
VARIANTS = [0, -4]          # conv_ring values; -2 = two-stage kernel on 128 x 320 tiles; -7 = halo-patch conv kernel (conv_ring 1)


def rn(*s, scale=1.0):
    return (torch.randn(*s, device=DEV, dtype=torch.float32) * scale).to(DT)


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
    return a.elapsed_time(b) / iters


def run_variants(name, fn, flops, bytes_, iters, check):
    outs, times = [], []
    for v in VARIANTS:
        K.tuning_set("conv_ring", 1 if v == -7 else max(v, 0))
        K.tuning_set("conv_big", 2 if v == -2 else 3 if v == -3 else 4 if v == -4 else 1)
        K.tuning_set("conv_halo", 1 if v == -7 else 0)
        K.tuning_set("conv_cm", 0 if v == -8 else 1)       # -8: two-stage kernel, tap-major K order (round-2 start)
        if check:
            outs.append(fn().clone())
        times.append(timeit(fn, iters))
    K.tuning_set("conv_ring", 1)
    K.tuning_set("conv_big", 1)
    K.tuning_set("conv_halo", 0)
    K.tuning_set("conv_cm", 1)
    same = ""
    if check:
        same = " identical" if all(torch.equal(outs[0], o) for o in outs[1:]) else " max|diff| " + " ".join(
            f"{(o.float() - outs[0].float()).abs().max().item():.3e}" for o in outs[1:])
        assert torch.isfinite(outs[0].float()).all()
    best = min(times)
    print(f"{name:34s} " + " ".join(f"v{v}={t:7.3f}ms" for v, t in zip(VARIANTS, times))
          + f" | best {flops / best / 1e9:6.0f} TF/s {bytes_ / best / 1e9:5.2f} TB/s{same}", flush=True)


def ablate(iters):
    """Where the time of the ring kernel (variant 3: no stagger) goes: switch off parts of it (results are garbage)."""
    torch.manual_seed(0)
    cases = []
    x, w = rn(655360, 1, 1, 320), K.pack_conv_weight(rn(320, 320, 1, 1, scale=320 ** -0.5))
    b, r = rn(320), rn(655360, 1, 1, 320)
    cases.append(("lin L0 out 320>320 +b+res", lambda: K.conv2d(x, w, 320, bias=b, res=r)))
    x2, w2 = rn(163840, 1, 1, 2560), K.pack_conv_weight(rn(640, 2560, 1, 1, scale=2560 ** -0.5))
    cases.append(("lin L1 ffout 2560>640", lambda: K.conv2d(x2, w2, 640)))
    x3, w3 = rn(640, 32, 32, 320), K.pack_conv_weight(rn(320, 320, 3, 3, scale=2880 ** -0.5))
    cases.append(("conv pers L0 320>320", lambda: K.conv2d(x3, w3, 320, bias=b)))
    x4, w4 = rn(640, 16, 16, 640), K.pack_conv_weight(rn(640, 640, 3, 3, scale=5760 ** -0.5))
    cases.append(("conv pers L1 640>640", lambda: K.conv2d(x4, w4, 640)))
    # the two-stage kernel (the default for convolutions), taps innermost (conv_cm 1, default) and tap-major
    for cm, big in ((1, 1), (0, 1), (1, 4)):            # big 4: the same tile on four waves (192 x 320, one wave per SIMD)
        K.tuning_set("conv_ring", 0)
        K.tuning_set("conv_cm", cm)
        K.tuning_set("conv_big", big)
        for name, fn in cases[2:]:
            row = []
            for dbg, lab in [(0, "full"), (1, "noDMA"), (2, "noMFMA"), (3, "noDMA+noMFMA")]:
                K.tuning_set("conv_dbg", dbg)
                row.append(f"{lab}={timeit(fn, iters):.3f}")
            K.tuning_set("conv_dbg", 0)
            print(f"two-stage cm={cm} big={big} {name:28s} " + " ".join(row), flush=True)
    K.tuning_set("conv_cm", 1)
    K.tuning_set("conv_big", 1)
    K.tuning_set("conv_ring", 3)
    for name, fn in cases:
        row = []
        for dbg, lab in [(0, "full"), (1, "noDMA"), (2, "noMFMA"), (4, "noREAD"), (8, "noEPI"), (5, "noDMA+noREAD"),
                         (3, "noDMA+noMFMA"), (6, "noMFMA+noREAD"), (7, "barriers only"), (15, "nothing")]:
            K.tuning_set("conv_dbg", dbg)
            row.append(f"{lab}={timeit(fn, iters):.3f}")
        K.tuning_set("conv_dbg", 0)
        print(f"{name:28s} " + " ".join(row), flush=True)
    K.tuning_set("conv_ring", 1)


def main():
    if "--ablate" in sys.argv:
        torch.set_grad_enabled(False)
        return ablate(10)
    iters = 10
    if "--iters" in sys.argv:
        iters = int(sys.argv[sys.argv.index("--iters") + 1])
    check = "--no-check" not in sys.argv
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    # ---- token-major linears (the conv kernel on a [M, 1, 1, K] view)
    for name, M, Kd, N, has_b, has_r in [("L0 proj 320>320 +b", 655360, 320, 320, True, False),
                                         ("L0 out 320>320 +b+res", 655360, 320, 320, True, True),
                                         ("L0 qkv 320>960", 655360, 320, 960, False, False),
                                         ("L0 ffout 1280>320 +b+res", 655360, 1280, 320, True, True),
                                         ("pano L0 out 320>320 +b+res", 262144, 320, 320, True, True),
                                         ("L1 out 640>640 +b+res", 163840, 640, 640, True, True),
                                         ("L1 qkv 640>1920", 163840, 640, 1920, False, False),
                                         ("L1 ffout 2560>640 +b+res", 163840, 2560, 640, True, True),
                                         ("ragged M 131073 320>320 +res", 131073, 320, 320, True, True)]:
        x, w = rn(M, 1, 1, Kd), rn(N, Kd, 1, 1, scale=Kd ** -0.5)
        wp = K.pack_conv_weight(w)
        b = rn(N) if has_b else None
        r = rn(M, 1, 1, N) if has_r else None
        by = 2.0 * (M * Kd + M * N * (2 if has_r else 1))
        run_variants("lin " + name, lambda: K.conv2d(x, wp, N, bias=b, res=r), 2.0 * M * Kd * N, by, iters, check)
        if check and M == 131073:        # fp32 reference on the ragged tail rows
            ref = x[-300:, 0, 0].float() @ w[:, :, 0, 0].float().t() + b.float()
            ref = ref.to(DT).float() + r[-300:, 0, 0].float()
            got = K.conv2d(x, wp, N, bias=b, res=r)[-300:, 0, 0].float()
            print("    ragged tail rel err", ((got - ref).norm() / ref.norm()).item(), flush=True)
        del x, w, wp, r
    # ---- fused GEGLU
    for name, M, C in [("L0 pers", 655360, 320), ("L0 pano", 262144, 320), ("L1 pers", 163840, 640)]:
        x, w, b = rn(M, C), rn(8 * C, C, scale=C ** -0.5), rn(8 * C)
        wp, bp = K.pack_geglu(w, b)
        run_variants("geglu " + name, lambda: K.linear_geglu(x, wp, bp, 4 * C), 2.0 * M * C * 8 * C,
                     2.0 * (M * C + M * 4 * C), iters, check)
        del x, w, wp
    # ---- convolutions
    for name, N, H, W, Ci, Co, kw in [("pers L0 320>320 +temb+res", 640, 32, 32, 320, 320, dict(temb=True, res=True)),
                                      ("pano L0 wrap", 32, 64, 128, 320, 320, dict(wrap=True)),
                                      ("pano L0 conv2 x_off (W+4)", 32, 64, 132, 320, 320, dict(x_off=2, wout=128, res=True)),
                                      ("pers L1 640>640", 640, 16, 16, 640, 640, dict(temb=True)),
                                      ("pers L2 1280>1280", 640, 8, 8, 1280, 1280, dict()),
                                      ("pers up L0 960>320", 640, 32, 32, 960, 320, dict(temb=True)),
                                      ("pers down s2 320>320", 640, 32, 32, 320, 320, dict(stride=2)),
                                      ("pers upsample 640>640", 640, 16, 16, 640, 640, dict(up=True)),
                                      ("1x1 shortcut 960>320", 640, 32, 32, 960, 320, dict(taps=1)),
                                      ("ragged 145x30x31 320>320", 145, 30, 31, 320, 320, dict(res=True))]:
        taps = kw.pop("taps", 9)
        x = rn(N, H, W, Ci)
        kk = 3 if taps == 9 else 1
        w = K.pack_conv_weight(rn(Co, Ci, kk, kk, scale=(taps * Ci) ** -0.5))
        b = rn(Co)
        stride, up = kw.get("stride", 1), kw.get("up", False)
        ho, wo = (2 * H if up else H) // stride, kw.get("wout", (2 * W if up else W) // stride)
        temb = rn(N // 16, Co) if kw.pop("temb", False) else None
        res = rn(N, ho, wo, Co) if kw.pop("res", False) else None
        fl = 2.0 * N * ho * wo * Ci * Co * taps
        by = 2.0 * (N * H * W * Ci + N * ho * wo * Co * (2 if res is not None else 1))
        run_variants("conv " + name, lambda: K.conv2d(x, w, Co, bias=b, temb=temb, imgs_per_temb=16, res=res, **kw), fl, by, iters, check)
        del x, w, res


if __name__ == "__main__":
    main()
