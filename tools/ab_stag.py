#!/usr/bin/env python
"""A/B: ring GEMM pipeline variants (knob conv_ring) on the token-major GEMM shapes of cfg2: time + bit-identity.
    python tools/ab_stag.py [--iters N] [--conv]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit, rn  # noqa: E402

iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
VARIANTS = [1, 10, 11]      # 1 = default (staggered 64-channel stages), 10 = two activation stages in flight (gemm_a3_kernel), 11 = the same
                            # with the weight operand chunk-major (whole-line half stages; round 5); 8 = round 3 ring
if "--variants" in sys.argv:
    VARIANTS = [int(v) for v in sys.argv[sys.argv.index("--variants") + 1].split(",")]


def ab(name, make, fl, wp):
    """make(weight) -> the launch closure; variant 11 gets the chunk-major repack of the weight operand (kernels.chunk_major)."""
    outs, row = [], []
    wcm = K.chunk_major(wp) if 11 in VARIANTS else None
    for v in VARIANTS:
        K.tuning_set("conv_ring", v if v != 6 or not name.startswith("conv") else 7)
        fn = make(wcm if v == 11 else wp)
        y = fn()
        y = y[0] if isinstance(y, tuple) else y
        outs.append(y.clone())
        t = timeit(fn, iters)
        row.append(f"v{v}: {t * 1e3:7.3f} ms {fl / t / 1e12:6.0f} TF/s")
    K.tuning_set("conv_ring", 1)
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{name:34s} " + " | ".join(row) + f" | identical {same}", flush=True)


shapes = [("pers L0 qkv", 655360, 320, 960), ("pers L0 ff-out", 655360, 1280, 320), ("pers L0 proj", 655360, 320, 320),
          ("pers L1 qkv", 163840, 640, 1920), ("pers L1 ff-in", 163840, 640, 5120), ("pers L1 ff-out", 163840, 2560, 640),
          ("pers L2 ff-in", 40960, 1280, 10240), ("pers L2 ff-out", 40960, 5120, 1280), ("pano L0 ff-in", 262144, 320, 2560)]
for name, M, Kd, N in ([] if ("--only-ablate" in sys.argv or "--only" in sys.argv) else shapes):
    x, w, b, r = rn(M, Kd), rn(N, Kd) * Kd ** -0.5, rn(N), rn(M, N)
    wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
    fl = 2.0 * M * Kd * N
    t0 = timeit(lambda: F.linear(x, w, b), iters)
    print(f"{name:34s} hipBLASLt {t0 * 1e3:7.3f} ms {fl / t0 / 1e12:6.0f} TF/s")
    ab("linear " + name, lambda wq: (lambda: K.linear(x, wq, N, bias=b, res=r)), fl, wp)
    ab("linear+rowstats " + name, lambda wq: (lambda: K.linear(x, wq, N, bias=b, res=r, row_stats=True)), fl, wp)
    ab("linear nores " + name, lambda wq: (lambda: K.linear(x, wq, N, bias=b)), fl, wp)
    del x, w, b, r, wp
for name, M, C in ([] if ("--only-ablate" in sys.argv or "--only" in sys.argv) else [("pers L0", 655360, 320), ("pano L0", 262144, 320), ("pers L1", 163840, 640), ("pers L2", 40960, 1280)]):
    x, w, b = rn(M, C), rn(8 * C, C) * C ** -0.5, rn(8 * C)
    wp, bp = K.pack_geglu(w, b)
    ab("geglu " + name, lambda wq: (lambda: K.linear_geglu(x, wq, bp, 4 * C)), 2.0 * M * C * 8 * C, wp)
    del x, w, b, wp, bp
if "--conv" in sys.argv:
    for name, N, H, W, Cin, Cout, kw in [("pers L0 320->320", 640, 32, 32, 320, 320, {}), ("pers L0 640->320", 640, 32, 32, 640, 320, {}),
                                         ("pers L0 960->320", 640, 32, 32, 960, 320, {}), ("pers L1 640->640", 640, 16, 16, 640, 640, {}),
                                         ("pers L1 1920->640", 640, 16, 16, 1920, 640, {}), ("pano L0 320->320 (W+4)", 32, 64, 132, 320, 320, dict(x_off=2, wout=128)),
                                         ("pano L0 wrap", 32, 64, 128, 320, 320, dict(wrap=True)), ("pers L0 stride 2", 640, 32, 32, 320, 320, dict(stride=2)),
                                         ("pers L0 320->320 gn", 640, 32, 32, 320, 320, dict(gn_stats=True))]:
        x, w, b = rn(N, H, W, Cin), rn(Cout, Cin, 3, 3) * (9 * Cin) ** -0.5, rn(Cout)
        wp = K.pack_conv_weight(w)
        fl = 2.0 * N * H * W * Cin * Cout * 9 / (kw.get("stride", 1) ** 2)
        rowtxt, ys = [], []
        for v in (0, 1):
            K.tuning_set("conv_persist" if "--persist" in sys.argv else "conv_stag", v)
            y = K.conv2d(x, wp, Cout, bias=b, **kw)
            g = K._gn_of(y)
            ys.append((y.clone(), None if g is None else g[0].clone()))
            t = timeit(lambda: K.conv2d(x, wp, Cout, bias=b, **kw), iters)
            rowtxt.append(f"stag {v}: {t * 1e3:7.3f} ms {fl / t / 1e12:6.0f} TF/s")
        K.tuning_set("conv_stag", 0)
        K.tuning_set("conv_persist", 0)
        same = torch.equal(ys[0][0], ys[1][0]) and (ys[0][1] is None or torch.equal(ys[0][1], ys[1][1]))
        print(f"conv {name:24s} " + " | ".join(rowtxt) + f" | identical {same}", flush=True)
if "--ablate" in sys.argv:
    # ablation builds of the staggered loop (make ablate): conv_dbg bits 1 = no LDS-DMA, 2 = no MFMA, 4 = no fragment reads
    assert K.ablate_build(), "needs `make -C imagine360_amd/csrc ablate`"
    for name, M, Kd, N in [("pers L0 ff-out", 655360, 1280, 320), ("pers L1 ff-out", 163840, 2560, 640), ("pers L2 ff-in", 40960, 1280, 10240),
                           ("pers L0 proj", 655360, 320, 320)]:
        x, w, b = rn(M, Kd), rn(N, Kd) * Kd ** -0.5, rn(N)
        wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
        fl = 2.0 * M * Kd * N
        K.tuning_set("conv_ring", 6)
        row = []
        for bits, label in [(0, "full"), (1, "no DMA"), (2, "no MFMA"), (4, "no reads"), (3, "reads only"), (6, "DMA only"), (5, "MFMA only"), (7, "skeleton")]:
            K.tuning_set("conv_dbg", bits)
            t = timeit(lambda: K.linear(x, wp, N, bias=b), iters)
            row.append(f"{label} {t * 1e3:6.3f}")
        K.tuning_set("conv_dbg", 0)
        K.tuning_set("conv_ring", 1)
        print(f"ablate {name:16s} (ideal MFMA {fl / 2.5e15 * 1e3:5.3f} ms): " + " | ".join(row), flush=True)
