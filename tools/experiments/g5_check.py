#!/usr/bin/env python
"""The 256 x 320 four-wave register-staged tile (conv3x3_g5.hip, knob conv_ring 14) against the default kernels: bit identity and time,
token-major Linears (Cout % 320 == 0) and 3 x 3 convolutions.   python tools/g5_check.py [--conv]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit, rn  # noqa: E402


def ab(name, fn, fl, iters=10, rounds=3):
    outs, best = {}, {}
    for v in (1, 14):
        K.tuning_set("conv_ring", v)
        y = fn()
        outs[v] = (y[0] if isinstance(y, tuple) else y).clone()
        g = K._gn_of(y) if not isinstance(y, tuple) else None
        outs[(v, "gn")] = None if g is None else g[0].clone()
        timeit(fn, 3)
        best[v] = float("inf")
    for _ in range(rounds):
        for v in (1, 14):
            K.tuning_set("conv_ring", v)
            best[v] = min(best[v], timeit(fn, iters))
    K.tuning_set("conv_ring", 1)
    same = torch.equal(outs[1], outs[14]) and (outs[(1, "gn")] is None or torch.equal(outs[(1, "gn")], outs[(14, "gn")]))
    rel = ((outs[1].float() - outs[14].float()).norm() / outs[1].float().norm()).item()
    print(f"{name:34s} default {best[1] * 1e3:6.3f} ms {fl / best[1] / 1e12:5.0f} TF/s | g5 {best[14] * 1e3:6.3f} ms {fl / best[14] / 1e12:5.0f} TF/s | x{best[1] / best[14]:.3f} | identical {same} (rel {rel:.1e})", flush=True)


for name, M, Kd, N in [("pers L0 ff-out", 655360, 1280, 320), ("pers L1 ff-out", 163840, 2560, 640), ("pers L1 out-proj", 163840, 640, 640),
                       ("pers L2 ff-in", 40960, 1280, 10240), ("pers L2 ff-out", 40960, 5120, 1280), ("pers L0 proj", 655360, 320, 320)]:
    x, w, b, r = rn(M, Kd), rn(N, Kd) * Kd ** -0.5, rn(N), rn(M, N)
    wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
    fl = 2.0 * M * Kd * N
    ab("linear nores " + name, lambda: K.linear(x, wp, N, bias=b), fl)
    ab("linear+res " + name, lambda: K.linear(x, wp, N, bias=b, res=r), fl)
    del x, w, b, r, wp
if "--conv" in sys.argv:
    for name, N, H, W, Cin, Cout, kw in [("pers L0 320->320", 640, 32, 32, 320, 320, {}), ("pers L0 320->320 gn", 640, 32, 32, 320, 320, dict(gn_stats=True)),
                                         ("pers L0 640->320", 640, 32, 32, 640, 320, {}), ("pers L0 960->320 gn", 640, 32, 32, 960, 320, dict(gn_stats=True)),
                                         ("pers L1 640->640", 640, 16, 16, 640, 640, {}), ("pers L1 1920->640", 640, 16, 16, 1920, 640, {}),
                                         ("pers L2 1280->1280", 640, 8, 8, 1280, 1280, {})]:
        x, w, b = rn(N, H, W, Cin), rn(Cout, Cin, 3, 3) * (9 * Cin) ** -0.5, rn(Cout)
        t, r = rn(N // 20, Cout), rn(N, H, W, Cout)
        wp = K.pack_conv_weight(w)
        fl = 2.0 * N * H * W * Cin * Cout * 9
        ab("conv " + name, lambda: K.conv2d(x, wp, Cout, bias=b, **kw), fl)
        ab("conv+temb+res " + name, lambda: K.conv2d(x, wp, Cout, bias=b, temb=t, imgs_per_temb=20, res=r, **kw), fl)
        del x, w, b, wp, t, r
