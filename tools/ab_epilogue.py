#!/usr/bin/env python
"""What the ring kernel's epilogue costs per tile: conv_dbg bit 8 skips it (results are garbage), optional bits of experiment builds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit, rn  # noqa: E402

bits = [int(b) for b in sys.argv[1:]] or [0, 8]
for name, M, Kd, N, use_res in [("pers L0 proj+res", 655360, 320, 320, True), ("pers L0 proj", 655360, 320, 320, False), ("pers L0 qkv", 655360, 320, 960, False),
                                ("pers L1 ff-out+res", 163840, 2560, 640, True), ("pers L2 ff-in", 40960, 1280, 10240, False)]:
    x, w, b, r = rn(M, Kd), rn(N, Kd) * Kd ** -0.5, rn(N), rn(M, N) if use_res else None
    wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
    tiles = (M // 256) * (N // 320) / 256
    row = []
    for d in bits:
        K.tuning_set("conv_dbg", d)
        t = timeit(lambda: K.linear(x, wp, N, bias=b, res=r), 10)
        row.append(f"dbg {d}: {t * 1e3:6.3f} ms ({t * 1e6 / tiles:5.1f} us/tile)")
    K.tuning_set("conv_dbg", 0)
    print(f"{name:20s} {tiles:5.0f} tiles/CU  " + " | ".join(row), flush=True)
    del x, w, b, r, wp
