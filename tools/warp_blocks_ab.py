#!/usr/bin/env python
"""WarpAttn attention with the real cross-view masks of cfg2 (levels 0 - 2, both directions): the shifted packed masks with and
without their block maps (kernels.attn_bias_blocks) -- time, share of non-background blocks, bit identity.
    python tools/warp_blocks_ab.py [--iters N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K, synthetic as S  # noqa: E402
from imagine360_amd.mv_model import WarpAttn  # noqa: E402
from tools.bench_kernels import timeit, rn  # noqa: E402

iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
dev = torch.device("cuda", 0)
for lvl, (ph, eh, C) in enumerate([(32, 64, 320), (16, 32, 640), (8, 16, 1280)]):
    cams = {k: v[0] for k, v in S.icosahedron_cameras(90, ph * 8).items()}
    blk = WarpAttn(C).to(dev)
    b_e2p, b_p2e, _, _, packed = blk.geometry(ph, ph, eh, 2 * eh, cams, False, dev, torch.bfloat16)
    x = blk.geometry_extra(ph, ph, eh, 2 * eh, cams, False, dev, torch.bfloat16)
    H, B = C // 32, 32
    for name, bias, bm in (("e2p", b_e2p, x["blocks_e2p"]), ("p2e", b_p2e, x["blocks_p2e"])):
        Nq, Nk = bias.shape
        q, k, v = rn(B, Nq, C), rn(B, Nk, C), rn(B, Nk, C)
        words = bm.to(torch.int64) & 0xffffffff
        share = float(((words[:, :, None] >> torch.arange(32, device=dev)) & 1).float().sum() / ((Nq // 32) * (Nk // 32)))
        f0 = lambda: K.attention(q, k, v, H, bias=bias, bias_packed=True)
        f1 = lambda: K.attention(q, k, v, H, bias=bias, bias_packed=True, bias_blocks=bm)
        same = torch.equal(f0(), f1())
        best = [float("inf")] * 2
        for _ in range(3):
            for i, f in enumerate((f0, f1)):
                best[i] = min(best[i], timeit(f, iters))
        fl = 4.0 * B * H * Nq * Nk * 32
        print(f"warp L{lvl} {name} Nq={Nq:5d} Nk={Nk:5d} H={H:2d}: no map {best[0] * 1e3:6.3f} ms {fl / best[0] / 2.5e15 * 100:4.1f} % | block map {best[1] * 1e3:6.3f} ms "
              f"{fl / best[1] / 2.5e15 * 100:4.1f} % of the MFMA peak | x{best[0] / best[1]:.3f} | non-background blocks {share * 100:4.1f} % | identical {same}", flush=True)
        del q, k, v
