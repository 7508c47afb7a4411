#!/usr/bin/env python
"""Is WarpAttn limited by its mask traffic?  The same launches with the real [Nq, Nk] mask and with ONE mask row broadcast
through a zero row stride (every fragment load hits the L1 / L2); and the head-group form (knob attn_hg)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from imagine360_amd import kernels as K
from bench_kernels import timeit, rn, DT, DEV
torch.set_grad_enabled(False)
for name, B, H, Nq, Nk in [("warp L1 e2p", 32, 10, 2048, 5120), ("warp L1 p2e", 32, 10, 5120, 2048), ("warp L2 e2p", 32, 20, 512, 1280), ("warp L2 p2e", 32, 20, 1280, 512)]:
    q, k, v = rn(B, Nq, H * 32), rn(B, Nk, H * 32), rn(B, Nk, H * 32)
    bb = K.pack_attn_bias((torch.rand(Nq, Nk, device=DEV) * 2 - 1).to(DT))
    b0 = K.pack_attn_bias((torch.rand(1, Nk, device=DEV) * 2 - 1).to(DT)).expand(Nq, Nk)
    row = []
    for hg in (0, 1, 0, 1):
        K.tuning_set("attn_hg", hg)
        for nm, b in (("real mask", bb), ("one row broadcast", b0)):
            t = timeit(lambda: K.attention(q, k, v, H, bias=b, bias_packed=True), 10)
            row.append(f"hg={hg} {nm}: {t * 1e3:6.3f} ms")
    K.tuning_set("attn_hg", 0)
    for qb, w3 in ((0, 1), (1, 1), (1, 2), (0, 1), (1, 2)):          # the rule (two blocks per wave) vs one block at two / three waves per SIMD
        K.tuning_set("attn_qb", qb)
        K.tuning_set("attn_w3", w3)
        t = timeit(lambda: K.attention(q, k, v, H, bias=bb, bias_packed=True), 10)
        row.append(f"qb={qb} w3={w3}: {t * 1e3:6.3f} ms")
    K.tuning_set("attn_qb", 0)
    K.tuning_set("attn_w3", 1)
    print(f"{name}: " + " | ".join(row))
