"""Cycle stamps of one workgroup of the persistent GEMM kernel (experiment build only):
    git apply tools/patches/ring_cycle_stamps.patch && make -C imagine360_amd/csrc && python tools/stamps_probe.py ; git checkout imagine360_amd/csrc/conv3x3.hip
Stamps at: tile top, stage 0 landed, K loop done, next tile requested, epilogue code done -- waves 0 (leading group) and 4 (trailing)."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K
from tools.bench_kernels import rn
L = K.lib()
def report(name, L):
    buf = (ctypes.c_ulonglong * 64)()
    L.im360_debug_stamps(buf)
    for g, base in (("leading", 0), ("trailing", 32)):
        v = list(buf)[base:base + 30]
        rows = []
        for t in range(1, 5):
            s = v[t * 5:t * 5 + 5]
            rows.append(f"tile {t}: top->stage0 landed {s[1]-s[0]:6d} | K loop {s[2]-s[1]:6d} | barrier+next-tile requests {s[3]-s[2]:5d} | epilogue code {s[4]-s[3]:6d} | whole tile {v[(t + 1) * 5] - s[0]:6d}")
        print(name, g, "(shader-clock cycles, s_memtime)\n   " + "\n   ".join(rows), flush=True)


cases = []
for name, M, Kd, N, use_res in [("pers L0 proj+res", 655360, 320, 320, True), ("pers L0 qkv plain", 655360, 320, 960, False), ("pers L2 ff-in", 40960, 1280, 10240, False)]:
    x, w, b, r = rn(M, Kd), rn(N, Kd) * Kd ** -0.5, rn(N), rn(M, N) if use_res else None
    wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
    K.tuning_set("conv_dbg", 64)
    for _ in range(3): K.linear(x, wp, N, bias=b, res=r)
    torch.cuda.synchronize()
    K.tuning_set("conv_dbg", 0)
    report(name, L)
    del x, w, b, r, wp
# LayerNorm folded into the QKV projection (EPI 3) and the fused GEGLU with the fold (EPI 4), level 0
M, C = 655360, 320
x = rn(M, C)
gam, bet = rn(C) * 0.1 + 1, rn(C) * 0.1
xs = x.float().reshape(M, C // K.ROW_SLICE, K.ROW_SLICE)
st = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], dim=-1).contiguous()
wq, bq = rn(3 * C, C) * C ** -0.5, rn(3 * C)
wg, c1, c2 = K.fold_layer_norm(wq, bq, gam, bet)
wgp = K.pack_conv_weight(wg.reshape(3 * C, C, 1, 1).contiguous())
K.tuning_set("conv_dbg", 64)
for _ in range(3): K.linear_ln(x, wgp, c1, c2, st, 1e-5, 3 * C)
torch.cuda.synchronize()
K.tuning_set("conv_dbg", 0)
report("pers L0 qkv LN-folded", L)
gw, gb = rn(8 * C, C) * C ** -0.5, rn(8 * C)
gwf, gc1, gc2 = K.fold_layer_norm(gw, gb, gam, bet)
gwfp, gc1p = K.pack_geglu(gwf, gc1)
gc2p = K.interleave_geglu(gwf, gc2)[1].contiguous()
K.tuning_set("conv_dbg", 64)
for _ in range(3): K.linear_geglu_ln(x, gwfp, gc1p.contiguous(), gc2p, st, 1e-5, 4 * C)
torch.cuda.synchronize()
K.tuning_set("conv_dbg", 0)
report("pers L0 GEGLU LN-folded", L)
