import os, sys, ctypes, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/imagine360_amd") else os.getcwd())
from imagine360_amd import kernels as K
from tools.bench_kernels import rn
L = K.lib()
for name, M, Kd, N, use_res in [("pers L0 proj+res", 655360, 320, 320, True), ("pers L0 qkv", 655360, 320, 960, False), ("pers L1 ff-out+res", 163840, 2560, 640, True), ("pers L2 ff-in", 40960, 1280, 10240, False)]:
    x, w, b, r = rn(M, Kd), rn(N, Kd) * Kd ** -0.5, rn(N), rn(M, N) if use_res else None
    wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
    K.tuning_set("conv_dbg", 64)
    for _ in range(3): K.linear(x, wp, N, bias=b, res=r)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    L.im360_debug_stamps(buf)
    K.tuning_set("conv_dbg", 0)
    for g, base in (("leading", 0), ("trailing", 32)):
        v = list(buf)[base:base + 30]
        t0 = v[0]
        rows = []
        for t in range(5):
            s = v[t * 5:t * 5 + 5]
            nxt = v[(t + 1) * 5] if t < 5 else 0
            rows.append(f"tile {t}: top->stage0 landed {s[1]-s[0]:6d} | K loop {s[2]-s[1]:6d} | barrier+next-tile requests {s[3]-s[2]:5d} | epilogue code {s[4]-s[3]:6d}")
        print(name, g, "(cycles of the 100 MHz * ? counter)\n   " + "\n   ".join(rows))
