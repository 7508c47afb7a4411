#!/usr/bin/env python
"""HBM-side traffic of the conv / GEMM kernels at the cfg2 shapes (profiles/r02_hbm_traffic.json).

    tools/hbm_traffic.sh <out_dir_under_gpurun_out>        # drives this file under rocprofv3

`run` launches every shape of the manifest exactly REPS times (conv class: the step's convolutions; gemm class: the routed
token-major Linears and the fused GEGLU) and writes the manifest (shape order + algorithmic bytes); `parse` reads the
rocprofv3 counter CSVs of the two passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, rocprofv3
PMC slots) and divides.  gfx950: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 B -> x2
(MI355X_MICROARCH.md, HBM); Infinity-Cache hits are counted, not excluded.
"""
import csv
import glob
import json
import os
import sys

REPS = 3
KERNELS = ("conv_igemm_kernel", "conv_ring_kernel", "conv_halo_kernel")


def manifest_and_run(path, halo):
    """Every shape is launched through the wrappers the model uses (imagine360_amd.kernels), so each manifest entry carries
    the SHAPE KEY the launch-mix statistics of bench.py use (kernels.SHAPES) next to its algorithmic bytes."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from imagine360_amd import kernels as K
    torch.set_grad_enabled(False)
    dt, dev = torch.bfloat16, "cuda"
    rn = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32).to(dt)
    K.tuning_set("conv_halo", 1 if halo else 0)
    man = []

    def entry(cls, name, fn, input_bytes):
        K.SHAPES = {}
        for _ in range(REPS):
            fn()
        (kind, key), (n, by) = next(iter(K.SHAPES.items()))
        assert len(K.SHAPES) == 1 and kind == cls and n == REPS, K.SHAPES
        K.SHAPES = None
        man.append(dict(cls=cls, name=name, key=key, algorithmic_bytes=by / n, input_bytes=input_bytes))

    def conv(name, N, H, W, Ci, Co, **kw):
        x = rn(N, H, W, Ci)
        w = K.pack_conv_weight(rn(Co, Ci, 3, 3) * (9 * Ci) ** -0.5)
        b = rn(Co)
        entry("conv", name, lambda: K.conv2d(x, w, Co, bias=b, **kw), 2.0 * N * H * W * Ci)

    def lin(name, M, Kd, N, res, stats=False):
        x, w = rn(M, Kd), K.pack_conv_weight(rn(N, Kd, 1, 1) * Kd ** -0.5)
        b, r = rn(N), (rn(M, N) if res else None)
        if stats:
            entry("gemm", name, lambda: K.linear(x, w, N, bias=b, res=r, row_stats=True), 2.0 * M * Kd)
        else:
            x4, r4 = x.reshape(M, 1, 1, Kd), (r.reshape(M, 1, 1, N) if res else None)
            entry("gemm", name, lambda: K.conv2d(x4, w, N, bias=b, res=r4), 2.0 * M * Kd)

    def lin_ln(name, M, Kd, N):
        x, w = rn(M, Kd), rn(N, Kd) * Kd ** -0.5
        wg, c1, c2 = K.fold_layer_norm(w, None, rn(Kd), rn(Kd))
        wp = K.pack_conv_weight(wg.reshape(N, Kd, 1, 1).contiguous())
        st = torch.rand(M, Kd // 160, 2, device=dev) + 1.0
        entry("gemm", name, lambda: K.linear_ln(x, wp, c1, c2, st, 1e-5, N), 2.0 * M * Kd)

    def geglu(name, M, C, ln=False):
        x, w, b = rn(M, C), rn(8 * C, C) * C ** -0.5, rn(8 * C)
        if ln:
            wg, c1, c2 = K.fold_layer_norm(w, b, rn(C), rn(C))
            wp, c1p = K.pack_geglu(wg, c1)
            c2p = K.interleave_geglu(wg, c2)[1].contiguous()
            st = torch.rand(M, C // 160, 2, device=dev) + 1.0
            entry("gemm", name, lambda: K.linear_geglu_ln(x, wp, c1p.contiguous(), c2p, st, 1e-5, 4 * C), 2.0 * M * C)
        else:
            wp, bp = K.pack_geglu(w, b)
            entry("gemm", name, lambda: K.linear_geglu(x, wp, bp, 4 * C), 2.0 * M * C)

    conv("pers L0 320->320", 640, 32, 32, 320, 320)
    conv("pano L0 conv2 320->320 (W+4 -> W)", 32, 64, 132, 320, 320, x_off=2, wout=128)
    conv("pers L1 640->640", 640, 16, 16, 640, 640)
    conv("pers L2 1280->1280", 640, 8, 8, 1280, 1280)
    conv("pers up L0 960->320", 640, 32, 32, 960, 320)
    conv("pers up L1 1920->640", 640, 16, 16, 1920, 640)
    lin("pers L0 proj-in 320->320", 655360, 320, 320, False)
    lin("pers L0 proj-out 320->320 + res", 655360, 320, 320, True)
    lin("pers L0 out-proj 320->320 + res + row stats", 655360, 320, 320, True, stats=True)
    lin("pers L0 proj-in 320->320 + row stats", 655360, 320, 320, False, stats=True)
    lin("pers L0 qkv 320->960", 655360, 320, 960, False)
    lin_ln("pers L0 LN-folded qkv 320->960", 655360, 320, 960)
    lin_ln("pers L0 LN-folded to_q 320->320", 655360, 320, 320)
    lin("pers L0 FF-out 1280->320 + res", 655360, 1280, 320, True)
    lin("pers L0 FF-out 1280->320 + res + row stats", 655360, 1280, 320, True, stats=True)
    lin("pers L1 qkv 640->1920", 163840, 640, 1920, False)
    lin_ln("pers L1 LN-folded qkv 640->1920", 163840, 640, 1920)
    lin_ln("pers L1 LN-folded to_q 640->640", 163840, 640, 640)
    lin("pers L1 out-proj 640->640 + res + row stats", 163840, 640, 640, True, stats=True)
    lin("pers L1 proj-out 640->640 + res", 163840, 640, 640, True)
    lin("pers L1 FF-out 2560->640 + res", 163840, 2560, 640, True)
    geglu("pers L0 GEGLU 320->2x1280", 655360, 320)
    geglu("pers L0 LN-folded GEGLU 320->2x1280", 655360, 320, ln=True)
    geglu("pers L1 GEGLU 640->2x2560", 163840, 640)
    geglu("pers L1 LN-folded GEGLU 640->2x2560", 163840, 640, ln=True)
    torch.cuda.synchronize()
    json.dump(man, open(path, "w"), indent=1)


def per_dispatch(d, counter):
    rows = {}
    for f in glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in KERNELS):
                rows[int(r["Dispatch_Id"])] = rows.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


def parse(d):
    man = json.load(open(os.path.join(d, "manifest.json")))
    fetch, write = per_dispatch(d, "FETCH_SIZE"), per_dispatch(d, "WRITE_SIZE")
    # pack_conv_weight etc. are other kernels; the conv kernels are launched REPS times per manifest entry, in order
    assert len(fetch) == len(write) == REPS * len(man), (len(fetch), len(write), len(man))
    out = {"_note": "bytes per launch; traffic_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (rocprofv3 --pmc, separate passes; "
                    "gfx950 fetch correction of MI355X_MICROARCH.md; Infinity-Cache hits are counted)"}
    for i, m in enumerate(man):
        fk = sum(fetch[i * REPS:(i + 1) * REPS]) / REPS
        wk = sum(write[i * REPS:(i + 1) * REPS]) / REPS
        out.setdefault(m["cls"], {})[m["name"]] = {
            "key": m.get("key"), "fetch_kb_raw": fk, "write_kb": wk, "traffic_bytes": (2.0 * fk + wk) * 1024.0,
            "algorithmic_bytes": m["algorithmic_bytes"], "fetch_over_input": 2.0 * fk * 1024.0 / m["input_bytes"],
            "traffic_over_algorithmic": (2.0 * fk + wk) * 1024.0 / m["algorithmic_bytes"]}
    json.dump(out, open(os.path.join(d, "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        manifest_and_run(sys.argv[2], halo=len(sys.argv) > 3 and sys.argv[3] == "halo")
    else:
        parse(sys.argv[2])
