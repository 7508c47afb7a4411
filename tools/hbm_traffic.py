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
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from imagine360_amd import kernels as K
    torch.set_grad_enabled(False)
    dt, dev = torch.bfloat16, "cuda"
    rn = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32).to(dt)
    K.tuning_set("conv_halo", 1 if halo else 0)
    man = []

    def conv(name, N, H, W, Ci, Co, **kw):
        x = rn(N, H, W, Ci)
        w = K.pack_conv_weight(rn(Co, Ci, 3, 3) * (9 * Ci) ** -0.5)
        b = rn(Co)
        wo = kw.get("wout", W)
        for _ in range(REPS):
            K.conv2d(x, w, Co, bias=b, **kw)
        man.append(dict(cls="conv", name=name, algorithmic_bytes=2.0 * (N * H * W * Ci + 9 * Ci * Co + N * H * wo * Co),
                        input_bytes=2.0 * N * H * W * Ci, flops=2.0 * N * H * wo * Ci * Co * 9))

    def lin(name, M, Kd, N, res):
        x, w = rn(M, 1, 1, Kd), K.pack_conv_weight(rn(N, Kd, 1, 1) * Kd ** -0.5)
        b, r = rn(N), (rn(M, 1, 1, N) if res else None)
        for _ in range(REPS):
            K.conv2d(x, w, N, bias=b, res=r)
        man.append(dict(cls="gemm", name=name, algorithmic_bytes=2.0 * (M * Kd + Kd * N + M * N * (2 if res else 1)),
                        input_bytes=2.0 * M * Kd, flops=2.0 * M * Kd * N))

    def geglu(name, M, C):
        x, w, b = rn(M, C), rn(8 * C, C) * C ** -0.5, rn(8 * C)
        wp, bp = K.pack_geglu(w, b)
        for _ in range(REPS):
            K.linear_geglu(x, wp, bp, 4 * C)
        man.append(dict(cls="gemm", name=name, algorithmic_bytes=2.0 * (M * C + 8 * C * C + M * 4 * C),
                        input_bytes=2.0 * M * C, flops=2.0 * M * C * 8 * C))

    conv("pers L0 320->320", 640, 32, 32, 320, 320)
    conv("pano L0 conv2 320->320 (W+4 -> W)", 32, 64, 132, 320, 320, x_off=2, wout=128)
    conv("pers L1 640->640", 640, 16, 16, 640, 640)
    conv("pers L2 1280->1280", 640, 8, 8, 1280, 1280)
    conv("pers up L0 960->320", 640, 32, 32, 960, 320)
    conv("pers up L1 1920->640", 640, 16, 16, 1920, 640)
    lin("pers L0 out-proj 320->320 + res", 655360, 320, 320, True)
    lin("pers L0 qkv 320->960", 655360, 320, 960, False)
    lin("pers L0 FF-out 1280->320 + res", 655360, 1280, 320, True)
    lin("pers L1 qkv 640->1920", 163840, 640, 1920, False)
    lin("pers L1 FF-out 2560->640 + res", 163840, 2560, 640, True)
    geglu("pers L0 GEGLU 320->2x1280", 655360, 320)
    geglu("pers L1 GEGLU 640->2x2560", 163840, 640)
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
            "fetch_kb_raw": fk, "write_kb": wk, "traffic_bytes": (2.0 * fk + wk) * 1024.0,
            "algorithmic_bytes": m["algorithmic_bytes"], "fetch_over_input": 2.0 * fk * 1024.0 / m["input_bytes"],
            "traffic_over_algorithmic": (2.0 * fk + wk) * 1024.0 / m["algorithmic_bytes"]}
    json.dump(out, open(os.path.join(d, "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        manifest_and_run(sys.argv[2], halo=len(sys.argv) > 3 and sys.argv[3] == "halo")
    else:
        parse(sys.argv[2])
