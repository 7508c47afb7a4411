#!/usr/bin/env python
"""What clock / power does the MI355X sustain under each kernel class?  Runs one kernel back-to-back for ~2.5 s while a
background thread samples `rocm-smi` (sclk, socket power), and prints the medians next to the kernel's rate.

    python tools/clock_probe.py            # writes gpurun_out/clock_probe.txt
"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402

DT, DEV = torch.bfloat16, "cuda"


def rn(*s, scale=1.0):
    return (torch.randn(*s, device=DEV, dtype=torch.float32) * scale).to(DT)


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        except Exception as e:       # noqa: BLE001
            out.append(("err", str(e)))
            return
        m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
        pw = re.search(r"Power \(W\): ([\d.]+)", txt)
        out.append((int(m.group(1)) if m else None, float(pw.group(1)) if pw else None, txt if not m else ""))
        time.sleep(0.05)


def probe(name, fn, flops, secs=2.5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    t_cold = a.elapsed_time(b) / 10
    n = max(10, int(secs * 1e3 / t_cold))
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples))
    th.start()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    t = a.elapsed_time(b) / n
    clk = sorted(s[0] for s in samples if s[0])
    pw = sorted(s[1] for s in samples if len(s) > 1 and s[1])
    med = lambda v: v[len(v) // 2] if v else None       # noqa: E731
    line = (f"{name:30s} first-10 {t_cold:7.3f} ms | sustained {t:7.3f} ms {flops / t / 1e9:7.0f} TF/s | sclk median {med(clk)} MHz "
            f"(min {clk[0] if clk else None}, max {clk[-1] if clk else None}, {len(clk)} samples) | power median {med(pw)} W")
    if not clk and samples:
        line += " | raw: " + repr(samples[0])[:400]
    print(line, flush=True)
    return line


def main():
    torch.set_grad_enabled(False)
    lines = []
    x, w, b = rn(655360, 320), rn(2560, 320, scale=320 ** -0.5), rn(2560)
    wp, bp = K.pack_geglu(w, b)
    lines.append(probe("geglu L0 pers (fused)", lambda: K.linear_geglu(x, wp, bp, 1280), 2.0 * 655360 * 320 * 2560))
    x3, w3 = rn(640, 32, 32, 320), K.pack_conv_weight(rn(320, 320, 3, 3, scale=2880 ** -0.5))
    lines.append(probe("conv 3x3 pers L0 320>320", lambda: K.conv2d(x3, w3, 320), 2.0 * 655360 * 2880 * 320))
    x4, w4 = rn(640, 16, 16, 640), K.pack_conv_weight(rn(640, 640, 3, 3, scale=5760 ** -0.5))
    lines.append(probe("conv 3x3 pers L1 640>640", lambda: K.conv2d(x4, w4, 640), 2.0 * 163840 * 5760 * 640))
    a, bm = rn(8192, 8192), rn(8192, 8192)
    lines.append(probe("hipBLASLt 8192^3 bf16", lambda: torch.matmul(a, bm), 2.0 * 8192 ** 3))
    q = rn(32, 8192, 3 * 320)
    lines.append(probe("attention pano L0 self", lambda: K.attention(q[..., :320], q[..., 320:640], q[..., 640:], 5, 64 ** -0.5),
                       4.0 * 32 * 5 * 8192 * 8192 * 64))
    xs = rn(655360, 320)
    g, be = rn(320), rn(320)
    lines.append(probe("layernorm L0 pers (HBM)", lambda: K.layer_norm(xs, g, be), 0.0))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/clock_probe.txt", "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
