#!/usr/bin/env python
"""Where does the product's error come from?  The SAME host code (imagine360_amd, width / 5 dual model) runs twice on the
same 16-bit weights and inputs: on the GPU through the HIP kernels, and on the CPU through tests/_emu_kernels.py (exact
fp32 arithmetic inside every kernel, outputs rounded to the 16-bit dtype at the same points).  The per-stage difference of
the two (after every ResnetBlock / spatial transformer / motion module of the encoder, every WarpAttn, the outputs) is
what the kernels add beyond storage rounding -- and names the stage where they add it.

    python tools/diag_stage_errors.py [bf16|fp16]
"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import _emu_kernels as E  # noqa: E402
from imagine360_amd import configs, synthetic as S  # noqa: E402

torch.set_grad_enabled(False)
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.bfloat16
rel = lambda a, b: ((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm().clamp_min(1e-30)).item()
inp = S.mv_inputs(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), seed=0, sam_frames=16)
cams = S.icosahedron_cameras(90, 128)


def run(device):
    mv = configs.build_mv_model(5, device=device, dtype=dt, xformers=True)
    mv.noise_on_host = True
    dinp = S.cast_mv_inputs(inp, device, dt)
    torch.manual_seed(7)
    random.seed(7)
    mv.taps, mv.debug_taps = {}, {}
    pers, pano = mv(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True, **dinp)
    stages = dict(mv.debug_taps)
    stages.update({k: v for k, v in mv.taps.items()})
    stages["out"] = (pers, pano)
    return {k: (a.float().cpu(), b.float().cpu()) for k, (a, b) in stages.items()}


gpu = run("cuda")
with E.patched_kernels():
    cpu = run("cpu")
print(f"stage-by-stage: HIP kernels vs exact kernels with the same 16-bit rounding points ({dt})")
for k in cpu:
    print(f"{k:10s} pers {rel(gpu[k][0], cpu[k][0]):.2e}   pano {rel(gpu[k][1], cpu[k][1]):.2e}")
