"""Eager two-stream forward at cfg2 size: is it deterministic, and how far from the one-stream forward?  (bench.py's parity_check
found the eager two-stream step not bit-identical to the one-stream step at full size, while the captured graph is.)"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import configs, synthetic as S, tuning  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
dt = torch.bfloat16
if "--tuned" in sys.argv:
    print("tuned:", tuning.enable())
mv = configs.build_mv_model(1, device=dev, dtype=dt, xformers=True)
inp = S.mv_inputs(frames=16, pano_hw=(64, 128), pers_hw=(32, 32), seed=1, sam_frames=16, dtype=dt, device=dev)
cams = S.icosahedron_cameras(90, 256, device=dev)
rel = lambda a, b: float(((a.double() - b.double()).norm() / b.double().norm()).item())
outs = []
for name, dual, warp in (("one", False, False), ("one", False, False), ("two", True, False), ("two", True, False), ("two+warp", True, True), ("two+warp", True, True), ("one", False, False)):
    mv.dual_stream, mv.warp_streams = dual, warp
    random.seed(9)
    torch.manual_seed(3)
    pp, pn = mv(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True, **inp)
    torch.cuda.synchronize()
    outs.append((name, pp.clone(), pn.clone()))
ref = outs[0]
for name, pp, pn in outs[1:]:
    print(f"{name:9s} vs first one-stream: pers rel {rel(pp, ref[1]):.3e} equal {torch.equal(pp, ref[1])} | pano rel {rel(pn, ref[2]):.3e} equal {torch.equal(pn, ref[2])}")
