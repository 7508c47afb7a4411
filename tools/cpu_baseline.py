#!/usr/bin/env python
"""The CPU baseline of BASELINE.md section 4, measured once per round on a box of the GPU pool (no GPU work: the oracle on the host
cores): one full-width oracle step of BASELINE cfg1 with the probe's thread count and with every logical core, then ONE step of
the benchmarked cfg2 itself with the better of the two (~7 min).  Writes JSON lines.

    python tools/cpu_baseline.py [--skip-cfg2] > gpurun_out/cpu_baseline.jsonl
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--skip-cfg2", action="store_true")
a = ap.parse_args()
torch.set_grad_enabled(False)
args = argparse.Namespace(width_div=1, workload="cfg2")
mv = bench.configs.build_mv_model(1, device="cpu", dtype=torch.float32, xformers=True)
res = {}
for threads in ("probe", "all"):
    r = bench.cpu_baseline_step(mv, args, workload="cfg1", threads=threads)
    res[threads] = r
    print(json.dumps({"what": f"cfg1 step, threads={threads}", **r}), flush=True)
best = min(res, key=lambda k: res[k]["measured_cfg1_s_per_step"])
if not a.skip_cfg2:
    r = bench.cpu_baseline_step(mv, args, workload="cfg2", threads=res[best]["cores"])
    r["threads_chosen_by"] = f"the faster of probe ({res['probe']['cores']} threads, {res['probe']['measured_cfg1_s_per_step']:.1f} s per cfg1 step) and all " \
                             f"({res['all']['cores']} threads, {res['all']['measured_cfg1_s_per_step']:.1f} s)"
    print(json.dumps({"what": "cfg2 step measured directly", **r}), flush=True)
