#!/usr/bin/env python
"""The CPU baseline of BASELINE.md section 4, measured once per round on a box of the GPU pool (no GPU work: the oracle on the host
cores): one full-width oracle step of BASELINE cfg1 with the probe's thread count and with every logical core, then ONE step of
the benchmarked cfg2 itself with the better of the two (~7 min).  Writes JSON lines.

    python tools/cpu_baseline.py [--skip-cfg2] > gpurun_out/cpu_baseline.jsonl
    python tools/cpu_baseline.py --cfg2-threads 32        # only the direct cfg2 step, on a thread count already chosen

Round 5 measured (AMD EPYC 9575F, 256 logical cores, profiles/r05_cpu_baseline.jsonl): one cfg1 step takes 51.9 s on the probe's 32
threads and **816 s on all 256** -- the per-frame ops of this path are small, and oversubscribing them costs 16x: BASELINE.md section 4's
`torch.set_num_threads(os.cpu_count())` is the WORSE baseline on this host, which is why bench.py keeps the probe.
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
ap.add_argument("--cfg2-threads", type=int, default=0, help="run ONLY the direct cfg2 step, on this many threads")
a = ap.parse_args()
torch.set_grad_enabled(False)
args = argparse.Namespace(width_div=1, workload="cfg2")
mv = bench.configs.build_mv_model(1, device="cpu", dtype=torch.float32, xformers=True)
if a.cfg2_threads:
    r = bench.cpu_baseline_step(mv, args, workload="cfg2", threads=a.cfg2_threads)
    r["threads_chosen_by"] = f"given ({a.cfg2_threads}): the probe's choice on this host model, where every logical core is 16x slower (profiles/r05_cpu_baseline.jsonl)"
    print(json.dumps({"what": "cfg2 step measured directly", **r}), flush=True)
    sys.exit(0)
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
