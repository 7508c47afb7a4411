"""Where does the eagerly issued TWO-stream forward stop being bit-identical to the one-stream forward?  (round 4
saw the eager two-stream step at cfg2 size differ from the one-stream step in one bench run of two, switched eager issue to one
stream and left the cause open.)

Every tensor that leaves a WarpAttn call / a down-block layer is reduced ON THE FLY to a 64-bit checksum of its bits (no
reference to the tensor is kept: holding the intermediates alive would hide an allocator-lifetime hazard), for

    one      one stream (the reference checksums)
    two      panorama segments + the panorama direction of every WarpAttn on the side stream (MultiViewBaseModel.dual_stream_eager)
    two+rec  the same with every tensor that crosses streams handed to Tensor.record_stream (mv.record_streams = True)
    two+sync the same with a device synchronise at every join (mv.sync_joins = True): no concurrency left, same allocation order

several times each, alternating; prints, per run, the first tap whose checksum differs from the one-stream run.

    python tools/dual_stream_race.py [--tuned] [--runs 6] [--workload cfg2|cfg1] [--width-div 1]
"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import configs, synthetic as S, tuning  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tuned", action="store_true")
ap.add_argument("--runs", type=int, default=6)
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--width-div", type=int, default=1)
ap.add_argument("--fine", action="store_true", help="also checksum every down-block layer output (keeps a segment's intermediates alive until its join: may hide a lifetime hazard)")
args = ap.parse_args()

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
dt = torch.bfloat16
if args.tuned:
    print("tuned gemm table:", tuning.enable(), tuning.STATUS)
W = {"cfg2": dict(frames=16, pano_hw=(64, 128), pers_hw=(32, 32), px=256), "cfg1": dict(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), px=128)}[args.workload]
mv = configs.build_mv_model(args.width_div, device=dev, dtype=dt, xformers=True)
inp = S.mv_inputs(frames=W["frames"], pano_hw=W["pano_hw"], pers_hw=W["pers_hw"], seed=1, sam_frames=16, dtype=dt, device=dev)
cams = S.icosahedron_cameras(90, W["px"], device=dev)


def checksum(t):
    return t.contiguous().view(torch.int16).to(torch.int64).sum()


def run(mode):
    mv.dual_stream = mode != "one"
    mv.dual_stream_eager = mode != "one"
    mv.warp_streams = True
    mv.record_streams = mode == "two+rec"
    mv.sync_joins = mode == "two+sync"
    mv.tap_fn = checksum
    mv.taps, mv.debug_taps = {}, ({} if args.fine else None)
    random.seed(9)
    torch.manual_seed(3)
    pp, pn = mv(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True, **inp)
    torch.cuda.synchronize()
    sums = {}
    for k, v in list((mv.debug_taps or {}).items()) + list(mv.taps.items()):
        if isinstance(v, tuple) and all(torch.is_tensor(x) and x.numel() == 1 for x in v):
            sums[k] = tuple(int(x) for x in v)
    sums["out"] = (int(checksum(pp)), int(checksum(pn)))
    mv.taps = mv.debug_taps = mv.tap_fn = None
    return sums


ref = run("one")
again = run("one")
print(f"{len(ref)} taps; one-stream forward repeatable: {again == ref}")
order = list(ref)
bad = {}
for r in range(args.runs):
    for mode in ("two", "two+rec", "two+sync", "one"):
        s = run(mode)
        diff = [k for k in order if s.get(k) != ref[k]]
        bad.setdefault(mode, []).append(diff[0] if diff else None)
        if diff:
            which = ["pers", "pano"]
            first = diff[0]
            side = [which[i] for i in range(2) if s[first][i] != ref[first][i]]
            print(f"run {r} {mode:8s}: {len(diff)} of {len(order)} taps differ, first {first} ({'/'.join(side)})")
for mode, firsts in bad.items():
    n = sum(f is not None for f in firsts)
    print(f"{mode:8s}: {n} of {len(firsts)} runs differ from the one-stream forward; first differing taps: {sorted(set(f for f in firsts if f))}")
