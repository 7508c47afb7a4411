#!/usr/bin/env python
"""Group a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup): per-shape time of one step.

    rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python bench.py --steps S --warmup W --no-graph ...
    python tools/trace_by_shape.py DIR/.../t_kernel_trace.csv --steps N > profiles/rNN_step_by_shape.txt

Only the LAST `N x (dispatches per step)` dispatches are kept when --tail-from names the kernel that ends a step
(default: the DDIM update, two per step), so that model construction and warm-up stay out of the table.
"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:_ZN5im360\d+|im360::)(\w+)", name)
    if name.startswith("_ZN5im360"):
        m = re.match(r"_ZN5im360(?:12_GLOBAL__N_1)?\d+([a-z_0-9]+?)I", name)
        tail = name[m.end() - 1:][:70] if m else ""
        return (m.group(1) if m else name[:40]) + " " + tail
    if name.startswith("im360::"):
        return name[7:110]
    if "Cijk" in name:
        mt = re.search(r"MT\d+x\d+x\d+", name)
        return "hipBLASLt " + (mt.group(0) if mt else "") + (" custom" if name.startswith("Custom") else "")
    m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)<[^>]*?(\w+Functor\w*|\w+_kernel\w*)?", name)
    return "torch " + re.sub(r"\s+", " ", name)[:100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=1, help="steps the tail covers (times are divided by it)")
    ap.add_argument("--end-kernel", default="cfg_ddim", help="kernel whose launches end a step (2 per step)")
    ap.add_argument("--per-step", type=int, default=2)
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--skip-steps", type=int, default=0,
                    help="trailing steps to leave out (bench.py ends with its parity check: replayed graph, eager one-stream step, eager "
                         "TWO-stream step -- `--skip-steps 1` tabulates the eager one-stream step, where every kernel has the chip alone)")
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if args.end_kernel in r["Kernel_Name"]]
    need = args.steps * args.per_step
    skip = args.skip_steps * args.per_step
    assert len(ends) > need + skip, (len(ends), need, skip)
    last = len(ends) - 1 - skip
    lo, hi = ends[last - need] + 1, ends[last] + 1
    rows = rows[lo:hi]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        grid = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        wg = int(r["Workgroup_Size_X"])
        key = (short(r["Kernel_Name"]), tuple(g // w for g, w in zip(grid, (wg, int(r["Workgroup_Size_Y"]), int(r["Workgroup_Size_Z"])))), wg)
        a = agg[key]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(a[1] for a in agg.values())
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    print(f"# {len(rows)} dispatches over {args.steps} step(s): summed kernel time {tot / 1e6 / args.steps:.2f} ms per step, "
          f"first start to last end {span / 1e6 / args.steps:.2f} ms per step")
    print(f"# {'ms/step':>8} {'launches':>8} {'avg us':>8}  kernel, workgroups, threads")
    byk = collections.defaultdict(int)
    for (k, g, w), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f"  {t / 1e6 / args.steps:8.3f} {n / args.steps:8.1f} {t / n / 1e3:8.1f}  {k}  grid={g} x {w}")
    for (k, g, w), (n, t) in agg.items():
        byk[k.split(" ")[0] + (" " + k.split(" ")[1] if k.startswith(("hipBLASLt", "torch")) else "")] += t
    print("# by kernel family")
    for k, t in sorted(byk.items(), key=lambda kv: -kv[1])[:40]:
        print(f"  {t / 1e6 / args.steps:8.3f}  {k[:110]}")


if __name__ == "__main__":
    main()
