#!/usr/bin/env python
"""One line per kernel: the order of its matrix, memory and wait instructions, run-length coded -- enough to see whether a K loop
is clean (no scratch reload, no vmcnt wait of the compiler's between an LDS-DMA request and the wait the kernel places itself).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S imagine360_amd/csrc/conv3x3.hip -o /tmp/dev.s
    python tools/kernel_event_map.py /tmp/dev.s [substring of the mangled kernel name ...]

M v_mfma · G global_load_lds (LDS-DMA) · r global_load_dwordx4 · W global_store · d ds_read · D ds_write · S / L scratch store / load
(spill / reload: VMEM operations) · | s_barrier · vN s_waitcnt vmcnt(N).  Code order, not execution order: both arms of a branch
appear one after the other."""
import itertools
import re
import sys


def event_map(body):
    ev = []
    for line in body:
        t = line.strip()
        if t.startswith("v_mfma"):
            ev.append("M")
        elif t.startswith("scratch_store"):
            ev.append("S")
        elif t.startswith("scratch_load"):
            ev.append("L")
        elif t.startswith("s_barrier"):
            ev.append("|")
        elif t.startswith("global_store"):
            ev.append("W")
        elif t.startswith("global_load_lds"):
            ev.append("G")
        elif t.startswith("global_load_dwordx4"):
            ev.append("r")
        elif t.startswith("ds_write"):
            ev.append("D")
        elif t.startswith("ds_read"):
            ev.append("d")
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            ev.append("v" + re.search(r"vmcnt\((\d+)\)", t).group(1) + " ")
    out = []
    for c, g in itertools.groupby(ev):
        n = len(list(g))
        out.append(f"{c}{n}" if n > 1 else c)
    return " ".join(out)


def main():
    lines = open(sys.argv[1]).read().split("\n")
    wanted = sys.argv[2:]
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for i0 in starts:
        name = lines[i0].split(":")[0]
        if wanted and not any(w in name for w in wanted):
            continue
        i1 = next((i for i in range(i0 + 1, len(lines)) if lines[i].startswith(".Lfunc_end")), len(lines))
        print(name)
        print("   " + event_map(lines[i0:i1]))


if __name__ == "__main__":
    main()
