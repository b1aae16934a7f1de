#!/usr/bin/env python3
"""What lies between two persistent tower launches?  Reads a `rocprofv3 --kernel-trace --memory-copy-trace` directory of
`bench.py` (microbench, then the pump with fp32 planes, then the pump with packed planes) and prints, per consecutive pair of
tower launches, the gap between them (end -> next start) and what ran inside it (kernels and copies, start relative to the
first tower's end, duration) -- the microbench's gap is the two small kernels' own time; anything the pump's gap has on top of
that is what two tickets in flight cost.

    python tools/pump_gaps.py gpurun_out/pump_gaps/trace [--detail 6]
"""
import csv
import glob
import os
import statistics
import sys


def load(dirname):
    ev = []
    for f in glob.glob(os.path.join(dirname, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sayuri::", "")[:48],
                       "q%s" % r.get("Queue_Id", "?")))
    for f in glob.glob(os.path.join(dirname, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?").replace("MEMORY_COPY_", ""), ""))
    ev.sort()
    return ev


def main():
    d = sys.argv[1]
    detail = int(sys.argv[sys.argv.index("--detail") + 1]) if "--detail" in sys.argv else 4
    ev = load(d)
    towers = [i for i, e in enumerate(ev) if e[2].startswith("conv_tower_kernel")]
    rows = []
    for a, b in zip(towers, towers[1:]):
        end_a, start_b = ev[a][1], ev[b][0]
        inside = [e for e in ev if e[1] > end_a and e[0] < start_b and not e[2].startswith("conv_tower_kernel")]
        kind = "packed" if any("pack_bits" in e[2] for e in inside) else "planes"
        ncopy = sum(1 for e in inside if e[2].startswith("copy"))
        rows.append((a, (start_b - end_a) / 1e3, kind, ncopy, inside, end_a, (ev[b][1] - ev[b][0]) / 1e3))
    # segments: the microbench has no copies between towers; the pump has
    seg = {"microbench (no copies between towers)": [r for r in rows if r[3] == 0 and r[2] == "planes"],
           "pump, fp32 planes": [r for r in rows if r[3] > 0 and r[2] == "planes"], "pump, packed planes": [r for r in rows if r[2] == "packed"]}
    for name, rs in seg.items():
        rs = [r for r in rs if r[1] < 2000]  # segment boundaries (host work between two phases of the bench) are not gaps of a pipeline
        if not rs:
            continue
        gaps = [r[1] for r in rs]
        durs = [r[6] for r in rs]
        print(f"## {name}: {len(rs)} pairs, gap end -> next start: median {statistics.median(gaps):.1f} us, min {min(gaps):.1f}, max {max(gaps):.1f};"
              f" tower duration median {statistics.median(durs):.1f} us; period median {statistics.median(g + t for g, t in zip(gaps, durs)):.1f} us")
        for r in rs[len(rs) // 2: len(rs) // 2 + detail]:
            print(f"   gap {r[1]:7.1f} us:")
            for e in r[4]:
                print(f"      +{(e[0] - r[5]) / 1e3:8.1f} us  {(e[1] - e[0]) / 1e3:7.1f} us  {e[2]} {e[3]}")


if __name__ == "__main__":
    main()
