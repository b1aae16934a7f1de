#!/usr/bin/env python3
"""Copy what tools/gpu/profile.sh left under gpurun_out/<tag>/ into profiles/ under the round's names:

    python tools/evidence_to_profiles.py gpurun_out/r03 r03

kernel-trace stats, GPU suite log, configs[4], fp32, the default bench line, the PMC passes (with the medians over the dispatches
in the header: the means in the body include each kernel's first, cold launch) and the HBM traffic JSON (tools/traffic_json.py)."""
import collections
import csv
import glob
import os
import shutil
import statistics
import subprocess
import sys


def medians(path, kern):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"]:
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {c: statistics.median(v.values()) for c, v in per.items()}


def main(src, tag):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    for a, b in (("rocprofv3_kernel_stats.txt", f"{tag}_rocprofv3_kernel_stats.txt"), ("gpu_tests.log", f"{tag}_gpu_tests.log"),
                 ("config5.json", f"{tag}_config5.json"), ("bench_fp32.json", f"{tag}_bench_fp32.json")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(prof, b))
    bd = os.path.join(src, "bench_default.json")
    if os.path.exists(bd):
        open(os.path.join(prof, f"{tag}_bench_default.json"), "w").write(open(bd).read().strip().split("\n")[-1] + "\n")
    head = [f"# rocprofv3 --pmc <one set per pass> --kernel-trace -- python bench.py --steps 2 --warmup 1 ... (tools/gpu/profile.sh): counters of the",
            "# persistent tower launch ITSELF (conv_tower_kernel<4>: 41 convolutions + 6 SE units of one forward); `per-layer` rows: the compiled"
            " per-layer kernels of the same forward under SAYURI_TOWER=0; calib_* = 1 GiB streams under the same set",
            "# counters are sums over the 8 XCDs: GRBM_GUI_ACTIVE / 8 = cycles per launch.  MEDIANS over the dispatches (the means below include the"
            " first, cold launch of each kernel):"]
    p1 = glob.glob(os.path.join(src, "prof", "p1", "*counter_collection.csv"))
    p2 = glob.glob(os.path.join(src, "prof", "p2", "*counter_collection.csv"))
    for kern in ("conv_tower_kernel<4",):
        line = f"#   {kern}>"
        if p1:
            m = medians(p1[0], kern)
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            line += (f"  {cyc / 1e3:.1f} k cycles per launch, SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs) = {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.1f} %,"
                     f" SQ_WAIT_ANY / SQ_WAVE_CYCLES {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.0f} %")
        if p2:
            m = medians(p2[0], kern)
            line += (f", LDS bank-conflict cycles {100 * m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.0f} % of LDS-active,"
                     f" LDS active {100 * m['SQ_LDS_IDX_ACTIVE'] / (m['GRBM_GUI_ACTIVE'] / 8 * 256):.0f} % of the cycles")
        head.append(line)
    raw = os.path.join(src, "pmc_raw.txt")
    if os.path.exists(raw):
        open(os.path.join(prof, f"{tag}_rocprofv3_pmc_conv_tower.txt"), "w").write("\n".join(head) + "\n" + open(raw).read())
        subprocess.check_call([sys.executable, os.path.join(root, "tools", "traffic_json.py"), raw, os.path.join(prof, f"{tag}_hbm_traffic")])
    print("\n".join(head[3:]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
