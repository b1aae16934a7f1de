#!/usr/bin/env python3
"""profiles/rNN_hbm_traffic.json + .txt from the PMC passes of tools/gpu/profile.sh (gpurun_out/<tag>/pmc_raw.txt).

    python tools/traffic_json.py gpurun_out/r04/pmc_raw.txt profiles/r04_hbm_traffic

Per launch of the persistent tower kernel (conv_tower_kernel<4>: the input convolution, 40 tower convolutions and the 6 SE units
of a 20b x 256 forward over a batch of 256 boards): fabric reads = TCC_EA0_RDREQ x 128 B, fabric writes = TCC_EA0_WRREQ x 64 B,
both calibrated in the same passes on 1 GiB streams (tools/ubench/hbm_calib.hip: plain 16-byte stores and the board kernel's
64-byte-per-4-lanes pattern).  Round 4: the counters are collected on the shipped launch itself (rounds 2-3: on per-layer
launches, summed)."""
import json
import re
import subprocess
import sys


def main(src, dst):
    text = open(src).read()
    sets = re.split(r"^## set \d+ \[", text, flags=re.M)[1:]
    tower, calib = {}, {}
    for blk in sets:
        for ln in blk.split("\n"):
            m = re.match(r"^(\S+)\s+dispatches=\s*(\d+) mean=\s*([\d.]+)", ln)
            if m and m.group(1) not in tower:
                tower[m.group(1)] = (float(m.group(3)), int(m.group(2)))
            m = re.match(r"^\s+(calib_\w+) \(1 GiB\)\s+(\S+)\s+dispatches=\s*\d+ mean=\s*([\d.]+)", ln)
            if m:
                calib[(m.group(1), m.group(2))] = float(m.group(3))
    GiB = float(1 << 30)
    rd_b = GiB / calib[("calib_read", "TCC_EA0_RDREQ_sum")]
    wr_b = GiB / calib[("calib_write64", "TCC_EA0_WRREQ_sum")]
    out_b = 256 * 361 * 256 * 2
    w_b = 256 * 256 * 9 * 2
    # 20b x 256, SE every third block: 34 plain tower layers (20 without residual, 14 with) + the input convolution, 6 SE layers (with residual)
    algo_plain = 34 * (2 * out_b + w_b) + 14 * out_b + (256 * 361 * 64 * 2 + out_b + 64 * 256 * 9 * 2)
    algo_se = 6 * (3 * out_b + w_b + (768 * 64 + 64 * 512) * 2)
    rd, wr = tower["TCC_EA0_RDREQ_sum"][0] * rd_b, tower["TCC_EA0_WRREQ_sum"][0] * wr_b
    res = {"tower_run": {"kernel": "conv_tower_kernel<4> (one launch: 35 plain layers incl. the input convolution + 6 with the SE unit)",
                         "read_bytes": round(rd), "write_bytes": round(wr), "algorithmic_bytes": round(algo_plain + algo_se),
                         "stored_bytes": 41 * out_b, "tcc_hit_rate": round(tower["TCC_HIT_sum"][0] / tower["TCC_REQ_sum"][0], 4),
                         "dispatches": tower["TCC_EA0_RDREQ_sum"][1]},
           "bytes_per_request": {"read": round(rd_b, 2), "write": round(wr_b, 2)},
           "source": ("rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum, one set per pass, on the persistent tower launch itself, "
                      "bytes per request calibrated on 1 GiB streams in the same passes")}
    try:
        res["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:  # noqa: BLE001
        res["commit"] = "?"
    json.dump(res, open(dst + ".json", "w"), indent=1)
    with open(dst + ".txt", "w") as f:
        f.write("# HBM-side (fabric) traffic of the persistent tower launch, MI355X, batch 256 x 19x19, 20b x 256 fp16 -- raw PMC passes below\n")
        f.write("# summary: " + json.dumps(res) + "\n")
        f.write(text)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
