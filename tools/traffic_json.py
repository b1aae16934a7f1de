#!/usr/bin/env python3
"""profiles/rNN_hbm_traffic.json + .txt from the PMC passes of tools/gpu/profile.sh (gpurun_out/<tag>/pmc_raw.txt).

    python tools/traffic_json.py gpurun_out/r03b/pmc_raw.txt profiles/r03_hbm_traffic

Per launch of the plain tower convolution (conv_board_kernel<4>, the mean includes the input convolution) and of the
SE-carrying one (conv_board_se_kernel<4>): fabric reads = TCC_EA0_RDREQ x 128 B, fabric writes = TCC_EA0_WRREQ x 64 B, both
calibrated in the same passes on 1 GiB streams (tools/ubench/hbm_calib.hip: plain 16-byte stores and the board kernel's
64-byte-per-4-lanes pattern).  One persistent tower launch (20b x 256: 35 plain + 6 SE layers) = the sum over its layers."""
import json
import re
import subprocess
import sys


def main(src, dst):
    text = open(src).read()
    sets = re.split(r"^## set \d+ \[", text, flags=re.M)[1:]
    plain, se, calib = {}, {}, {}
    for blk in sets:
        seen = set()
        for ln in blk.split("\n"):
            m = re.match(r"^(\S+)\s+dispatches=\s*(\d+) mean=\s*([\d.]+)", ln)
            if m:
                tgt = se if m.group(1) in seen else plain
                seen.add(m.group(1))
                tgt[m.group(1)] = (float(m.group(3)), int(m.group(2)))
            m = re.match(r"^\s+(calib_\w+) \(1 GiB\)\s+(\S+)\s+dispatches=\s*\d+ mean=\s*([\d.]+)", ln)
            if m:
                calib[(m.group(1), m.group(2))] = float(m.group(3))
    GiB = float(1 << 30)
    rd_b = GiB / calib[("calib_read", "TCC_EA0_RDREQ_sum")]
    wr_b = GiB / calib[("calib_write64", "TCC_EA0_WRREQ_sum")]
    out_b = 256 * 361 * 256 * 2
    w_b = 256 * 256 * 9 * 2
    # 20b x 256, SE every third block: 34 plain tower layers (20 without residual, 14 with) + the input convolution, 6 SE layers (with residual)
    algo_plain = (34 * (2 * out_b + w_b) + 14 * out_b + (256 * 361 * 64 * 2 + out_b + 64 * 256 * 9 * 2)) / 35.0
    algo_se = 3 * out_b + w_b + (768 * 64 + 64 * 512) * 2
    res = {}
    for name, d, algo in (("conv3x3_tower", plain, algo_plain), ("conv3x3_tower_se", se, algo_se)):
        rd, wr = d["TCC_EA0_RDREQ_sum"][0] * rd_b, d["TCC_EA0_WRREQ_sum"][0] * wr_b
        res[name] = {"read_bytes": round(rd), "write_bytes": round(wr), "algorithmic_bytes": round(algo),
                     "tcc_hit_rate": round(d["TCC_HIT_sum"][0] / d["TCC_REQ_sum"][0], 4), "dispatches": d["TCC_EA0_RDREQ_sum"][1]}
    res["tower_run"] = {"layers": "35 plain (input convolution included) + 6 with the SE unit",
                        "read_bytes": 35 * res["conv3x3_tower"]["read_bytes"] + 6 * res["conv3x3_tower_se"]["read_bytes"],
                        "write_bytes": 35 * res["conv3x3_tower"]["write_bytes"] + 6 * res["conv3x3_tower_se"]["write_bytes"],
                        "algorithmic_bytes": round(35 * algo_plain + 6 * algo_se)}
    res["bytes_per_request"] = {"read": round(rd_b, 2), "write": round(wr_b, 2)}
    res["stored_bytes_check"] = {"plain_layer_stores": out_b, "measured_writes": res["conv3x3_tower"]["write_bytes"],
                                 "ratio": round(res["conv3x3_tower"]["write_bytes"] / out_b, 4)}
    res["source"] = ("rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum, one set per pass, per-layer launches (SAYURI_TOWER=0), "
                     "bytes per request calibrated on 1 GiB streams in the same passes")
    try:
        res["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:  # noqa: BLE001
        res["commit"] = "?"
    json.dump(res, open(dst + ".json", "w"), indent=1)
    with open(dst + ".txt", "w") as f:
        f.write("# HBM-side (fabric) traffic of the tower convolutions, MI355X, batch 256 x 19x19, 20b x 256 fp16 -- raw PMC passes below\n")
        f.write("# summary: " + json.dumps({k: res[k] for k in ("conv3x3_tower", "conv3x3_tower_se", "tower_run", "bytes_per_request", "stored_bytes_check")}) + "\n")
        f.write("# reading: the write counters now equal the stored bytes (739 328 requests x 64 B = 47.3 MB = 256 x 361 x 512 B); fabric reads are\n"
                "#   ~75 MB per plain layer = input + residual (14 of 34 layers) + what the XCD's 4 MiB L2 cannot keep of 5.9 MB per XCD per layer:\n"
                "#   activations come back from the Infinity Cache / HBM, not from L2 (round 2's 14.8 MB did not survive re-measurement);\n"
                "#   the weights (1.18 MB, read by all 256 workgroups) are what the 76 % L2 hit rate is.\n")
        f.write(text)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
