"""Per-kernel summary of rocprofv3 --pmc passes (counter_collection.csv files) as JSON / text (development aid).

    python tools/pmc_summary.py out.json label=dir [label=dir ...]

Every directory holds ONE pass (its own set of counters); per kernel name the dispatches are averaged.  Derived:
  hbm_bytes_per_launch_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950: FETCH_SIZE reports half of a wide
  coalesced read stream, /opt/skills/guides/MI355X_MICROARCH.md, HBM section; WRITE_SIZE is exact),
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs),
  lds_conflict_ratio = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE,
  l2_hit_rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum).
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("gprx::", "").replace("(anonymous namespace)::", "").replace(" ", "")


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for item in dirs:
        for f in glob.glob(item.split("=", 1)[1] + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kernels = {}
    for name, counters in sorted(acc.items()):
        if name.startswith("__amd"):
            continue
        k = {"dispatches": max(len(v) for v in counters.values())}
        for c, vals in counters.items():
            k[c + "_avg"] = sum(vals) / len(vals)
        if "FETCH_SIZE_avg" in k and "WRITE_SIZE_avg" in k:
            k["hbm_bytes_per_launch_corrected"] = (2 * k["FETCH_SIZE_avg"] + k["WRITE_SIZE_avg"]) * 1024
            k["hbm_bytes_per_launch_raw"] = (k["FETCH_SIZE_avg"] + k["WRITE_SIZE_avg"]) * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES_avg" in k and "GRBM_GUI_ACTIVE_avg" in k:
            k["mfma_util"] = k["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / (k["GRBM_GUI_ACTIVE_avg"] / 8 * 1024)
        if "SQ_LDS_BANK_CONFLICT_avg" in k and k.get("SQ_LDS_IDX_ACTIVE_avg"):
            k["lds_conflict_ratio"] = k["SQ_LDS_BANK_CONFLICT_avg"] / k["SQ_LDS_IDX_ACTIVE_avg"]
        if "TCC_HIT_sum_avg" in k and "TCC_MISS_sum_avg" in k and k["TCC_HIT_sum_avg"] + k["TCC_MISS_sum_avg"] > 0:
            k["l2_hit_rate"] = k["TCC_HIT_sum_avg"] / (k["TCC_HIT_sum_avg"] + k["TCC_MISS_sum_avg"])
        kernels[name] = k
    json.dump({"source": " ".join(dirs), "units": "FETCH_SIZE / WRITE_SIZE in KB per dispatch as reported", "kernels": kernels}, open(out_path, "w"), indent=1)
    for name, k in kernels.items():
        print(name[:70], {a: (round(b, 3) if isinstance(b, float) and b < 100 else (f"{b:.3e}" if isinstance(b, float) else b)) for a, b in k.items()
                          if a in ("dispatches", "hbm_bytes_per_launch_corrected", "mfma_util", "lds_conflict_ratio", "l2_hit_rate")})


if __name__ == "__main__":
    main()
