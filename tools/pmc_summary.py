"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (short name), per-launch averages of every
counter plus the derived figures used in DESIGN.md / profiles/:
  mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
  hbm_read_bytes     = FETCH_SIZE x 1024 x 2      (KiB units; gfx950 undercounts reads by 2 -- MI355X_MICROARCH.md)
  hbm_write_bytes    = WRITE_SIZE x 1024
Usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> [substring filter ...] > out.json"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:120]


def main():
    root = sys.argv[1]
    filt = sys.argv[2:]
    csv.field_size_limit(1 << 30)
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    dur = defaultdict(float)
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if filt and not any(s in k for s in filt):
                    continue
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                did = (path, row["Dispatch_Id"])
                if did not in launches[k]:
                    launches[k].add(did)
                    dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    out = {}
    for k, ctrs in acc.items():
        n = len(launches[k])
        e = {c: v / n for c, v in ctrs.items()}
        e["launches"] = n
        e["avg_ns_under_pmc"] = dur[k] / n
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_fraction"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * e["GRBM_GUI_ACTIVE"] / 8)
            e["clock_GHz_under_pmc"] = e["GRBM_GUI_ACTIVE"] / 8 / e["avg_ns_under_pmc"]
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes"] = e["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
        out[k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
