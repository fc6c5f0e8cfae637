#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel family: mean counter value per dispatch.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write ... [--config JSON] [--command STR] [--out FILE]
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts wide coalesced reads at half
their bytes (MI355X_MICROARCH.md, HBM section) -> the 'hbm_read_bytes' column doubles it.  mfma_busy_frac =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 4 SIMDs x 256 CUs), both collected in the same pass."""
import argparse
import collections
import csv
import glob
import json
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--config", default="{}")
    ap.add_argument("--command", default="")
    ap.add_argument("--note", default="")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in a.dirs:
        for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
                out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in sorted(out.items()):
        res[k] = {c: {"dispatches": len(v), "mean": sum(v) / len(v), "sum": sum(v)} for c, v in cs.items()}
        if "FETCH_SIZE" in cs:
            res[k]["hbm_read_bytes_mean"] = 2 * 1024 * res[k]["FETCH_SIZE"]["mean"]
        if "WRITE_SIZE" in cs:
            res[k]["hbm_write_bytes_mean"] = 1024 * res[k]["WRITE_SIZE"]["mean"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs and res[k]["GRBM_GUI_ACTIVE"]["mean"] > 0:
            res[k]["mfma_busy_frac"] = round(res[k]["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (res[k]["GRBM_GUI_ACTIVE"]["mean"] / 8 * 4 * 256), 4)
    doc = {"command": a.command, "config": json.loads(a.config), "note": a.note, "kernels": res}
    text = json.dumps(doc, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
