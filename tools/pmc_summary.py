#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch, per kernel.

usage: tools/pmc_summary.py <dir-with-counter_collection-csvs>... > profiles/rNN_pmc_summary.txt
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB on gfx950 (MI355X_MICROARCH.md, HBM section); this tool
also prints them in MiB per dispatch.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(dirs):
    acc = defaultdict(list)
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"].split("(")[0]
                    if not name.startswith("k_"):
                        continue
                    acc[(name, row["Counter_Name"])].append(float(row["Counter_Value"]))
    print("%-24s %-24s %8s %18s" % ("kernel", "counter", "launches", "mean per launch"))
    for (k, c), v in sorted(acc.items()):
        mean = sum(v) / len(v)
        extra = "   (%.1f MiB)" % (mean / 1024.0) if c in ("FETCH_SIZE", "WRITE_SIZE") else ""
        print("%-24s %-24s %8d %18.1f%s" % (k, c, len(v), mean, extra))


if __name__ == "__main__":
    main(sys.argv[1:] or ["."])
