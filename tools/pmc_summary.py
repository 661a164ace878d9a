#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc CSVs written by tools/pmc_passes.sh: per kernel, per counter, the
mean over dispatches.  FETCH_SIZE is doubled for gfx950 wide streaming reads and both sizes are
converted from KB to bytes, as MI355X_MICROARCH.md (HBM section) prescribes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = defaultdict(float)
    names = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            key = (f, row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            # launches of one kernel with different grids are different workloads (the update kernel runs
            # G x updates-per-launch workgroups): the grid size is part of the key
            names[(f, row["Dispatch_Id"])] = "%s [grid %s]" % (row["Kernel_Name"].split("(")[0][:60], row.get("Grid_Size", "?"))
    for (ff, d, c), v in per_dispatch.items():
        acc[names[(ff, d)]][c].append(v)
out = {}
for k, cs in sorted(acc.items()):
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
for k, cs in out.items():
    if "FETCH_SIZE" in cs:
        cs["hbm_read_bytes"] = cs["FETCH_SIZE"] * 1024 * 2      # KB -> B, x2 (gfx950 128-B requests tallied at 64 B)
    if "WRITE_SIZE" in cs:
        cs["hbm_write_bytes"] = cs["WRITE_SIZE"] * 1024
print(json.dumps(out, indent=1))
