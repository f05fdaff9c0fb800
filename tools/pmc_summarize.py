#!/usr/bin/env python3
"""pmc_summarize.py DIR LABEL ARGS -- per-kernel means of the rocprofv3 counter CSVs written by tools/pmc_collect.sh.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (TCC_EA0_RDREQ x 64 B / 1024); they are converted to bytes here,
uncorrected: the x2 correction for wide coalesced streams (MI355X_MICROARCH.md) is applied by the reader per kernel."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

out_dir, label, args = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"blmx::(k_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][-48:]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out_dir + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = {}
for kname, counters in acc.items():
    d = {}
    for c, v in counters.items():
        # the same kernel runs in warm-up, timed and profiling steps with identical work: the mean over launches is per launch
        d[c] = sum(v) / len(v)
        d["launches"] = len(v)
    k = {"launches": d.get("launches", 0), "counters": {c: d[c] for c in d if c != "launches"}}
    k["fetch_bytes"] = d.get("FETCH_SIZE", 0.0) * 1024.0
    k["write_bytes"] = d.get("WRITE_SIZE", 0.0) * 1024.0
    k["valu_insts"] = d.get("SQ_INSTS_VALU", 0.0)
    k["l2_read_requests"] = d.get("TCP_TCC_READ_REQ_sum", 0.0)  # L1 -> L2 read requests per launch
    kernels[kname] = k
for f in glob.glob(out_dir + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r["Name"])
        if n in kernels:
            kernels[n]["trace_avg_ns"] = float(r["AverageNs"])
            kernels[n]["trace_calls"] = int(r["Calls"])
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(root, "blight_amd", "csrc", "*.h*"))):
    h.update(open(f, "rb").read())
ceil = None
try:
    for line in open(os.path.join(root, "profiles", "r01_gather_ceiling_8GiB.jsonl")):
        d = json.loads(line)
        if d["test"] == "random_64B_lines_independent":
            ceil = d["Glines_per_s"] * 1e9
except Exception:
    pass
print(json.dumps({"label": label, "bench_args": args, "source_digest": h.hexdigest()[:16], "kernels": kernels,
                  "gather_ceiling_lines_per_s": ceil,
                  "note": "means per launch over all dispatches of the run; FETCH_SIZE / WRITE_SIZE in bytes, uncorrected"}, indent=1))
