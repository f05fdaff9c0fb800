#!/usr/bin/env bash
# pmc_ab.sh LABEL [bench args...] -- the fabric / L2 counters of the query pipeline's kernels for ONE variant (environment variables
# select it), two rocprofv3 PMC passes + per-kernel means: what tools/pmc_collect.sh collects, cut down to the counters an A/B of
# index layouts needs (requests to the fabric, L2 hits and misses, L1 -> L2 requests).  Output: gpurun_out/pmcab_LABEL.txt
set -u
LABEL="$1"; shift
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/pmcab_$LABEL"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="${*:---cpu-seconds 0 --reads 0 --steps 4 --warmup 2 --no-fasta-leg --no-excursion --no-host-legs}"
RE="k_query_slots|k_bin_onepass|k_gather_runs"
i=0
while read -r counters; do
	[ -z "$counters" ] && continue
	i=$((i + 1))
	timeout -k 10 200 rocprofv3 --pmc $counters --kernel-include-regex "$RE" --output-format csv -d "$OUT/p$i" -- \
		python3 "$R/bench.py" $ARGS >"$OUT/p$i.json" 2>"$OUT/p$i.log" || echo "pass $i ($counters) failed" >>"$OUT/failed.txt"
done <<'LIST'
TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
LIST
python3 - "$OUT" > "$R/gpurun_out/pmcab_$LABEL.txt" <<'PY'
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"blmx::(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
        acc[m.group(1) if m else r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k, " ".join("%s=%.4g(n%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
PY
cat "$R/gpurun_out/pmcab_$LABEL.txt"
