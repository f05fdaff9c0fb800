#!/usr/bin/env bash
# pmc_query.sh -- collect PMC counters for the k_query kernel of the default bench (one rocprofv3 pass per counter
# group: FETCH_SIZE costs 3 of the 4 TCC slots, so it cannot share a pass with WRITE_SIZE; see MI355X_MICROARCH.md).
# Counters only (no --kernel-trace / --stats in the same run).  Output: gpurun_out/pmc/<pass>/.../counter_collection.csv
# and a per-counter summary in gpurun_out/pmc/summary.txt.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="${BENCH_ARGS:---cpu-seconds 0 --steps 2 --warmup 1}"
i=0
while read -r counters; do
	[ -z "$counters" ] && continue
	i=$((i + 1))
	timeout -k 10 240 rocprofv3 --pmc $counters --kernel-include-regex "k_query|k_bin_hist|k_bin_scatter|k_gather_ids|rocprim" --output-format csv -d "$OUT/p$i" -- \
		python3 "$R/bench.py" $ARGS >"$OUT/p$i.json" 2>"$OUT/p$i.log" || echo "pass $i failed" >>"$OUT/summary.txt"
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_READ_sum TCC_TAG_STALL_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
LIST
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"] + " " + r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "a") as fh:
    for k, v in acc.items():
        fh.write("%-90s launches=%d mean=%.6g min=%.6g max=%.6g\n" % (k, len(v), sum(v) / len(v), min(v), max(v)))
print(open(out + "/summary.txt").read())
PY
