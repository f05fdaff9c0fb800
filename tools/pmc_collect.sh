#!/usr/bin/env bash
# pmc_collect.sh LABEL [bench args...] -- rocprofv3 PMC passes over the bench command (counters only: no --kernel-trace / --stats
# in the same run; FETCH_SIZE takes 3 of the 4 TCC slots, so it cannot share a pass with WRITE_SIZE -- MI355X_MICROARCH.md),
# then one kernel-trace pass for the durations.  Per-dispatch CSVs go to gpurun_out/pmc_LABEL/, the per-kernel summary
# (mean per launch of the timed-region dispatches) to gpurun_out/pmc_LABEL.json: copy that file to profiles/rNN_pmc_<workload>.json.
set -u
LABEL="$1"; shift
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/pmc_$LABEL"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="${*:---cpu-seconds 0 --reads 0 --steps 4 --warmup 1 --no-fasta-leg --no-excursion}"
RE="k_query|k_bin_hist|k_bin_scatter|k_bin_onepass|k_plan_|k_gather_ids|k_gather_runs|k_read_front|k_read_query"
i=0
while read -r counters; do
	[ -z "$counters" ] && continue
	i=$((i + 1))
	timeout -k 10 300 rocprofv3 --pmc $counters --kernel-include-regex "$RE" --output-format csv -d "$OUT/p$i" -- \
		python3 "$R/bench.py" $ARGS >"$OUT/p$i.json" 2>"$OUT/p$i.log" || echo "pass $i ($counters) failed" >>"$OUT/failed.txt"
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_IFETCH
LIST
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" $ARGS >"$OUT/trace.json" 2>"$OUT/trace.log" \
	|| echo "trace pass failed" >>"$OUT/failed.txt"
python3 "$R/tools/pmc_summarize.py" "$OUT" "$LABEL" "$ARGS" >"$R/gpurun_out/pmc_$LABEL.json"
cat "$R/gpurun_out/pmc_$LABEL.json"
