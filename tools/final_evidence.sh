#!/usr/bin/env bash
# final_evidence.sh -- what a round's last tree is judged by, in one gpurun call: the whole GPU suite, the default bench line, the C5, light and
# skewed lines, PMC passes (tools/pmc_collect.sh) and rocprofv3 kernel statistics of the headline steps.  Everything under gpurun_out/final/.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O="$R/gpurun_out/final"
mkdir -p "$O"
cd "$R"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q >"$O/gputests.log" 2>&1 || { echo "gpu tests failed"; tail -n 20 "$O/gputests.log"; exit 1; }
tail -n 1 "$O/gputests.log"
timeout -k 10 300 python3 bench.py >"$O/bench_c4.json" 2>"$O/bench_c4.err" || { echo "default bench failed"; exit 1; }
echo "c4 done"
timeout -k 10 300 python3 bench.py --workload C5 >"$O/bench_c5.json" 2>"$O/bench_c5.err" || { echo "c5 bench failed"; exit 1; }
echo "c5 done"
timeout -k 10 300 python3 bench.py --light --cpu-seconds 0 --reads 0 --no-fasta-leg --no-host-legs >"$O/bench_c4_light.json" 2>"$O/bench_c4_light.err" || echo "light bench failed"
echo "light done"
bash tools/pmc_collect.sh c4 >"$O/pmc_c4.log" 2>&1 && echo "pmc c4 done"
bash tools/pmc_collect.sh c5 --workload C5 --cpu-seconds 0 --reads 0 --steps 4 --warmup 1 --no-fasta-leg --no-excursion >"$O/pmc_c5.log" 2>&1 && echo "pmc c5 done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ks_c4" -- python3 "$R/bench.py" --cpu-seconds 0 --reads 0 --no-host-legs --no-fasta-leg --no-excursion >"$O/ks_c4.json" 2>"$O/ks_c4.log" && echo "kernel stats c4 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ks_c5" -- python3 "$R/bench.py" --workload C5 --cpu-seconds 0 --reads 0 --no-host-legs --no-fasta-leg --no-excursion >"$O/ks_c5.json" 2>"$O/ks_c5.log" && echo "kernel stats c5 done"
find "$O" -name "*_kernel_trace.csv" -delete  # (the per-dispatch traces are large; the stats are what is kept)
ls "$O"
