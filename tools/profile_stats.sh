#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of bench.py for one config -> gpurun_out/stats_<tag>/ (kernel stats CSV only)
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/stats_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$REPO/bench.py" --no-cpu-baseline --no-host-boundary $* > "$OUT/bench.json" 2> "$OUT/err.txt" || exit 1
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/trace" -name "*.csv" -size +4M -delete
head -4 "$OUT/kernel_stats.csv" | cut -c1-160
