#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of bench.py's lane coder mode -> gpurun_out/stats_lcb_<tag>/{kernel_stats.csv,timeline.txt}
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/stats_lcb_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$REPO/bench.py" --no-cpu-baseline --qp 16 --config C2 $* > "$OUT/bench.json" 2> "$OUT/err.txt" || exit 1
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 "$REPO/tools/lc_timeline.py" "$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)" > "$OUT/timeline.txt" 2>&1
find "$OUT/trace" -name "*.csv" -size +4M -delete
cat "$OUT/bench.json" | cut -c1-300
