#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of tools/lanecoder_big.py -> gpurun_out/stats_lc_<tag>/kernel_stats.csv
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/stats_lc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$REPO/tools/lanecoder_big.py" $* > "$OUT/out.txt" 2> "$OUT/err.txt" || exit 1
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/trace" -name "*.csv" -size +4M -delete
cat "$OUT/out.txt"
cut -d, -f1-4 "$OUT/kernel_stats.csv" | cut -c1-150 | head -16
