#!/bin/bash
# GPU box: T-stage time of the column-walking kernel for a few segment fractions vs the one-block kernel
export FFV2AMD_TSTAGE=1
run() { python bench.py --no-cpu-baseline --steps 100 $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['roofline']['frac'])"; }
for c in C3 C2; do
for f in "0.5 0.7" "0.55 0.7" "0.6 0.7" "0.55 0.5" "0.55 0.9" "0.45 0.6"; do set -- $f; FFV2AMD_WALK_SEG=$1 FFV2AMD_WALK_CAP=$2 run "walk $c seg $1 cap $2" "--config $c"; done
FFV2AMD_TSTAGE=0 run "block $c" "--config $c"
done
run "walk C3 1 frame" "--frames-per-step 1"; FFV2AMD_TSTAGE=0 run "block C3 1 frame" "--frames-per-step 1"
run "walk C3 2 frame" "--frames-per-step 2"; FFV2AMD_TSTAGE=0 run "block C3 2 frame" "--frames-per-step 2"
run "walk C3 32 frame" "--frames-per-step 32"
