#!/bin/bash
# GPU box: T-stage time of the column-walking kernel for a few run-length schedules vs the one-block kernel
export FFV2AMD_TSTAGE=1
run() { python bench.py --no-cpu-baseline --steps 100 $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['roofline']['frac'])"; }
run "walk default C3"
for t in "22,1,1,1" "16,5,2,1" "21,1,1,1"; do FFV2AMD_WALK_TIERS=$t run "walk $t"; done
run "walk default C3 again"
FFV2AMD_TSTAGE=0 run "block C3"
for c in C2 C4 C5; do run "walk $c" "--config $c"; FFV2AMD_TSTAGE=0 run "block $c" "--config $c"; done
