#!/bin/bash
# instruction-fetch side counters of the T-stage (run on the GPU box)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/prof_${1:-ic}; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_sq" -o p -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc.err"
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_INT32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OUT/pmc_lds" -o p -- python3 "$REPO/bench.py" $ARGS > /dev/null 2>> "$OUT/pmc.err"
cd "$REPO" && python3 tools/summarize_prof.py "$OUT"
