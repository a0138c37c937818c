# GPU box: A/B of range chains side by side (bench.py --qp 16, --calls-in-flight C --backs B) -> gpurun_out/backs/
set -o pipefail
O=gpurun_out/backs; mkdir -p $O
run() {  # config calls backs queues tag
  GPU_MAX_HW_QUEUES=$4 timeout -k 10 240 python bench.py --qp 16 --config $1 --steps ${STEPS:-6} --warmup 1 --no-cpu-baseline --no-host-boundary --calls-in-flight $2 --backs $3 > $O/$5.json 2> $O/$5.err || { tail -5 $O/$5.err; exit 1; }
  python - $O/$5.json <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], r["value"], r["ms_per_step"], r["config"]["range_coder"], r["config"]["coder_scratch_GB"], r["chain"]["ms"])
PY
}
for spec in $SPECS; do IFS=: read cfg c b q <<< "$spec"; run $cfg $c $b $q ${cfg}_c${c}_b${b}_q${q} || exit 1; done
