#!/bin/bash
# GPU box: HBM traffic of the T-stage per launch for one bench config, from two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE: MI355X_MICROARCH.md, HBM / rocprofv3 section) -> gpurun_out/traffic_<config>.json,
# to be copied to profiles/ (bench.py reads profiles/traffic_<config>.json for roofline.traffic).
# Usage: tools/profile_traffic.sh C2|C3|C4|C5
set -o pipefail
CFG=${1:-C3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/traffic_$CFG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--config $CFG --no-cpu-baseline --no-host-boundary --steady-seconds 0 --steps 20 --warmup 5"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err" || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err" || exit 1
cd "$REPO" && python3 - "$OUT" "$CFG" <<'PY'
import csv, glob, json, os, sys
out, cfg = sys.argv[1], sys.argv[2]
vals = {}
kern = None
for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    v = []
    for row in csv.DictReader(open(f)):
        if "tstage" in row["Kernel_Name"] and row["Counter_Name"] == name:
            v.append(float(row["Counter_Value"]))
            kern = row["Kernel_Name"]
    v = v[-20:]
    vals[name] = sum(v) / len(v)
b = json.load(open(os.path.join(out, "bench_fetch.json")))
res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, mean of the last 20 launches) of "
                 "`python3 bench.py --config %s --no-cpu-baseline --no-host-boundary --steady-seconds 0 --steps 20 --warmup 5`, MI355X "
                 "(tools/profile_traffic.sh)" % cfg,
       "kernel": kern,
       "frames_per_launch": b["config"]["frames_per_step_per_gpu"], "coef_writeback": b["config"]["coef_writeback"],
       "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact for 16-B/lane stores",
       "hbm_bytes_per_launch": int(round((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)),
       "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"]}
res["ratio_to_algorithmic"] = round(res["hbm_bytes_per_launch"] / res["algorithmic_bytes_per_launch"], 4)
json.dump(res, open(os.path.join(out, "..", "traffic_%s.json" % cfg), "w"), indent=1)
print(json.dumps(res))
PY
find "$OUT" -name "*.csv" -size +4M -delete
