#!/bin/bash
# GPU box: SQ counters of the lane coder's kernels (tools/lanecoder_big.py 64: one group of lanes) -> gpurun_out/pmc_lc_<tag>/summary.txt
set -o pipefail
TAG=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_lc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -o p -- python3 "$REPO/tools/lanecoder_big.py" 64 > "$OUT/out.txt" 2> "$OUT/err.txt" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "lc_" not in k and "pvq" not in k:
        continue
    import re
    m = re.search(r"(lc_\w+|ffv2_pvq_kernel)", k)
    name = m.group(1) if m else k[:30]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    did = (name, r["Dispatch_Id"])
    if did not in seen:
        seen.add(did); cnt[name] += 1
with open(out + "/summary.txt", "w") as o:
    for name, c in acc.items():
        n = cnt[name]
        line = "%-22s launches %4d" % (name, n) + "".join("  %s %.3g" % (k, v / n) for k, v in sorted(c.items()))
        print(line); o.write(line + "\n")
        if c.get("SQ_INSTS_VALU") and c.get("SQ_WAVE_CYCLES"):
            extra = "    wave-cycles (quad cycles x4) per VALU instruction: %.2f; waves busy per launch: see SQ_WAVE_CYCLES" % (4 * c["SQ_WAVE_CYCLES"] / c["SQ_INSTS_VALU"])
            print(extra); o.write(extra + "\n")
PY
find "$OUT" -name "*.csv" -size +8M -delete
