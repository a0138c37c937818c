#!/bin/bash
# GPU box: SQ counters of ffv2_pvq_kernel (16 1080p noise frames per launch) -> gpurun_out/pvq_pmc.txt
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pvq_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/a" -o p -- python3 "$REPO/tools/pvq_profile.py" > "$OUT/a.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/b" -o p -- python3 "$REPO/tools/pvq_profile.py" > "$OUT/b.log" 2>&1 || exit 1
cd "$REPO" && python3 - "$OUT" <<'PY' | tee "$OUT/../pvq_pmc.txt"
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for sub in ("a", "b"):
    f = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "ffv2_pvq_" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c, v in acc.items():
        v = v[-5:]
        print("%-24s %.4g per launch (24 480 block-planes)  %.1f per block-plane" % (c, sum(v) / len(v), sum(v) / len(v) / 24480))
PY
find "$OUT" -name "*.csv" -size +4M -delete
