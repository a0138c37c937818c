# GPU box: the qp > 0 ring at small batches, calls in flight / range chains side by side -> gpurun_out/backs/
set -o pipefail
O=gpurun_out/backs; mkdir -p $O
if [ -z "$SKIPTESTS" ]; then timeout -k 10 400 python -m pytest tests/test_qpring_gpu.py tests/test_codec_fanout_gpu.py tests/test_lanecoder_gpu.py -m gpu -x -q > $O/tests_ring.log 2>&1 || { tail -20 $O/tests_ring.log; exit 1; }; fi
[ -z "$SKIPTESTS" ] && tail -1 $O/tests_ring.log
for spec in $SPECS; do IFS=: read per calls backs <<< "$spec"
  FFV2AMD_QPRING_CALLS=$calls FFV2AMD_LC_BACKS=$backs GPU_MAX_HW_QUEUES=${Q:-16} timeout -k 10 300 python bench.py --qp 16 --config C2 --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 2048 --host-frames-per-call $per > $O/ring_${per}_c${calls}_b${backs}_q${Q:-16}.json 2> $O/ring_${per}_c${calls}_b${backs}_q${Q:-16}.err || { tail -5 $O/ring_${per}_c${calls}_b${backs}_q${Q:-16}.err; exit 1; }
  python - $O/ring_${per}_c${calls}_b${backs}_q${Q:-16}.json <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
hb=r["host_boundary"]
print(sys.argv[1], hb["frames_per_call"], hb["frames_sent"], {k:v["Mpix_s"] for k,v in hb.items() if isinstance(v,dict)})
PY
done
