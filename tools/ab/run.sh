SO=ffmpeg_ffv2_amd/libffv2amd.so
cp $SO /tmp/keep.so
for v in 0 1 2 4 8 15; do
  cp tools/ab/lib_$v.so $SO
  bash tools/profile_lanecoder.sh scat_$v 1024 > /dev/null 2>&1
  python3 - $v <<PY
import csv,sys
rows=list(csv.DictReader(open("gpurun_out/stats_lc_scat_%s/kernel_stats.csv"%sys.argv[1])))
for r in rows:
    if "scatter" in r["Name"] or "pvq" in r["Name"] or "lc_cdf" in r["Name"]: print(sys.argv[1], r["Name"][22:48], r["Calls"], round(int(r["TotalDurationNs"])/1e6,1), "ms total", round(float(r["AverageNs"])/1e3,1), "us avg")
PY
done
cp /tmp/keep.so $SO
