#!/usr/bin/env python3
"""Sample the shader clock (sysfs pp_dpm_sclk / rocm-smi) while a command runs.  Dev aid (GPU box).
usage: clock_watch.py <period_s> -- cmd ..."""
import glob, subprocess, sys, time, collections
per = float(sys.argv[1]); cmd = sys.argv[3:]
paths = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
print("paths", paths, flush=True)
p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
t0 = time.time(); samples = []
while p.poll() is None:
    cur = []
    for q in paths:
        try:
            for line in open(q):
                if "*" in line: cur.append(line.strip().replace(" *", ""))
        except OSError as e:
            cur.append("err")
    cur = " | ".join(cur)
    samples.append((round(time.time() - t0, 2), cur))
    time.sleep(per)
hist = collections.Counter(s for _, s in samples)
print("histogram", dict(hist))
# timeline compressed: print changes
last = None
for t, s in samples:
    if s != last: print(t, s); last = s
