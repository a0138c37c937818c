import json, os, shutil, subprocess, sys
ROOT='.'
SO=os.path.join(ROOT,"ffmpeg_ffv2_amd","libffv2amd.so")
libs=sys.argv[1:]
keep=SO+".keep"; shutil.copy(SO,keep)
res={l:[] for l in libs}
try:
    for _ in range(int(os.environ.get("AB_ROUNDS","2"))):
        for lib in libs:
            shutil.copy(lib,SO)
            out=subprocess.run([sys.executable,"bench.py","--no-cpu-baseline","--no-host-boundary","--steps","100"],capture_output=True,text=True,check=True).stdout
            d=json.loads(out.strip().splitlines()[-1]); res[lib].append(d["roofline"]["kernel_ms_avg"])
finally:
    shutil.copy(keep,SO); os.remove(keep)
for l in libs: print(os.path.basename(l), res[l])
