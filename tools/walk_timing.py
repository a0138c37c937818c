#!/usr/bin/env python3
"""Per-wavefront life and placement of the column-walking T-stage (development aid, GPU box only).
usage: python tools/phase_timing.py build ; FFV2AMD_TSTAGE=1 python tools/walk_timing.py"""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "phase", "libffv2amd_timing.so")
import ffmpeg_ffv2_amd._lib as L
L.SO = SO
from ffmpeg_ffv2_amd import FFV2Encoder, frames as synth
import torch
W, H, fmt, nf = 3840, 2160, "yuv444p10le", 8
enc = FFV2Encoder(W, H, fmt, max_batch=nf)
fr = np.stack([synth.make("S1" if n % 2 == 0 else "S2", n, 3, H, W, 10) for n in range(nf)])
dev = enc.upload(fr)
coef = torch.empty((nf, enc.info.block_planes, 4096), dtype=torch.int32, device=dev.device)
enc.set_coef_sink(coef)
pk = enc.alloc_packets(nf)
stream = torch.cuda.current_stream(dev.device).cuda_stream
lib = L.load()
lib.ffv2amd_debug_phase_alloc.argtypes = [C.c_size_t]
groups = 65536
assert lib.ffv2amd_debug_phase_alloc(groups) == 0
for _ in range(150):
    enc.encode_batch_device(dev, out=pk, stream=stream)
torch.cuda.synchronize()
raw = np.zeros((groups, 8), dtype=np.uint64)
lib.ffv2amd_debug_phase_raw.argtypes = [C.c_void_p]
lib.ffv2amd_debug_phase_raw(raw.ctypes.data)
live = raw[:, 0] != 0
r = raw[live].astype(np.int64)
st, hw, du = r[:, 0], r[:, 1], r[:, 2]
span = (st + du).max() - st.min()
print("waves %d, launch span %.1f us, life us p5/50/95/max %s, start spread us %.1f" % (
    live.sum(), span / 100.0, np.round(np.percentile(du, [5, 50, 95, 100]) / 100.0, 1), (st.max() - st.min()) / 100.0))
nb = r[:, 6]
print("blocks per wave min/max", nb.min(), nb.max())
print("ticks per block: front %.0f back %.0f prestep(per wave) %.0f" % (
    (r[:, 3] / np.maximum(nb, 1)).mean(), (r[:, 4] / np.maximum(nb, 1)).mean(), r[:, 5].mean()))
# HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9) ; XCC via a different register
cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1; simd = (hw >> 4) & 3
key = (se * 2 + sh) * 16 + cu
import collections
cnt = collections.Counter(zip(key.tolist(), simd.tolist()))
print("waves per (se,sh,cu,simd) histogram (all XCDs folded):", collections.Counter(cnt.values()))
end = st + du - st.min()
for q in (50, 90, 99, 100):
    print("finish time p%d: %.1f us" % (q, np.percentile(end, q) / 100.0))
# life by number of waves sharing the simd slot key
idx = np.nonzero(live)[0]
def grp(name, k):
    out = []
    for v in np.unique(k):
        m = k == v
        out.append("%s:%.0f(n=%d)" % (v, du[m].mean() / 100.0, m.sum()))
    print(name, " ".join(out))
grp("life by xcd(blockIdx%8)", idx % 8)
grp("life by se", se)
grp("life by simd", simd)
grp("life by cu", cu)
grp("life by blockIdx/256", idx // 256)
grp("life by wave slot", hw & 15)
# partners: waves on same (xcd, se, sh, cu, simd)
xcd = idx % 8
pk = ((xcd * 8 + se) * 2 + sh) * 16 + cu
full = pk * 4 + simd
c2 = collections.Counter(full.tolist())
alone = np.array([c2[v] for v in full.tolist()])
grp("life by #waves on my simd", alone)
ccu = collections.Counter(pk.tolist())
percu = np.array([ccu[v] for v in pk.tolist()])
grp("life by #waves on my cu", percu)
slow = du > np.percentile(du, 85)
print("slow waves: xcd hist", collections.Counter(xcd[slow].tolist()))
print("slow waves: se hist", collections.Counter(se[slow].tolist()))
print("slow: front/back ticks per block %.0f %.0f ; fast: %.0f %.0f" % (
    (r[slow, 3] / nb[slow]).mean(), (r[slow, 4] / nb[slow]).mean(), (r[~slow, 3] / nb[~slow]).mean(), (r[~slow, 4] / nb[~slow]).mean()))
