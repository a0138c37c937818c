"""Builds libffv2amd.so (HIP kernels + C-ABI + AVCodec-shaped host shim) in-tree for gfx950."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
SO = os.path.join(PKG, "libffv2amd.so")

HIP_SOURCES = ["ffv2_kernels.hip", "ffv2_pvq.hip", "ffv2_inverse.hip", "ffv2_upconv.hip", "ffv2_wide.hip", "ffv2_rangecoder.hip", "ffv2_lanecoder.hip", "ffv2_capi.cpp"]
C_SOURCES = ["ffv2enc_amd.c", "ffv2mkv.c"]
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-strict-aliasing",
            "-ffp-contract=off", "-Wall"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X build needs the ROCm toolchain")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.startswith("gen")]
    deps += [os.path.join(CSRC, "gen", f) for f in os.listdir(os.path.join(CSRC, "gen"))]
    deps += [os.path.join(ROOT, "include", h) for h in ("ffv2_amd.h", "ffv2_amd_codec.h", "ffv2_amd_mkv.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force=False, verbose=False):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_code.py")], check=True)
    if not force and not needs_build():
        return SO
    objs = []
    for c in C_SOURCES:
        src = os.path.join(CSRC, c)
        if not os.path.exists(src):
            continue
        obj = os.path.join(CSRC, c.replace(".c", ".o"))
        subprocess.run(["gcc", "-O2", "-fPIC", "-std=gnu11", "-Wall", "-I", os.path.join(ROOT, "include"),
                        "-c", src, "-o", obj], check=True)
        objs.append(obj)
    hipcc = _hipcc()
    for src in HIP_SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + HIPFLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
