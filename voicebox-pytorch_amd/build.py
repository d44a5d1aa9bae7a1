"""Builds libvbx_hip.so (gfx950 only) in-tree with hipcc.  `python voicebox-pytorch_amd/build.py`"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libvbx_hip.so")
SOURCES = ["api.hip", "gemm.hip", "gemm3.hip", "gemm4.hip", "attn.hip", "norm.hip", "gateloop.hip", "ops.hip", "runtime.hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "vbx.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(HERE, "lib", s.replace(".hip", ".o"))
        objs.append(obj)
        shared = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))]  # headers / included bodies
        shared.append(os.path.join(HERE, "..", "include", "vbx.h"))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max([os.path.getmtime(src)] + [os.path.getmtime(f) for f in shared]):
            continue
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
