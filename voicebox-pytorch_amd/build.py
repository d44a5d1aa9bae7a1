"""Builds libvbx_hip.so (gfx950 only) in-tree with hipcc.  `python voicebox-pytorch_amd/build.py`"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libvbx_hip.so")
SOURCES = ["api.hip", "gemm.hip", "gemm3.hip", "gemm4.hip", "gemm5.hip", "attn.hip", "norm.hip", "gateloop.hip", "ops.hip", "precise.hip", "runtime.hip"]


# gemm5.hip: its epilogue is hand-slotted into the gaps of the MFMA stream; SLP-packed fp32 (v_pk_mul_f32 / v_pk_fma_f32) costs more beside
# MFMAs than the two scalar operations it replaces (MI355X_MICROARCH.md, per-instruction constants)
EXTRA_FLAGS = {"gemm5.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _hashed_files():
    """What the library is compiled from: the SOURCES, the headers / included bodies beside them, include/vbx.h -- not whatever
    else sits in csrc/ (editor backups, sub-directories)."""
    files = [os.path.join(CSRC, f) for f in SOURCES]
    files += sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc")))
    return [f for f in files if os.path.isfile(f)] + [os.path.join(HERE, "..", "include", "vbx.h")]


def source_hash():
    """sha1 over everything the library is compiled from; None when the sources are not there (an installed package that ships
    the library without csrc/): "cannot verify", not an error."""
    import hashlib

    if not os.path.isdir(CSRC):
        return None
    h = hashlib.sha1()
    for f in _hashed_files():
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


HASH_FILE = LIB + ".srchash"  # written next to the library: a .so that was not rebuilt after a source edit is refused at load time


def recorded_hash():
    try:
        return open(HASH_FILE).read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.exists(LIB) or recorded_hash() != source_hash()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    stamp = source_hash()  # before compiling: an edit made during the build must not be recorded as built
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
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj]
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
    with open(HASH_FILE, "w") as fh:
        fh.write(stamp + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
