"""Builds curla_amd/libcurla_hip.so (the C-ABI HIP library) in-tree with hipcc.

The library is stamped with a hash of the sources it was compiled from
(``libcurla_hip.so.srchash``, next to it): ``build_library`` recompiles when the
stamp and the sources disagree (not on mtimes, which a checkout or a copy to the
GPU box rewrites), and ``_lib.load`` refuses a library whose stamp is stale."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcurla_hip.so")
STAMP = LIB + ".srchash"
HEADER = os.path.join(HERE, "..", "include", "curla_hip.h")
SOURCES = ["conv.hip", "gemm.hip", "heads.hip", "augment.hip", "options.hip"]
ARCH = "gfx950"  # MI355X only
FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC"]


def _compiler_id():
    """The compiler the stamp belongs to: HIPCC's path and the ROCm release next to it (read from its version file:
    running `hipcc --version` at every import would cost a second and need the compiler at run time)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
    ver = "unknown"
    for name in (os.path.join(root, ".info", "version"), "/opt/rocm/.info/version"):
        try:
            with open(name) as f:
                ver = f.read().strip()
            break
        except OSError:
            continue
    return hipcc + " | rocm " + ver


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources, the shared header, the C-ABI header, the flags and the
    compiler (HIPCC's path and the ROCm release it belongs to: another release is another build)."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    h.update(_compiler_id().encode())
    for path in sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [HEADER]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_hash():
    try:
        with open(STAMP) as f:
            return f.read().strip()
    except OSError:
        return None


def is_stale():
    return not os.path.exists(LIB) or built_hash() != source_hash()


def build_library(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
