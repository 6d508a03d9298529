"""In-tree build of the gfx950 C-ABI library (``rectipy_amd/_C/librectipy_hip.so``).

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting ``.so`` travels to the GPU box with the repository snapshot.

    python -m rectipy_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_C")
LIB_PATH = os.path.join(OUT_DIR, "librectipy_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: every fma in the sources is explicit; the oracle is built the same
# way, which is what makes HIP-vs-oracle comparisons bit-exact.
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
    "-Wall", "-Wno-unused-function",
]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    deps = sources()
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(ROOT, "include")
    deps += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    deps.append(os.path.abspath(__file__))
    return deps


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=True):
    """Compile every HIP source into one shared library; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", LIB_PATH] + sources()
    if verbose:
        print("[rectipy_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
