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
# The measurement build (-DRP_MEASUREMENT): the same library plus the benchmark-only knobs (rp_set_*), the sweep-only
# kernel instantiations and the ablation kernels the scripts under scripts/ drive.  Never loaded by the product unless
# RECTIPY_AMD_MEASUREMENT=1 is set (rectipy_amd/_lib.py); built on demand: python -m rectipy_amd.build --measurement
MEASURE_LIB_PATH = os.path.join(OUT_DIR, "librectipy_hip_measure.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: every fma in the sources is explicit; the oracle is built the same
# way, which is what makes HIP-vs-oracle comparisons bit-exact.
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
    "-Wall", "-Wno-unused-function",
]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(ROOT, "include")
    deps += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    deps.append(os.path.abspath(__file__))
    return deps


def _deps():
    return sources() + _headers()


def is_stale(measurement=False):
    lib = MEASURE_LIB_PATH if measurement else LIB_PATH
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=True, measurement=False):
    """Compile every HIP source (one object per source, the stale ones in parallel) and link them into one shared
    library; returns its path.  ``measurement``: the -DRP_MEASUREMENT build (see MEASURE_LIB_PATH)."""
    lib = MEASURE_LIB_PATH if measurement else LIB_PATH
    if not force and not is_stale(measurement):
        return lib
    obj_dir = os.path.join(OUT_DIR, "obj_measure" if measurement else "obj")
    os.makedirs(obj_dir, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in _headers())
    compile_flags = [f for f in FLAGS if f != "-shared"] + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    if measurement:
        compile_flags.append("-DRP_MEASUREMENT")
    jobs, objects = [], []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objects.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            cmd = [HIPCC] + compile_flags + ["-c", src, "-o", obj]
            if verbose:
                print("[rectipy_amd.build]", " ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objects
    if verbose:
        print("[rectipy_amd.build]", " ".join(link), flush=True)
    subprocess.run(link, check=True)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, measurement="--measurement" in sys.argv))
