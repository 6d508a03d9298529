"""Registers, scratch and kernel-argument bytes of the kernels in a built library (from the code-object metadata).
    python scripts/kernel_resources.py [path/to/lib.so] [name filter]
Prints every kernel whose name contains the filter (default: all with scratch) and the totals."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rectipy_amd", "_C", "librectipy_hip.so")
pattern = sys.argv[2] if len(sys.argv) > 2 else None
readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)]
    kernels = []
    for n, i in enumerate(starts):
        co = os.path.join(tmp, f"co{n}.elf")
        open(co, "wb").write(data[i:starts[n + 1] if n + 1 < len(starts) else len(data)])
        notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
        for block in notes.split("- .agpr_count")[1:]:
            get = lambda key: re.search(r"\." + key + r":\s+(\S+)", block)
            if get("name"):
                kernels.append(dict(name=get("name").group(1), scratch=int(get("private_segment_fixed_size").group(1)),
                                    sgpr=int(get("sgpr_count").group(1)), vgpr=int(get("vgpr_count").group(1)),
                                    kernarg=int(get("kernarg_segment_size").group(1)), lds=int(get("group_segment_fixed_size").group(1))))
for k in kernels:
    if (pattern and pattern in k["name"]) or (not pattern and k["scratch"]):
        print(f"{k['name'][:90]:90s} scratch {k['scratch']:4d}  sgpr {k['sgpr']:3d}  vgpr {k['vgpr']:3d}  lds {k['lds']:6d}  kernarg {k['kernarg']}")
print(f"{len(kernels)} kernels, {sum(1 for k in kernels if k['scratch'])} with scratch, library {os.path.getsize(lib) / 1e6:.2f} MB")
