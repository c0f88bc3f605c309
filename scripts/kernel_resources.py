#!/usr/bin/env python3
"""Per-kernel register / spill / scratch / LDS figures of a built object or shared library, read from
the code-object metadata of the embedded gfx950 image:  scripts/kernel_resources.py path/to/file.o
(also imported by tests/test_host_logic.py, which asserts that the hot kernels do not spill)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(path):
    with tempfile.TemporaryDirectory() as t:
        fat, co = os.path.join(t, "fat.bin"), os.path.join(t, "k.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
        lst = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={fat}"],
                             check=True, capture_output=True, text=True).stdout.split()
        target = [x for x in lst if "gfx950" in x][0]
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        f"--targets={target}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        def g(key):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else None
        name = g("name")
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        out[demangled] = {"vgpr": int(g("vgpr_count")), "agpr": int(blk.split()[0]), "sgpr": int(g("sgpr_count")),
                          "spill": int(g("vgpr_spill_count")), "scratch": int(g("private_segment_fixed_size")),
                          "lds": int(g("group_segment_fixed_size"))}
    return out


if __name__ == "__main__":
    for name, r in kernel_resources(sys.argv[1]).items():
        print(f"{name[:100]:100s} " + " ".join(f"{k}={v}" for k, v in r.items()))
