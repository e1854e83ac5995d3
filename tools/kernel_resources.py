#!/usr/bin/env python3
"""Register, spill, scratch and LDS use of every kernel in librt_hip.so, from the code object's own metadata.

usage: tools/kernel_resources.py [path/to/librt_hip.so]   (runs anywhere: no GPU needed)
rocprofv3's rocpd tables report `arch_vgpr_count` (64 for a kernel whose ELF metadata says .vgpr_count 128 on this
wave64 target), which misled round 1's profiles; this reads the authoritative numbers: .vgpr_count,
.vgpr_spill_count, .sgpr_count, .sgpr_spill_count, .private_segment_fixed_size (scratch bytes per lane),
.group_segment_fixed_size (LDS bytes per block).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(so_path: str):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", so_path, os.path.join(td, "x")])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    keys = ("vgpr_count", "vgpr_spill_count", "agpr_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size",
            "group_segment_fixed_size", "max_flat_workgroup_size")
    out = []
    for block in re.split(r"\n  - (?=\.agpr_count:)", notes)[1:]:  # one list item of amdhsa.kernels per kernel
        k = {}
        m = re.search(r"^\s+\.name:\s+(\S+)\s*$", block, re.M)
        for mm in re.finditer(r"^    \.name:\s+(\S+)\s*$", block, re.M):
            k["name"] = mm.group(1)  # the kernel's own .name is the one at item indent (arguments are deeper)
        for key in keys:
            mm = re.search(r"^\s*\.%s:\s+(\d+)\s*$" % key, block, re.M)
            if mm:
                k[key] = int(mm.group(1))
        out.append(k)
    return out


def demangle(name: str) -> str:
    try:
        return subprocess.check_output(["c++filt", name], text=True).strip()
    except Exception:
        return name


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                            "raytracer_amd", "csrc", "librt_hip.so")
    rows = kernel_resources(so)
    print(f"== ELF metadata of {os.path.basename(so)} (gfx950 code object)")
    print(f"{'kernel':<64} {'vgpr':>5} {'vspill':>6} {'agpr':>5} {'sgpr':>5} {'sspill':>6} {'scratch/lane':>12} {'lds/block':>9}")
    for k in rows:
        name = demangle(k.get("name", "?"))
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = name if len(name) <= 64 else name[:61] + "..."
        print(f"{name:<64} {k.get('vgpr_count', 0):>5} {k.get('vgpr_spill_count', 0):>6} {k.get('agpr_count', 0):>5} "
              f"{k.get('sgpr_count', 0):>5} {k.get('sgpr_spill_count', 0):>6} {k.get('private_segment_fixed_size', 0):>12} "
              f"{k.get('group_segment_fixed_size', 0):>9}")


if __name__ == "__main__":
    main()
