#!/usr/bin/env python3
"""Registers, spills and scratch of the kernels in the built library, read from the code object's notes
(no GPU needed).  usage: python tools/kernel_resources.py [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    so = os.environ.get("RDST_HIP_LIB", os.path.join(ROOT, "rdst_amd", "librdst_hip.so"))
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    want = sys.argv[1:]
    for b in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        # (the names are long enough that c++filt gives up on some: strip the namespace, keep the template arguments readable)
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
        short = re.sub(r"EEv.*$", ">", short).replace("ILi", "<").replace("ELi", ",").replace("ELb", ",b").replace("Ij", "<u32,").replace("Im", "<u64,").replace("It", "<u16,").replace("Ih", "<u8,").replace("Li", "")
        if want and not any(w in short for w in want):
            continue
        f = {k: int(re.search(rf"\.{k}:\s+(\d+)", b).group(1)) for k in
             ("vgpr_count", "vgpr_spill_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size")}
        print(f"{short[:100]:100s} vgpr {f['vgpr_count']:3d} spill {f['vgpr_spill_count']:3d} sgpr {f['sgpr_count']:3d} "
              f"scratch {f['private_segment_fixed_size']:4d} static_lds {f['group_segment_fixed_size']}")


if __name__ == "__main__":
    main()
