#!/usr/bin/env python3
"""VGPR / spill / SGPR / LDS / scratch of every kernel in the gfx950 code objects of prosper_amd/csrc/*.o (or the object
files given): the AMDGPU metadata note, through llvm-objcopy + clang-offload-bundler + llvm-readelf.

    python scripts/kernel_resources.py [-k substring] [file.o ...]
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
pattern = ""
if args[:1] == ["-k"]:
    pattern, args = args[1], args[2:]
files = args or sorted(glob.glob(os.path.join(ROOT, "prosper_amd", "csrc", "pt_*.o")))
for path in files:
    with tempfile.TemporaryDirectory() as tmp:
        fat, out = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            continue
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", out], capture_output=True, text=True).stdout
    for k in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        def f(key):
            m = re.search(r"\." + key + r":\s+(\S+)", k)
            return m.group(1) if m else "?"
        name = f("name")
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"\(.*", "", short).replace("void ppt::", "")
        if pattern and pattern not in short:
            continue
        print("%-58s vgpr %3s spill %3s sgpr %3s lds %6s scratch %5s" % (short[:58], f("vgpr_count"), f("vgpr_spill_count"), f("sgpr_count"),
                                                                       f("group_segment_fixed_size"), f("private_segment_fixed_size")))
