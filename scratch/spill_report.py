#!/usr/bin/env python3
"""Register use and spills of every kernel of a built library: scratch/spill_report.py [lib.so]  (reads the code object's metadata)"""
import re, subprocess, sys, tempfile, os
import glob
libs = sys.argv[1:] or sorted(glob.glob("alphagpu_amd/csrc/build/*.o"))
LL = "/opt/rocm/lib/llvm/bin/"
txt = ""
for lib in libs:
  with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "co")
    subprocess.run([LL + "clang-offload-bundler", "--unbundle", "--type=o", f"--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={lib}", f"--output={out}"], check=False, capture_output=True)
    if not os.path.exists(out) or os.path.getsize(out) == 0:
        # the fat binary is a section of the shared object
        sec = os.path.join(d, "fat")
        if subprocess.run([LL + "llvm-objcopy", "--dump-section", f".hip_fatbin={sec}", lib, os.path.join(d, "copy.o")], capture_output=True).returncode != 0:
            continue                                      # (an object without device code)
        subprocess.run([LL + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={sec}", f"--output={out}"], check=True)
    txt += subprocess.run([LL + "llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
rows = []
for blk in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
    g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
    rows.append((g("name"), g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
bad = 0
for n, v, vs, s, ss, l, p in sorted(rows):
    dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void agz::", "")
    flag = " <-- VGPR SPILL" if vs not in ("0", "?") else ""
    bad += bool(flag)
    print(f"{dem:64s} vgpr {v:>4s} spill {vs:>3s}  sgpr {s:>4s} spill {ss:>4s}  scratch {p}{flag}")
print(f"{len(rows)} kernels, {bad} with vector-register spills")
