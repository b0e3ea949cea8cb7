"""Where do the scratch (spill) instructions of a kernel sit: for every scratch_* instruction, the innermost backward-branch loop (by
address range) that contains it.   scratch/where_spills.py <object> <kernel name substring (mangled)>"""
import re, subprocess, sys, tempfile, os
LL = "/opt/rocm/lib/llvm/bin/"
obj, pat = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
    subprocess.run([LL + "llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(d, "copy.o")], check=True)
    subprocess.run([LL + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
    txt = subprocess.run([LL + "llvm-objdump", "-d", co], capture_output=True, text=True).stdout
blocks = re.split(r"\n(?=[0-9a-f]{16} <)", txt)
for b in blocks:
    m = re.match(r"([0-9a-f]{16}) <([^>]+)>:", b)
    if not m or pat not in m.group(2): continue
    base = int(m.group(1), 16)
    ins = []
    for ln in b.splitlines()[1:]:
        mm = re.match(r"\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if mm: ins.append((int(mm.group(2), 16), mm.group(1), ln))
    loops = []
    for a, t, ln in ins:
        if t.startswith(("s_cbranch", "s_branch")):
            tm = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", ln)
            if tm:
                tgt = base + int(tm.group(1), 16)
                if tgt <= a: loops.append((tgt, a))
    print(m.group(2)[:100], "size", ins[-1][0] - base, "bytes;", len(loops), "loops")
    for a, t, ln in ins:
        if t.startswith("scratch_") or t.startswith("v_readlane") and False:
            inner = [l for l in loops if l[0] <= a <= l[1]]
            inner.sort(key=lambda l: l[1] - l[0])
            print(f"  {a - base:#8x}  {t:60s} loops: " + ", ".join(f"[{l[0]-base:#x},{l[1]-base:#x}]({l[1]-l[0]})" for l in inner[:4]))
