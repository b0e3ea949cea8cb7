#!/usr/bin/env python3
"""Instruction mix of the largest loops of one kernel of a built object: scratch/isa_loopmix.py <object> <mangled kernel name prefix> [dump.s]
(lane = v_readlane / v_writelane: scalar-register spill traffic)"""
import re, subprocess, collections, os, sys, tempfile
LL = "/opt/rocm/lib/llvm/bin/"
obj, pat = sys.argv[1], sys.argv[2]
d = tempfile.mkdtemp()
fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
subprocess.run([LL + "llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(d, "copy.o")], check=True)
subprocess.run([LL + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
txt = subprocess.run([LL + "llvm-objdump", "-d", co], capture_output=True, text=True).stdout
for b in re.split(r"\n(?=[0-9a-f]{16} <)", txt):
    m = re.match(r"([0-9a-f]{16}) <([^>]+)>:", b)
    if not m or pat not in m.group(2):
        continue
    base = int(m.group(1), 16)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(b)
    ins = []
    for ln in b.splitlines()[1:]:
        mm = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)", ln)
        if mm:
            ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2), mm.group(4)))
    loops = []
    for a, t, ops, tail in ins:
        if t.startswith(("s_cbranch", "s_branch")):
            tm = re.search(r"\+0x([0-9a-fA-F]+)>", tail)
            if tm and base + int(tm.group(1), 16) <= a:
                loops.append((base + int(tm.group(1), 16), a))

    def kind(t):
        return ('mfma' if t.startswith('v_mfma') else 'lane' if t in ('v_readlane_b32', 'v_writelane_b32') else 'valu' if t.startswith('v_') else
                'salu' if t.startswith('s_') else 'lds' if t.startswith('ds_') else 'vmem' if t.startswith(('global', 'buffer', 'scratch', 'flat')) else 'o')
    print(m.group(2)[:90], "bytes", ins[-1][0] - base, "loops", len(loops))
    for lo, hi in sorted(set(loops), key=lambda x: x[1] - x[0], reverse=True)[:24]:
        c = collections.Counter(kind(t) for a, t, o, _ in ins if lo <= a <= hi)
        ops = collections.Counter(t for a, t, o, _ in ins if lo <= a <= hi and t.startswith('v_'))
        print(f"  [{lo - base:#x},{hi - base:#x}] {hi - lo:6d} B  {dict(c)}  top: {ops.most_common(8)}")
