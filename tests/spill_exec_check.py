#!/usr/bin/env python3
"""Code-generation hazard check over the built objects: VECTOR-REGISTER SPILL CODE IN FRONT OF AN EXEC RESTORE.
The compiler of this ROCm release can place the spill stores / reloads of a merge block of divergent control flow before the block's
`s_or_b64 exec, exec, s[..]` — they then run with the lanes of the branch that just ended only, and a register that is reused by ALL lanes
behind the restore comes back wrong for the others (found in k_rollout_eager<F_LINE,3,24,3> with the register prefetch: a game whose leaf was
terminal lost its node count; scratch/repro_13.py).  A basic block starts at every branch target; inside a block no scratch_* instruction may
precede an exec restore.   usage: python tests/spill_exec_check.py [objects...]   -> lines 'kernel  address  instruction', exit code 1 if any"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LL = "/opt/rocm/lib/llvm/bin/"

def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat"), os.path.join(d, "co")
        if subprocess.run([LL + "llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(d, "copy.o")], capture_output=True).returncode:
            return ""
        if subprocess.run([LL + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"],
                          capture_output=True).returncode or not os.path.exists(co) or not os.path.getsize(co):
            return ""
        return subprocess.run([LL + "llvm-objdump", "-d", co], capture_output=True, text=True).stdout

def check(txt):
    """-> list of (kernel, address, instruction)"""
    bad = []
    kernels = re.split(r"\n(?=[0-9a-f]{16} <)", txt)
    for k in kernels:
        m = re.match(r"([0-9a-f]{16}) <([^>]+)>:", k)
        if not m:
            continue
        base, name = int(m.group(1), 16), m.group(2)
        ins = []                                    # (address, text)
        targets = set()
        for ln in k.splitlines()[1:]:
            mm = re.match(r"\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
            if not mm:
                continue
            text, addr = mm.group(1), int(mm.group(2), 16)
            ins.append((addr, text))
            if text.startswith(("s_cbranch", "s_branch")):
                t = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", ln)
                if t:
                    targets.add(base + int(t.group(1), 16))
                elif re.search(r"<[^>+]*>\s*$", ln):
                    targets.add(base)
        # the pattern: a block (it starts at a branch target or behind a branch) whose head is nothing but spill code (and scalar
        # instructions of any kind) up to an exec restore.  Spill code deeper inside a block — a reload for the lanes of a short branch-free region, the
        # load / modify / store of a variable that lives in scratch — runs under the mask its lanes need and is not flagged.
        pending, clean = [], True                   # scratch instructions since the block began; nothing else seen since then
        for i, (addr, text) in enumerate(ins):
            if addr in targets or (i and ins[i - 1][1].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc"))):
                pending, clean = [], True
            if text.startswith("scratch_") or ("offen" in text and text.startswith("buffer_")):
                if clean:
                    pending.append((addr, text))
            elif re.match(r"s_or_b64 exec, exec, s\[", text) or re.match(r"s_mov_b64 exec, s\[", text):
                if clean:
                    for a, t in pending:
                        bad.append((name, a, t))
                pending, clean = [], False
            elif not text.startswith("s_"):
                # any SCALAR instruction (an s_add for a scratch offset, an s_load, a lane read into an SGPR ...) may sit between the head of
                # the block and the exec restore without making the spill code legitimate: only vector / memory work that runs under the
                # narrower mask on purpose ends the window
                clean = False
    return bad

if __name__ == "__main__":
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "alphagpu_amd", "csrc", "build", "*.o")))
    n = 0
    for o in objs:
        for name, a, t in check(disassemble(o)):
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print(f"{os.path.basename(o)}  {re.sub(r'[(].*', '', dem)}  {a:#x}  {t}")
            n += 1
    print(f"{n} spill instruction(s) in front of an exec restore")
    sys.exit(1 if n else 0)
