#!/usr/bin/env python3
"""Static per-phase instruction counts of one kernel of a gfx950 code object built with -g.

usage: isa_phases.py <code object> <kernel symbol substring> [--lines]
Every instruction is attributed to the source line of rollout_reg_body (agz_tree_reg.hpp) at the bottom of its inline stack
(llvm-symbolizer -i), then to the phase whose line range contains it (PHASES below).  Prints VALU / SALU / LDS / VMEM / MFMA counts
per phase = the static code of the phase; scratch/valu_table.py multiplies them by the dynamic trip counts.
"""
import collections
import json
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin/"


def kernel_range(co, name):
    out = subprocess.check_output([LLVM + "llvm-objdump", "-t", co], text=True)
    for ln in out.splitlines():
        if name in ln and " F .text" in ln:
            f = ln.split()
            return int(f[0], 16), int(f[4], 16), f[-1]
    raise SystemExit("kernel not found")


def classify(mn):
    if mn.startswith("v_mfma"):
        return "mfma"
    if mn.startswith("v_"):
        return "valu"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    co, name = sys.argv[1], sys.argv[2]
    start, size, sym = kernel_range(co, name)
    dis = subprocess.check_output([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", f"--start-address={start}",
                                   f"--stop-address={start + size}", co], text=True)
    ins = []
    for ln in dis.splitlines():
        m = re.match(r"\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):", ln)
        if m:
            ins.append((int(m.group(2), 16), m.group(1), ln.strip()))
    addrs = "\n".join(hex(a) for a, _, _ in ins)
    sy = subprocess.run([LLVM + "llvm-symbolizer", "--obj=" + co, "-i", "-a", "--output-style=JSON"], input=addrs, text=True,
                        capture_output=True).stdout
    res = []
    for ln in sy.splitlines():
        ln = ln.strip()
        if ln.startswith("{"):
            res.append(json.loads(ln))
    assert len(res) == len(ins), (len(res), len(ins))
    out = []
    for (a, mn, txt), r in zip(ins, res):
        frames = [(f["FunctionName"], f["FileName"].split("/")[-1], f["Line"]) for f in r.get("Symbol", [])]
        out.append(dict(addr=a, mn=mn, cls=classify(mn), frames=frames, txt=txt))
    json.dump(dict(symbol=sym, ins=out), open(sys.argv[3] if len(sys.argv) > 3 else "/tmp/isa.json", "w"))
    c = collections.Counter(x["cls"] for x in out)
    print(sym, dict(c), "total", len(out))


if __name__ == "__main__":
    main()
