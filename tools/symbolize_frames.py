#!/usr/bin/env python3
"""Place the anonymous frames of a glog-style crash trace ("@ 0x7f...  (unknown)") inside a stripped shared library.

A stripped library gives no names, but its code still tells which addresses can be return addresses (the instruction after
a call) and the load base is page aligned.  Given the library and the frame addresses that are thought to lie in it, every
base is tried that puts the first frame on a return address; a base is a solution when it does so for all frames.  With
three or more frames the solution is normally unique.  --pc ADDRESS additionally prints the instruction a faulting program
counter (not a return address) falls on for each solution, and drops solutions where it is not an instruction start.

    tools/symbolize_frames.py /opt/rocm/lib/libhsa-runtime64.so.1 7f9024c72266 7f9024c635c0
    tools/symbolize_frames.py /opt/rocm/lib/librocprofiler-sdk.so.1 7f902fd47ec0 --pc 7f902f9e72fb
"""
import argparse
import bisect
import re
import subprocess

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def scan(lib):
    p = subprocess.Popen([OBJDUMP, "-d", "--no-show-raw-insn", lib], stdout=subprocess.PIPE, text=True)
    rets, insn, funcs, prev = set(), {}, [], False
    for line in p.stdout:
        m = re.match(r"^\s*([0-9a-f]+):\s+(.*)$", line)
        if m:
            a = int(m.group(1), 16)
            insn[a] = m.group(2)
            if prev:
                rets.add(a)
            prev = m.group(2).startswith("call")
            continue
        m = re.match(r"^([0-9a-f]+) <(.+)>:", line)
        if m:
            funcs.append((int(m.group(1), 16), m.group(2)))
        prev = False
    return rets, insn, funcs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("frames", nargs="+", help="return addresses, hex, innermost first")
    ap.add_argument("--pc", help="faulting program counter, hex")
    a = ap.parse_args()
    frames = [int(x, 16) for x in a.frames]
    pc = int(a.pc, 16) if a.pc else None
    rets, insn, funcs = scan(a.lib)
    starts = [f[0] for f in funcs]

    def where(off):
        i = bisect.bisect_right(starts, off) - 1
        return "%s+0x%x" % (funcs[i][1][:120], off - funcs[i][0]) if i >= 0 else "?"

    n = 0
    for r in sorted(rets):
        if (r & 0xfff) != (frames[0] & 0xfff):
            continue
        base = frames[0] - r
        if not all((f - base) in rets for f in frames[1:]):
            continue
        if pc is not None and (pc - base) not in insn:
            continue
        n += 1
        print("base 0x%x" % base)
        for f in frames:
            print("  0x%x = +0x%x  nearest exported symbol: %s" % (f, f - base, where(f - base)))
        if pc is not None:
            print("  pc 0x%x = +0x%x  %s   [%s]" % (pc, pc - base, insn[pc - base], where(pc - base)))
    print("%d solution(s)" % n)


if __name__ == "__main__":
    main()
