#!/usr/bin/env python3
"""tools/profc/rewrite_ir.py in.ll out.ll -- every update of a clang region counter (`atomicrmw add <ptr into @__profc_*>, i64 V
monotonic`, what -fprofile-instr-generate -fprofile-update=atomic leaves in the optimised IR) becomes
`call void @cn_prof_hit(ptr <the same address>, i64 V)` (tools/profc/profc_block.h: +1 per wavefront in the counter's high
word, + active lanes in its low word).  Updates inside cn_prof_hit / cn_profc_kernel themselves are left alone."""
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
rx = re.compile(r"^(\s*)(?:%[\w.]+\s*=\s*)?atomicrmw\s+(?:volatile\s+)?add\s+(ptr(?:\s+addrspace\(\d+\))?)\s+(.*@\"?__profc_.*?),\s*i64\s+([^,\s]+)\s+(?:syncscope\(\"[^\"]*\"\)\s+)?monotonic.*$")
n = 0
skip = False
with open(src) as f, open(dst, "w") as o:
    for line in f:
        if line.startswith("define "):
            skip = ("@cn_prof_hit(" in line) or ("@cn_profc_kernel(" in line)
        m = None if skip else rx.match(line)
        if m and "__profc_" in line:
            ind, pty, pexp, val = m.groups()
            arg = ("ptr %s" % pexp) if pty == "ptr" else ("ptr addrspacecast (%s %s to ptr)" % (pty, pexp))
            dbg = re.search(r"(, !dbg !\d+)", line)           # a call between two functions with debug info must keep its location
            o.write("%scall void @cn_prof_hit(%s, i64 %s)%s\n" % (ind, arg, val, dbg.group(1) if dbg else ""))
            n += 1
        else:
            o.write(line)
left = sum(1 for l in open(dst) if "atomicrmw" in l and "__profc_" in l)
print("rewrite_ir: %d counter updates rewritten, %d atomicrmw on __profc_ left (cn_prof_hit / cn_profc_kernel own)" % (n, left))
