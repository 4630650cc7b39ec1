#!/usr/bin/env python3
"""tools/profc/report.py [--counters gpurun_out/profc/counters_4096.npz] [--asm /tmp/isa/k1.s] [--kernel cn_env_kernel_fair_s360_w4]

The DYNAMIC instruction ledger of the step kernel: every instruction of the PRODUCT's code (assembly with line tables: the same
instructions as lib/libcrowdnav.so, `tools/isa_ledger.py --rebuild --flags="-DCN_TU=1 -fdebug-info-for-profiling" --asm /tmp/isa/k1.s` writes it -- the second flag puts the
unroller's duplication factors into the line table's discriminators, without it unrolled loops are counted several times) weighted with how often a
wavefront executed the source region it belongs to (clang region counters of the counted build, tools/profc/run.py).

  instruction -> its inlining chain of (file, line, column) from the .loc comments
              -> per frame: the innermost counted source region containing that position (coverage mapping, tools/profc/covmap.py)
              -> executions = the MINIMUM over the chain's frames (a helper's own counters sum over all of its call sites; the
                 caller's region bounds this site; a cold branch inside the helper bounds its own instructions)
Per env-step = divided by (envs x launches) of the counted run.  Printed: totals per instruction class (compare with the PMC counts of
profiles/rNN/counters.json), the stage x class table, the heaviest source lines, and the lane occupancy of the heaviest regions."""
import argparse
import bisect
import collections
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_ledger as IL


def evaluate(cov, waves, lanes):
    """counts of every region of one function instance -> list of (file idx, kind, ls, cs, le, ce, waves, lanes)"""
    ex = cov["exprs"]
    memo = {}

    def val(c, depth=0):
        t, i = c
        if t == "z":
            return (0, 0)
        if t == "c":
            return (int(waves[i]), int(lanes[i])) if i < len(waves) else (0, 0)
        key = (t, i)
        if key in memo:
            return memo[key]
        if depth > 200 or i >= len(ex):
            return (0, 0)
        a, b = val(ex[i][0], depth + 1), val(ex[i][1], depth + 1)
        r = (a[0] - b[0], a[1] - b[1]) if t == "-" else (a[0] + b[0], a[1] + b[1])
        memo[key] = r
        return r
    out = []
    for r in cov["regions"]:
        if r["kind"] != "code":
            continue
        w, l = val(r["c"])
        out.append((cov["files"][r["file"]], r["ls"], r["cs"], r["le"], r["ce"], max(w, 0), max(l, 0), r["file"]))
    return out


def dup_factor(d):
    def nxt(v):
        return (v >> (14 if (v & 0x40) else 7)) if (v & 1) == 0 else (v >> 1)

    def comp(v):
        if v & 1:
            return 0
        v >>= 1
        return (((v >> 7) << 5) | (v & 0x1f)) if (v & 0x20) else (v & 0x1f)
    df = comp(nxt(d))
    return df if df > 0 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--counters", default=os.path.join(ROOT, "gpurun_out", "profc", "counters_4096.npz"))
    ap.add_argument("--symbols", default=os.path.join(ROOT, "tools", "profc", "symbols.json"))
    ap.add_argument("--asm", default="/tmp/isa/k1d.s")
    ap.add_argument("--kernel", default="cn_env_kernel_fair_s360_w4")
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--lines", default="", help="print every instruction of these source lines (comma separated, kernel file) with its count")
    a = ap.parse_args()
    sym = json.load(open(a.symbols))
    z = np.load(a.counters)
    c = z["counters"].astype(np.uint64)
    per = float(int(z["envs"]) * int(z["steps"]))
    W = (c >> np.uint64(32)).astype(np.int64)
    Ln = (c & np.uint64(0xffffffff)).astype(np.int64)
    base = [os.path.basename(f) for f in sym["filenames"]]
    # function instances -> aggregated by (file, body span)
    funcs = {}      # (fname, ls, cs, le, ce) -> {region coords -> [waves, lanes]}
    for f in sym["functions"]:
        if not f["cov"]:
            continue
        cov = f["cov"][0]
        regs = evaluate(cov, W[f["cnt_off"]:f["cnt_off"] + f["ncnt"]], Ln[f["cnt_off"]:f["cnt_off"] + f["ncnt"]])
        body = [r for r in regs if r[7] == 0]
        if not body:
            continue
        b0 = body[0]
        key = (base[b0[0]], b0[1], b0[2], b0[3], b0[4])
        d = funcs.setdefault(key, {})
        for (fi, ls, cs, le, ce, w, l, lf) in regs:
            k = (base[fi], ls, cs, le, ce)
            v = d.setdefault(k, [0, 0])
            v[0] += w; v[1] += l
    by_file = collections.defaultdict(list)
    for key in funcs:
        by_file[key[0]].append(key)

    def contains(r, line, col):
        _, ls, cs, le, ce = r
        return (ls, cs) <= (line, col) and (line, col) < (le, ce) if (le, ce) > (ls, cs) else False
    cache = {}

    def lookup(fname, line, col):
        k = (fname, line, col)
        if k in cache:
            return cache[k]
        best = None
        for fk in by_file.get(fname, ()):
            if contains(fk, line, col):
                if best is None or (fk[3] - fk[1], fk[4]) < (best[3] - best[1], best[4]) or (fk[3] - fk[1] == best[3] - best[1] and (fk[1], fk[2]) > (best[1], best[2])):
                    best = fk
        res = None
        if best is not None:
            inner = None
            for rk, v in funcs[best].items():
                if rk[0] == fname and contains(rk, line, col):
                    if inner is None or (rk[1], rk[2]) > (inner[0][1], inner[0][2]):
                        inner = (rk, v)
            if inner is not None:
                ent = funcs[best].get(best, [0, 0])[0]          # executions of the function's body region = calls of the function
                res = (inner[1][0], inner[1][1], inner[0], ent)
        cache[k] = res
        return res

    # ---- the product's instructions -------------------------------------------------------------------------------------------
    table = IL.stage_table(IL.SRC)
    helpers = IL.helper_ranges(IL.SRC)
    starts = [s for s, _ in table]

    def stage_of_line(ln):
        k = bisect.bisect_right(starts, ln) - 1
        return table[k][1] if k >= 0 else None
    loc_rx = re.compile(r"([\w./+-]+):(\d+):(\d+)")
    inside = False
    cur_chain, cur_stage, cur_df = [], "kernel.entry", 1
    tot = collections.Counter()
    st_cls = collections.defaultdict(collections.Counter)
    line_cls = collections.defaultdict(collections.Counter)
    line_exec = {}
    static = collections.Counter()
    unknown = 0
    lost = 0
    block = 0
    recs = []
    want = set()
    for x in a.lines.split(","):            # "1183,247-345": single lines and ranges
        if "-" in x:
            lo_, hi_ = x.split("-"); want |= set(range(int(lo_), int(hi_) + 1))
        elif x:
            want.add(int(x))
    detail = []
    entry = lookup("crowdnav_kernel.hip", 1, 1)
    for line in open(a.asm):
        if not inside:
            if line.startswith(a.kernel + ":"):
                inside = True
            continue
        if line.startswith(".Lfunc_end"):
            break
        s = line.strip()
        if s.startswith(".loc"):
            cm = s.split(";", 1)[1] if ";" in s else ""
            # -fdebug-info-for-profiling: an instruction copied by the loop unroller carries the number of copies in its discriminator
            # (llvm DILocation::encodeDiscriminator: base discriminator | duplication factor | copy id, prefix-coded); a copy runs
            # 1 / factor of the region's executions
            dm = re.search(r"discriminator (\d+)", s.split(";", 1)[0])
            cur_df = dup_factor(int(dm.group(1))) if dm else 1
            cur_chain = [(os.path.basename(p_), int(l_), int(c_)) for p_, l_, c_ in loc_rx.findall(cm)]
            for (fn, ln, _c) in cur_chain:
                if fn != "crowdnav_kernel.hip" or ln == 0 or any(lo <= ln <= hi for lo, hi in helpers):
                    continue
                st = stage_of_line(ln)
                if st:
                    cur_stage = st
                    break
            continue
        if s.endswith(":") and not s.startswith(";"):
            block += 1                                     # a label: a new basic block
            continue
        if not s or s.startswith((".", ";")):
            continue
        parts = s.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        cls = "other" if op == "s_nop" else IL.classify(op, args)      # (s_nop: its own column, `other`)
        # executions of this instruction: outermost frame first -- E(kernel frame) = its region's count; one level further in,
        # E = (count of the callee's region at this position) x E(call site) / (calls of the callee over ALL its call sites): a
        # callee's counters sum over its call sites, the caller's region says how many of those calls came from here
        cnt = None
        hit = None
        for (fn, ln, col) in reversed(cur_chain):
            r = lookup(fn, ln, col)
            if r is None:
                continue                                   # a frame in a header without counters: transparent
            if cnt is None:
                cnt = float(r[0])
            else:
                cnt = (r[0] * cnt / r[3]) if r[3] > 0 else 0.0
            hit = (cnt, (r[1] * cnt / r[0]) if r[0] > 0 else 0.0)
        if cnt is not None:
            cnt /= cur_df
        # the line the instruction is booked to: the first frame inside the kernel file that is not a shared helper
        lk = None
        for (fn, ln, col) in cur_chain:
            if fn == "crowdnav_kernel.hip" and ln > 0 and not any(lo <= ln <= hi for lo, hi in helpers):
                lk = ln
                break
        if lk is None and cur_chain and cur_chain[0][1] > 0:
            lk = -cur_chain[0][1]
        recs.append([block, cls, cnt, cur_stage, lk, hit, s.split(";")[0].strip()])
        if cls == "br" and op != "s_endpgm":
            block += 1                                     # ... and a branch ends one
    # compiler-generated instructions carry line 0 (no frame to look up): they run as often as their basic block -- the count of the
    # nearest instruction of the same block that has one (the previous one first)
    for i, r in enumerate(recs):
        if r[2] is not None:
            continue
        unknown += 1
        for j in list(range(i - 1, -1, -1)) + list(range(i + 1, len(recs))):
            if recs[j][0] != r[0]:
                if j < i:
                    continue_back = False
                    # left the block going backwards: look forwards instead
                    for k in range(i + 1, len(recs)):
                        if recs[k][0] != r[0]:
                            break
                        if recs[k][2] is not None and recs[k][5] != "filled":
                            r[2] = recs[k][2]; break
                    break
                break
            if recs[j][2] is not None and recs[j][5] != "filled":
                r[2] = recs[j][2]; break
        if r[2] is None:
            r[2] = 0.0; lost += 1
        else:
            r[5] = "filled"
            if r[4] is None:
                # book it to the line of the neighbour it took its count from
                nb = [q for q in recs[max(0, i - 8):i + 8] if q[0] == r[0] and q[4] is not None]
                r[4] = nb[0][4] if nb else None
    for (blk, cls, cnt, stg, lk, hit, txt) in recs:
        x = cnt / per
        tot[cls] += x
        static[cls] += 1
        st_cls[stg][cls] += x
        line_cls[lk][cls] += x
        if hit is not None and hit != "filled" and lk is not None:
            e = line_exec.setdefault(lk, [0.0, 0.0])
            e[0] = max(e[0], hit[0] / per); e[1] = max(e[1], hit[1] / per)
        if lk in want:
            detail.append((lk, x, txt))
    CL = IL.CLASSES
    valu = ["f64", "f32", "int", "cvt", "cmp", "mov", "sel", "lane", "dpp"]
    print("# dynamic instruction ledger of %s: wave instructions per env-step (%d envs x %d launches counted; %d compiler-generated instructions took their basic block's count, %d found none)" % (
        a.kernel, int(z["envs"]), int(z["steps"]), unknown, lost))
    print("%-16s" % "stage" + "".join("%7s" % c_ for c_ in CL) + "%8s%8s" % ("total", "VALU"))
    order = [n for _, n in table if n in st_cls]
    seen = set()
    order = [n for n in order if not (n in seen or seen.add(n))]
    for st in order:
        row = st_cls[st]
        print("%-16s" % st + "".join("%7.0f" % row[c_] for c_ in CL) + "%8.0f%8.0f" % (sum(row.values()), sum(row[c_] for c_ in valu)))
    print("%-16s" % "TOTAL" + "".join("%7.0f" % tot[c_] for c_ in CL) + "%8.0f%8.0f" % (sum(tot.values()), sum(tot[c_] for c_ in valu)))
    print("%-16s" % "static" + "".join("%7d" % static[c_] for c_ in CL) + "%8d%8d" % (sum(static.values()), sum(static[c_] for c_ in valu)))
    print("\n# heaviest source lines (crowdnav_kernel.hip; negative = a header line): wave instructions per env-step, executions of the line's "
          "hottest region per env-step, mean active lanes")
    src = open(IL.SRC).read().split("\n")
    rows = sorted(line_cls.items(), key=lambda kv: -sum(kv[1].values()))
    for lk, row in rows[:a.top]:
        t = sum(row.values())
        e = line_exec.get(lk, [0, 0])
        txt = src[lk - 1].strip()[:110] if lk and lk > 0 and lk <= len(src) else ""
        print("%6s %7.1f  valu %6.1f salu %5.1f br %5.1f lds %4.1f | x%6.2f lanes %4.1f | %s" % (
            lk, t, sum(row[c_] for c_ in valu), row["salu"], row["br"], row["lds"], e[0], (e[1] / e[0]) if e[0] else 0.0, txt))
    if detail:
        print("\n# instructions of the requested lines")
        for lk, x, s in detail:
            print("%6d x%7.3f  %s" % (lk, x, s))


if __name__ == "__main__":
    main()
