#!/usr/bin/env python3
"""ISA ledger of one kernel: every instruction of the compiled gfx950 code, bucketed by pipeline stage x instruction class.

    tools/isa_ledger.py [--kernel cn_env_kernel_fair] [--asm /tmp/kern.s] [--detail STAGE]

Compiles crowdnav_kernel.hip to assembly with line tables (-gline-tables-only; code generation is unchanged) unless --asm
names an existing file, then walks the kernel's instructions.  Each `.loc` comment carries the whole inlining chain
(`file:line @[ caller:line @[ ... ] ]`), so an instruction inside cn_round_scaled() inlined into the ray loop is booked
to the RAY LOOP, not to crowdnav_device.h.  The stage of an instruction = the stage of the first chain element (innermost
outwards) that lies inside observe() / env_kernel_body(), looked up in the line table below (kept next to the CN_T stamps
of the source: the table is re-derived from the source's own `CN_T(k)` markers, so it follows edits).

The counts are STATIC (one per instruction in the binary); loops execute their bodies several times.  Dynamic totals per
env-step come from the PMC counters (profiles/rNN/pmc_extra.txt); this table says where the non-arithmetic instructions
are generated and which stages carry the spill traffic (v_readlane / v_writelane of SGPR spills, scratch).
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd", "csrc", "crowdnav_kernel.hip")

CLASSES = ["f64", "f32", "int", "cvt", "cmp", "mov", "sel", "lane", "dpp", "salu", "smem", "lds", "vmem", "br", "wait", "other"]


def classify(op, args):
    if op.startswith("s_"):
        if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64", "s_endpgm"):
            return "br"
        if op.startswith("s_waitcnt") or op in ("s_nop", "s_sleep", "s_barrier", "s_setprio", "s_memtime", "s_getreg_b32"):
            return "wait"
        if op.startswith("s_load") or op.startswith("s_buffer") or op.startswith("s_store") or op.startswith("s_dcache"):
            return "smem"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if "dpp" in op or "dpp" in args or op.startswith(("v_permlane", "ds_bpermute", "ds_permute", "v_mov_b32_dpp")):
        return "dpp"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "mov"
    if op.startswith(("v_cndmask",)):
        return "sel"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "cmp"
    if op.startswith("v_cvt"):
        return "cvt"
    if re.search(r"_f64", op):
        return "f64"
    if re.search(r"_f32|_f16", op):
        return "f32"
    if op.startswith("v_"):
        return "int"
    return "other"


def stage_table(src):
    """[(first_line, name)] in ascending order, from the source's own structure."""
    lines = open(src).read().split("\n")

    def find(pat, start=0):
        rx = re.compile(pat)
        for i in range(start, len(lines)):
            if rx.search(lines[i]):
                return i + 1
        raise SystemExit("isa_ledger: pattern %r not found in %s" % (pat, src))

    obs = find(r"^__device__ __forceinline__ void observe\(KP p")
    t = []
    t.append((find(r"^__device__ __forceinline__ void ped_advance"), "sim.peds"))
    t.append((find(r"^__device__ __forceinline__ double lane_d"), "sim.robot"))
    t.append((find(r"^__device__ __forceinline__ void sim_advance_contact"), "sim.ticks"))
    t.append((find(r"^__device__ __forceinline__ int near_peds"), "near_peds"))
    t.append((find(r"^__device__ __forceinline__ double cast_ray"), "ray.cast"))
    t.append((find(r"^__device__ __forceinline__ double orig_heading"), "layout1"))
    t.append((find(r"^__device__ __forceinline__ double bbox_size"), "bbox"))
    t.append((find(r"^__device__ __forceinline__ void tracker_stage"), "tracker"))
    t.append((obs, "obs.head"))
    t.append((find(r"CN_T\(2\);", obs), "obs.lidar_setup"))
    t.append((find(r"CN_T\(3\);", obs), "ray.loop"))
    t.append((find(r"CN_T\(4\);", obs), "obs.bbox_deque"))
    t.append((find(r"CN_T\(5\);", obs), "gradients"))
    t.append((find(r"CN_T\(6\);", obs), "flags"))
    t.append((find(r"CN_T\(7\);", obs), "type_machine"))
    t.append((find(r"CN_T\(9\);", obs), "association"))
    t.append((find(r"CN_T\(10\);", obs), "order_split"))
    t.append((find(r"CN_T\(11\);", obs), "prefix"))
    t.append((find(r"CN_T\(12\);", obs), "confirmation"))
    t.append((find(r"ENV:637-654", obs), "counters"))
    t.append((find(r"risk_mode gt \(SURVEY", obs), "gt_entries"))
    t.append((find(r"CN_T\(14\);", obs), "speeds"))
    t.append((find(r"CN_T\(15\);", obs), "cone_topk"))
    t.append((find(r"CN_T\(16\);", obs), "tail"))
    t.append((find(r"CN_T\(17\);", obs), "trk_writeback"))
    t.append((find(r"^struct RwLds"), "layout2"))
    t.append((find(r"^__device__ __forceinline__ double compute_reward\(KP"), "reward"))
    body = find(r"^__device__ __forceinline__ void env_kernel_body")
    t.append((body, "body.setup"))
    t.append((find(r"---- load env state", body), "body.load"))
    t.append((find(r"CN_T\(1\);", body), "body.step"))
    t.append((find(r"CN_T\(18\);", body), "body.store"))
    t.append((find(r"^#ifdef CN_TIMING\n?", find(r"CN_T\(19\);", body)), "kernel.entry"))
    t.sort()
    return t


HELPERS = [  # functions of crowdnav_kernel.hip that are called from several stages: skipped when looking for the stage
    r"heading_to_goal", r"dist3", r"in_box", r"ring_segment", r"bcast_d", r"uni64", r"waypoint_refresh"]


def helper_ranges(src):
    lines = open(src).read().split("\n")
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^__device__ __forceinline__ .*?\b(\w+)\(", l)
        if m and m.group(1) in HELPERS:
            j = i
            while j < len(lines) and lines[j] != "}":
                j += 1
            out.append((i + 1, j + 1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="cn_env_kernel_fair")
    ap.add_argument("--asm", default="/tmp/isa/kern.s")
    ap.add_argument("--rebuild", action="store_true")
    ap.add_argument("--detail", default=None, help="print the mnemonic histogram of one stage")
    ap.add_argument("--flags", default="")
    a = ap.parse_args()
    if a.rebuild or not os.path.exists(a.asm):
        os.makedirs(os.path.dirname(a.asm), exist_ok=True)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-builtin-pow",
               "-gline-tables-only", "--cuda-device-only", "-S", "-o", a.asm, SRC] + a.flags.split()
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    table = stage_table(SRC)
    helpers = helper_ranges(SRC)
    starts = [s for s, _ in table]

    def stage_of_line(ln):
        import bisect
        k = bisect.bisect_right(starts, ln) - 1
        return table[k][1] if k >= 0 else None

    def is_helper(ln):
        return any(lo <= ln <= hi for lo, hi in helpers)

    counts = collections.defaultdict(lambda: collections.Counter())
    detail = collections.Counter()
    cur = "kernel.entry"
    inside = False
    loc_rx = re.compile(r"crowdnav_kernel\.hip:(\d+):\d+")
    n_instr = 0
    spill_notes = collections.Counter()
    for line in open(a.asm):
        if not inside:
            if line.startswith(a.kernel + ":"):
                inside = True
            continue
        if line.startswith(".Lfunc_end"):
            break
        s = line.strip()
        if s.startswith(".loc"):
            chain = [int(x) for x in loc_rx.findall(s.split(";", 1)[1] if ";" in s else "")]
            st = None
            for ln in chain:               # innermost first
                if ln == 0 or is_helper(ln):
                    continue
                st = stage_of_line(ln)
                if st:
                    break
            if st:
                cur = st
            continue
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        args = parts[1] if len(parts) > 1 else ""
        c = classify(op, args)
        counts[cur][c] += 1
        n_instr += 1
        if "Spill" in args or "Reload" in args or "spill" in args:
            spill_notes[cur] += 1
        if a.detail and cur == a.detail:
            detail[op] += 1
    if not inside:
        raise SystemExit("kernel %s not found in %s" % (a.kernel, a.asm))
    order = [n for _, n in table if n in counts]
    seen = set()
    order = [n for n in order if not (n in seen or seen.add(n))]
    print("# static ISA ledger of %s (%d instructions)" % (a.kernel, n_instr))
    print("%-16s" % "stage" + "".join("%7s" % c for c in CLASSES) + "%8s%8s" % ("total", "spill"))
    tot = collections.Counter()
    for st in order:
        row = counts[st]
        tot.update(row)
        print("%-16s" % st + "".join("%7d" % row[c] for c in CLASSES) + "%8d%8d" % (sum(row.values()), spill_notes[st]))
    print("%-16s" % "TOTAL" + "".join("%7d" % tot[c] for c in CLASSES) + "%8d%8d" % (sum(tot.values()), sum(spill_notes.values())))
    if a.detail:
        print("\n# mnemonics of stage %s" % a.detail)
        for op, k in detail.most_common(60):
            print("  %-28s %5d" % (op, k))


if __name__ == "__main__":
    main()
