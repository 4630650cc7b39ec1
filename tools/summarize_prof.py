#!/usr/bin/env python
"""Condense a tools/profile.sh output directory into the few numbers DESIGN.md / bench.py quote."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_hash   # stamps counters.json / traffic.json: bench.py only quotes a profile of THIS kernel


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def main(d):
    for tdir, what in (("trace", "bench.py --steps 300 --warmup 30"), ("trace_driver", "the driver's command: bench.py --steps 20 --warmup 5")):
        st = find(os.path.join(d, tdir), "*kernel_stats.csv")
        if st:
            print("== kernel stats, %s (%s)" % (what, os.path.relpath(st, d)))
            for row in csv.DictReader(open(st)):
                if not row.get("Name", "").startswith("cn_"):
                    continue
                print("  %-28s calls %6s  total %12s ns  avg %12s ns  %6s%%" % (
                    row.get("Name", "")[:28], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"),
                    row.get("Percentage")))
        tr = find(os.path.join(d, tdir), "*kernel_trace.csv")
        if tr:
            legs = {}
            for r in csv.DictReader(open(tr)):
                kn = r["Kernel_Name"].split("(")[0].strip()
                if not kn.startswith("cn_env_kernel"):
                    continue
                g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
                legs.setdefault((kn, g), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            # bench.py legs: cn_env_kernel_same (same-call reset), cn_env_kernel at the full grid (one launch per step),
            # cn_env_kernel at grid / G (the stream-group leg: G launches in flight, so durations overlap)
            print("== env kernel average duration by (kernel, grid = 64 x envs per launch), %s:" % what)
            for (kn, g), dd in sorted(legs.items()):
                print("  %-22s grid %7d (%5d envs)  launches %5d  avg %.1f us  min %.1f  max %.1f" % (
                    kn, g, g // 64, len(dd), sum(dd) / len(dd) / 1e3, min(dd) / 1e3, max(dd) / 1e3))
    sq = {}
    for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        f = find(os.path.join(d, tag), "*counter_collection.csv")
        if not f:
            print("== %s: no counter csv" % tag)
            continue
        allrows = list(csv.DictReader(open(f)))
        # the one-launch-per-step legs (full grid): cn_env_kernel_fair is what cn_step launches for a handle that fills the device
        # (CN_ARB_AUTO), cn_env_kernel the same step under the hardware's oldest-first issue order (bench leg 1_groups_oldest_first)
        for kname in ("cn_env_kernel", "cn_env_kernel_fair"):
            acc, cnt = {}, {}
            rows = [r for r in allrows if r.get("Kernel_Name", "").split("(")[0].strip() == kname]
            gmax = max([int(r.get("Grid_Size", 0) or 0) for r in rows] or [0])
            for row in rows:
                if int(row.get("Grid_Size", 0) or 0) != gmax:
                    continue
                k = row["Counter_Name"]; v = float(row["Counter_Value"])
                acc[k] = acc.get(k, 0.0) + v; cnt[k] = cnt.get(k, 0) + 1
            if not acc:
                continue
            print("== %s (per %s dispatch of %d envs, mean of %s)" % (tag, kname, gmax // 64, sorted(set(cnt.values()))))
            for k in sorted(acc):
                print("  %-24s %.6g" % (k, acc[k] / cnt[k]))
                sq[k] = acc[k] / cnt[k]                # the later (fair) kernel's figures are the ones counters.json keeps
            sq["_envs"] = gmax // 64; sq["_kernel"] = kname
    if "SQ_INSTS_VALU" in sq and "SQ_BUSY_CYCLES" in sq:
        import json
        n = float(sq["_envs"])
        # SQ_ACTIVE_INST_VALU counts quad-cycles per wave (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"); SQ_BUSY_CYCLES is
        # summed over the 32 shader engines (8 XCDs x 4) in cycles, so the launch lasted SQ_BUSY_CYCLES / 32 cycles on 1024 SIMDs
        kcycles = sq["SQ_BUSY_CYCLES"] / 32.0
        out = {"valu_busy": 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (1024.0 * kcycles),
               "wave_instr_per_env_step": {"valu": sq["SQ_INSTS_VALU"] / n, "salu": sq.get("SQ_INSTS_SALU", 0) / n,
                                           "lds": sq.get("SQ_INSTS_LDS", 0) / n},
               "wave_quad_cycles_per_env_step": sq.get("SQ_WAVE_CYCLES", 0) / n,
               "wait_any_frac": sq.get("SQ_WAIT_ANY", 0) / max(1.0, sq.get("SQ_WAVE_CYCLES", 1)),
               "launch_cycles": kcycles, "envs_per_launch": n, "csrc_hash": csrc_hash(), "kernel": sq.get("_kernel"),
               "lds_bank_conflict_frac": (sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]) if sq.get("SQ_LDS_IDX_ACTIVE") else None,
               "source": "rocprofv3 --pmc SQ_* pass of this bench command, full-grid (one launch per step) dispatches: valu_busy = "
                         "4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32)"}
        json.dump(out, open(os.path.join(d, "counters.json"), "w"), indent=1)
        print("== counters.json: VALU busy %.3f, %.0f VALU + %.0f SALU + %.0f LDS wave instructions per env-step" % (
            out["valu_busy"], out["wave_instr_per_env_step"]["valu"], out["wave_instr_per_env_step"]["salu"],
            out["wave_instr_per_env_step"]["lds"]))
    # calibration: known 1 GiB streams at 4 and 8 bytes per lane -> KB reported per byte moved
    calib = {}
    for tag, ctr in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
        f = find(os.path.join(d, tag), "*counter_collection.csv")
        if not f:
            continue
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            if "cn_calib" not in kn or row["Counter_Name"] != ctr:
                continue
            kind = ("read" if "read" in kn else "write") + ("4" if "float" in kn else "8")
            calib.setdefault((ctr, kind), []).append(float(row["Counter_Value"]))
    if calib:
        print("== PMC calibration (1 GiB streamed per launch; counter unit = KB)")
        fac = {}
        for (ctr, kind), v in sorted(calib.items()):
            mean = sum(v) / len(v)
            fac[(ctr, kind)] = (1 << 30) / (mean * 1024.0) if mean else float("nan")
            print("  %-10s %-7s reported %.0f KB for 1048576 KB moved -> multiply by %.3f" % (ctr, kind, mean, fac[(ctr, kind)]))
        try:
            import json
            fe = fac.get(("FETCH_SIZE", "read8")); wr8 = fac.get(("WRITE_SIZE", "write8")); wr4 = fac.get(("WRITE_SIZE", "write4"))
            fsz = wsz = None
            for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
                ff = find(os.path.join(d, tag), "*counter_collection.csv")
                rows = [r for r in csv.DictReader(open(ff))
                        if r.get("Kernel_Name", "").split("(")[0].strip() in ("cn_env_kernel", "cn_env_kernel_fair") and r["Counter_Name"] == ctr]
                gmax = max(int(r.get("Grid_Size", 0) or 0) for r in rows)
                vals = [float(r["Counter_Value"]) for r in rows if int(r.get("Grid_Size", 0) or 0) == gmax]
                m = sum(vals) / max(1, len(vals))   # the one-launch-per-step leg (4096 envs per launch)
                if ctr == "FETCH_SIZE":
                    fsz = m
                else:
                    wsz = m
            # the env kernel's reads are 8-byte lanes; its writes are ~60 % 4-byte (observation) and ~40 % 8-byte (state)
            wfac = 0.6 * wr4 + 0.4 * wr8
            out = {"fetch_kb": fsz, "write_kb": wsz, "fetch_factor_read8": fe, "write_factor_mix": wfac,
                   "bytes_per_launch": fsz * 1024 * fe + wsz * 1024 * wfac, "csrc_hash": csrc_hash(),
                   "note": "per cn_env_kernel launch of the full 4096-env grid (next-step-reset, one launch per step), "
                           "corrected with this box's calibration; a stream-group launch of n envs moves n/4096 of it"}
            json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
            print("== corrected HBM traffic per launch: %.2f MB (fetch %.2f MB, write %.2f MB)" % (
                out["bytes_per_launch"] / 1e6, fsz * 1024 * fe / 1e6, wsz * 1024 * wfac / 1e6))
        except Exception as ex:  # noqa
            print("traffic.json not written:", ex)
    for f in sorted(glob.glob(os.path.join(d, "bench_*.log"))):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            import json
            j = json.loads(lines[-1])
            print("== %s: value %.4g env-steps/s, kernel_ms %.4f" % (os.path.basename(f), j["value"], j["roofline"]["kernel_ms"]))


if __name__ == "__main__":
    main(sys.argv[1])
