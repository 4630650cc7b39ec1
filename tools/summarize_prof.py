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
    for tdir, what in (("trace", "bench.py --steps 300 --warmup 30"), ("trace_driver", "the driver's command: bench.py --steps 20 --warmup 5"),
                       ("pol_trace", "tools/policy_perf.py 4096 --profile (cn_rollout_policy, %s periods per launch)" % os.environ.get("CN_PROFILE_POL_STEPS", "100"))):
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
                if not (kn.startswith("cn_env_kernel") or kn.startswith("cn_policy_kernel")):
                    continue
                g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
                legs.setdefault((kn, g), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            # bench.py legs: cn_env_kernel_same (same-call reset), cn_env_kernel at the full grid (one launch per step),
            # cn_env_kernel at grid / G (the stream-group leg: G launches in flight, so durations overlap)
            print("== env kernel average duration by (kernel, grid = 64 x envs per launch), %s:" % what)
            for (kn, g), dd in sorted(legs.items()):
                print("  %-22s grid %7d (%5d envs)  launches %5d  avg %.1f us  min %.1f  max %.1f" % (
                    kn, g, g // 64, len(dd), sum(dd) / len(dd) / 1e3, min(dd) / 1e3, max(dd) / 1e3))
    # ---- per-kernel counters: every cn_env_kernel* variant the bench command launched, at its largest grid (the one-launch-per-step
    # legs and the sequence leg run the full 4096-env grid; stream-group launches of the same kernel are smaller and skipped).
    # A sequence kernel's launch covers `seq_steps` control periods: its counters are divided by that.
    import json
    seq_steps = float(os.environ.get("CN_PROFILE_SEQ_STEPS", "300"))
    pol_steps = float(os.environ.get("CN_PROFILE_POL_STEPS", "100"))
    per_kernel = {}
    # pol_*: the same passes over tools/policy_perf.py --profile (every cn_policy_kernel launch there covers pol_steps periods)
    traj_steps = float(os.environ.get("CN_PROFILE_TRAJ_STEPS", "50"))
    # traj_*: tools/seq_traj_profile.py -- the sequence kernel writing every step into trajectory buffers; keyed "<kernel>+traj"
    for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pol_fetch", "pol_write", "pol_sq", "traj_fetch", "traj_write"):
        f = find(os.path.join(d, tag), "*counter_collection.csv")
        if not f:
            print("== %s: no counter csv" % tag)
            continue
        allrows = list(csv.DictReader(open(f)))
        prefix = "cn_policy_kernel" if tag.startswith("pol_") else "cn_env_kernel"
        names = sorted({r.get("Kernel_Name", "").split("(")[0].strip() for r in allrows if r.get("Kernel_Name", "").startswith(prefix)})
        for kname in names:
            rows = [r for r in allrows if r.get("Kernel_Name", "").split("(")[0].strip() == kname]
            gmax = max([int(r.get("Grid_Size", 0) or 0) for r in rows] or [0])
            if gmax < 1024 * 64:
                continue                         # not a bench leg (bench.py's stream probe steps a few small handles)
            acc, cnt = {}, {}
            for row in rows:
                if int(row.get("Grid_Size", 0) or 0) != gmax:
                    continue
                k = row["Counter_Name"]; v = float(row["Counter_Value"])
                acc[k] = acc.get(k, 0.0) + v; cnt[k] = cnt.get(k, 0) + 1
            if not acc:
                continue
            steps = traj_steps if tag.startswith("traj_") else seq_steps if "_seq" in kname else pol_steps if kname.startswith("cn_policy_kernel") else 1.0
            if tag.startswith("traj_"):
                if "_seq" not in kname:
                    continue                     # (its reset launch)
                kname = kname + "+traj"
            e = per_kernel.setdefault(kname, {"envs_per_launch": gmax // 64, "steps_per_launch": steps, "raw": {}})
            print("== %s (per %s dispatch of %d envs%s, mean of %s)" % (tag, kname, gmax // 64, " x %d steps" % steps if steps > 1 else "", sorted(set(cnt.values()))))
            for k in sorted(acc):
                print("  %-24s %.6g" % (k, acc[k] / cnt[k]))
                e["raw"][k] = acc[k] / cnt[k]
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
    fac = {}
    if calib:
        print("== PMC calibration (1 GiB streamed per launch; counter unit = KB)")
        for (ctr, kind), v in sorted(calib.items()):
            mean = sum(v) / len(v)
            fac[(ctr, kind)] = (1 << 30) / (mean * 1024.0) if mean else float("nan")
            print("  %-10s %-7s reported %.0f KB for 1048576 KB moved -> multiply by %.3f" % (ctr, kind, mean, fac[(ctr, kind)]))
    counters = {"csrc_hash": csrc_hash(), "kernels": {},
                "source": "rocprofv3 --pmc SQ_* passes of the bench command (tools/profile.sh), per kernel at its largest grid; a sequence kernel's "
                          "figures are per control period (launch / steps_per_launch): valu_busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32)"}
    traffic = {"csrc_hash": csrc_hash(), "kernels": {},
               "note": "HBM bytes per launch and per env-step of each kernel at its largest grid (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate "
                       "passes, corrected with this box's calibration: reads are 8-byte lanes; writes ~60 % 4-byte (observation), ~40 % 8-byte (state))"}
    fe = fac.get(("FETCH_SIZE", "read8")); wr8 = fac.get(("WRITE_SIZE", "write8")); wr4 = fac.get(("WRITE_SIZE", "write4"))
    for kname, e in sorted(per_kernel.items()):
        sq, n, steps = e["raw"], float(e["envs_per_launch"]), e["steps_per_launch"]
        if "SQ_INSTS_VALU" in sq and "SQ_BUSY_CYCLES" in sq:
            # SQ_ACTIVE_INST_VALU counts quad-cycles per wave (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"); SQ_BUSY_CYCLES is
            # summed over the 32 shader engines (8 XCDs x 4) in cycles, so the launch lasted SQ_BUSY_CYCLES / 32 cycles on 1024 SIMDs
            kcycles = sq["SQ_BUSY_CYCLES"] / 32.0
            counters["kernels"][kname] = {
                "valu_busy": 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (1024.0 * kcycles) if "SQ_ACTIVE_INST_VALU" in sq else None,
                "wave_instr_per_env_step": {"valu": sq["SQ_INSTS_VALU"] / n / steps, "salu": sq.get("SQ_INSTS_SALU", 0) / n / steps,
                                            "lds": sq.get("SQ_INSTS_LDS", 0) / n / steps},
                "wave_quad_cycles_per_env_step": sq.get("SQ_WAVE_CYCLES", 0) / n / steps,
                "wait_any_frac": sq.get("SQ_WAIT_ANY", 0) / max(1.0, sq.get("SQ_WAVE_CYCLES", 1)),
                "launch_cycles": kcycles, "envs_per_launch": n, "steps_per_launch": steps,
                "lds_bank_conflict_frac": (sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]) if sq.get("SQ_LDS_IDX_ACTIVE") else None}
            c = counters["kernels"][kname]
            print("== counters %-26s VALU busy %s, %.0f VALU + %.0f SALU + %.0f LDS wave instructions per env-step" % (
                kname, "%.3f" % c["valu_busy"] if c["valu_busy"] is not None else "-", c["wave_instr_per_env_step"]["valu"],
                c["wave_instr_per_env_step"]["salu"], c["wave_instr_per_env_step"]["lds"]))
        if "FETCH_SIZE" in sq and "WRITE_SIZE" in sq and fe and wr4 and wr8:
            wfac = 0.6 * wr4 + 0.4 * wr8
            fb, wb = sq["FETCH_SIZE"] * 1024 * fe, sq["WRITE_SIZE"] * 1024 * wfac
            traffic["kernels"][kname] = {"fetch_kb": sq["FETCH_SIZE"], "write_kb": sq["WRITE_SIZE"], "fetch_factor_read8": fe, "write_factor_mix": wfac,
                                         "bytes_per_launch": fb + wb, "envs_per_launch": n, "steps_per_launch": steps,
                                         "bytes_per_env_step": (fb + wb) / n / steps, "fetch_bytes_per_env_step": fb / n / steps,
                                         "write_bytes_per_env_step": wb / n / steps}
            print("== traffic  %-26s %.2f MB per launch = %.0f B per env-step (fetch %.0f, write %.0f)" % (
                kname, (fb + wb) / 1e6, (fb + wb) / n / steps, fb / n / steps, wb / n / steps))
    if counters["kernels"]:
        json.dump(counters, open(os.path.join(d, "counters.json"), "w"), indent=1)
    if traffic["kernels"]:
        json.dump(traffic, open(os.path.join(d, "traffic.json"), "w"), indent=1)
    for f in sorted(glob.glob(os.path.join(d, "bench_*.log"))):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            j = json.loads(lines[-1])
            print("== %s: value %.4g env-steps/s, kernel_ms %.4f" % (os.path.basename(f), j["value"], j["roofline"]["kernel_ms"]))


if __name__ == "__main__":
    main(sys.argv[1])
