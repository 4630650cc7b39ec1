#!/usr/bin/env python
"""Print the figures of a bench.py JSON line that DESIGN.md quotes.  Usage: python tools/show_bench.py file.json [...]"""
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        print(f, "ERR", e); continue
    c, r = j["config"], j["roofline"]
    M = lambda x: None if x is None else round(x / 1e6, 1)
    print("%s: value %.2f M env-steps/s, ms/step %.4f, decomposition %s" % (f, j["value"] / 1e6, j["ms_per_step"], c.get("decomposition", c["stream_groups"])))
    print("  samples", [M(x) for x in c["samples_env_steps_s"]], " probe", {k: M(v) for k, v in c["headline_choice"]["probe_env_steps_s"].items()})
    print("  legs", {k: M(v) for k, v in c["legs_env_steps_s"].items()}, " same-call", M(c["same_call_reset_value"]))
    print("  enqueue ms/step", {k: (None if v is None else round(v, 4)) for k, v in c["host_enqueue_ms_per_step"].items()})
    print("  roofline frac %.4f (f64 layout %.4f), kernel_ms %.4f, plateau %s, valu_busy %s, traffic %s [%s]" % (
        r["frac"], r["frac_f64_layout"], r["kernel_ms"], M(r["issue_bound_env_steps_s"]), r["valu_busy"], r["traffic"], r["traffic_source"]))
    for k, v in (c.get("other_configs") or {}).items():
        print("  %s: %.2f M, ms/step %.4f, %s, frac %.4f, legs %s" % (k, v["value"] / 1e6, v["ms_per_step"], v["decomposition"], v["roofline"]["frac"],
                                                                  {a: M(b) for a, b in v["legs_env_steps_s"].items()}))
    su = c.get("sustained")
    if su:
        print("  sustained %.2f M env-steps/s over %.2f s (%d launches), shader clock %.0f MHz (idle %.0f), burst/sustained %.3f" % (
            su["env_steps_s"] / 1e6, su["seconds"], su["launches"], su["clock_mhz"] or 0, su["clock_mhz_idle"] or 0, c["burst_over_sustained"]))
    cb = j.get("cpu_baseline")
    if cb:
        print("  cpu_baseline %.0f env-steps/s on %d cores (one core %.0f); reference python %s" % (cb["value"], cb["cores"], cb["one_core_value"], cb["reference_python"]["value"]))
