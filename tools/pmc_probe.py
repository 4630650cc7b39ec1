#!/usr/bin/env python
"""Small fixed workload for rocprofv3 --pmc passes: cn_env_kernel as one launch per step at the env counts given
(default 4096 and 16384), next-step reset, 200 pre-roll + 60 profiled-size steps each.  CN_LIB selects a library variant.
    rocprofv3 --pmc SQ_INSTS_VALU ... -d out -o pmc --output-format csv -- python tools/pmc_probe.py
    python tools/pmc_probe.py --parse out      -> per-env-step figures by grid size"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv, glob, collections
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"].split("(")[0].strip()
            if not kn.startswith("cn_env_kernel"):
                continue
            acc[(kn, int(r["Grid_Size"]) // 64)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for (kn, n), d in sorted(acc.items()):
            if n < 1024:
                continue
            # the last 60 dispatches of each size are the steady-state ones (after the pre-roll)
            print("%s  %d envs per launch:" % (kn, n))
            for k in sorted(d):
                v = d[k][-60:]; m = sum(v) / len(v)
                print("    %-22s %14.6g per launch  %10.4g per env" % (k, m, m / n))
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import _abi
if os.environ.get("CN_LIB"):
    _abi.LIB_PATH = os.path.abspath(os.environ["CN_LIB"]); _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv
for N in [int(x) for x in (sys.argv[1:] or ["4096", "16384"])]:
    env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400)); env.reset()
    if os.environ.get("CN_ABLATE"):
        env.L.cn_debug_set_ablate(env.h, int(os.environ["CN_ABLATE"]))
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
    for i in range(260):
        env.step(acts[i % 16], auto_reset="next")
    torch.cuda.synchronize(); env.close()
print("pmc_probe done")
