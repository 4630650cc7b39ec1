#!/usr/bin/env python
"""tools/profc/run.py [envs=4096] [steps=100] [preroll=200] -> gpurun_out/profc/counters_<envs>.npz

Runs the headline workload (bench.py's configuration: 4096 envs x 20 pedestrians x 360 rays, open-loop actions, next-step reset,
one cn_step launch per step) on the REGION-COUNTER build (lib/prof/libcrowdnav_profc.so, tools/profc/build.sh), zeroes the counters
after the pre-roll and copies them out after `steps` launches.  Every counter holds (wavefront executions << 32) | lane executions
of one source region of the step kernel.  tools/profc/report.py turns the file into a per-line / per-stage dynamic instruction ledger."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd")
sys.path.insert(0, PKG)
import numpy as np
import torch
from crowdnav import _abi
_abi.LIB_PATH = os.path.join(PKG, "lib", "prof", "libcrowdnav_profc.so"); _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
PRE = int(sys.argv[3]) if len(sys.argv) > 3 else 200
sym = json.load(open(os.path.join(ROOT, "tools", "profc", "symbols.json")))
L = _abi.lib()
L.cn_debug_profc.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_void_p]
n_cnt = int(sym["cnts_u64"])
buf = torch.zeros(n_cnt, dtype=torch.int64, device="cuda")


def dump(zero):
    torch.cuda.synchronize()
    rc = L.cn_debug_profc(buf.data_ptr(), int(sym["anchor_before_bytes"]), n_cnt, int(zero), None)
    assert rc == 0
    torch.cuda.synchronize()
    return buf.cpu().numpy().view(np.uint64).copy()


# CN_PROFC_CFG: extra Config fields as "key=value,key=value" (e.g. ped_mode=2 for the social-force kernels)
extra = {}
for kv in filter(None, os.environ.get("CN_PROFC_CFG", "").split(",")):
    k_, v_ = kv.split("="); extra[k_] = float(v_) if "." in v_ else int(v_)
env = VecEnv(Config(**{**dict(n_envs=N, n_peds=20, n_rays=360, k_obstacles=8, max_steps=1000, seed=1234, ped_cycle_ms=1400, room_half=1.40), **extra}),
             arbitration=os.environ.get("CN_ARB", "auto"))
env.reset()
g = torch.Generator(device="cuda").manual_seed(1234)
acts = torch.stack([torch.rand((64, N), generator=g, device="cuda") * 0.22, torch.rand((64, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
for i in range(PRE):
    env.step(acts[i % 64], auto_reset="next")
ep0 = env.counters()[:, 8].sum().item() if hasattr(env, "counters") else 0
dump(True)
for i in range(STEPS):
    env.step(acts[(PRE + i) % 64], auto_reset="next")
c = dump(False)
kern = env.kernel_name("step") if hasattr(env, "kernel_name") else "?"
out = os.path.join(ROOT, "gpurun_out", "profc"); os.makedirs(out, exist_ok=True)
np.savez_compressed(os.path.join(out, "counters_%d%s.npz" % (N, os.environ.get("CN_PROFC_TAG", ""))), counters=c, envs=N, steps=STEPS, preroll=PRE, kernel=kern)
nz = int((c != 0).sum())
print("profc: %d envs x %d launches of %s; %d of %d counters non-zero; wave executions total %d, lane executions total %d" % (
    N, STEPS, kern, nz, n_cnt, int((c >> np.uint64(32)).sum()), int((c & np.uint64(0xffffffff)).sum())))
