#!/usr/bin/env python
"""cn_div / cn_sqrt (crowdnav_device.h: the compiler's correctly rounded expansions minus their range handling) against IEEE
division / square root on 240 M / 80 M arguments of the kinds the kernels feed them (tests/test_gpu_parity.py checks 4e5 per run):
    python tools/check_device_div_sqrt.py        -> profiles/r06/device_div_sqrt.txt"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch, crowdnav
L = C.CDLL(crowdnav._abi.build_timing())
L.cn_debug_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
def dm(op, x, y):
    xd = torch.tensor(x, device="cuda"); yd = torch.tensor(y, device="cuda"); out = torch.empty_like(xd); torch.cuda.synchronize()
    assert L.cn_debug_math(op, xd.data_ptr(), yd.data_ptr(), out.data_ptr(), xd.numel(), None) == 0
    return out.cpu().numpy()
rng = np.random.default_rng(1)
tot = bad = 0
for rep in range(12):
    n = 20_000_000
    kind = rep % 4
    if kind == 0: a = rng.uniform(-5, 5, n); b = rng.uniform(-5, 5, n)
    elif kind == 1: a = rng.integers(-4000, 4000, n) / 1000.0 - rng.integers(-4000, 4000, n) / 1000.0; b = rng.integers(-4000, 4000, n) / 1000.0 - rng.integers(-4000, 4000, n) / 1000.0
    elif kind == 2: a = rng.uniform(-1, 1, n); b = np.exp(rng.uniform(np.log(1e-6), np.log(1e3), n)) * rng.choice([-1.0, 1.0], n)
    else: a = rng.integers(0, 4000, n) / 1000.0; b = rng.uniform(0.05, 0.25, n)
    b[b == 0] = 1.0; a[a == 0] = 0.0
    got = dm(1, a, b); ref = a / b
    m = got != ref
    tot += n; bad += int(m.sum())
    if m.any():
        i = np.nonzero(m)[0][:3]
        print("kind", kind, "mismatches", int(m.sum()), [(float(a[j]).hex(), float(b[j]).hex(), float(got[j]).hex(), float(ref[j]).hex()) for j in i])
print("cn_div vs IEEE division: %d of %d differ" % (bad, tot))
# the same for cn_sqrt
bad = tot = 0
for rep in range(4):
    n = 20_000_000
    x = rng.uniform(0, 30, n) if rep % 2 == 0 else (rng.integers(0, 4000, n) / 1000.0) ** 2 + (rng.integers(0, 4000, n) / 1000.0) ** 2
    got = dm(0, x, x); ref = np.sqrt(x); m = got != ref; tot += n; bad += int(m.sum())
print("cn_sqrt vs IEEE sqrt: %d of %d differ" % (bad, tot))
