import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


def load_seq(name):
    import numpy as np
    z = np.load(os.path.join(GOLDEN, "seq_%s.npz" % name))
    kw = {str(k): float(v) for k, v in zip(z["config_keys"], z["config_vals"])}
    for k in ("n_peds", "max_steps", "seed", "k_obstacles", "obs_layout", "geos_untyped_empty", "ped_contact", "risk_mode", "dt_ms", "py2_round", "ped_mode", "scan_f32", "waypoint_reward"):
        if k in kw:
            kw[k] = int(kw[k])
    return z, kw
