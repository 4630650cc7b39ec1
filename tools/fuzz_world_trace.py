#!/usr/bin/env python
"""One environment of a world tools/fuzz_parity.py reported, step by step through cn_step with the actions the fuzzer's sequence /
policy forms draw: after every step the kernel's internal state of that environment (cn_debug_env: track table, confirmed objects,
collision probability, ego score, waypoint, box size) against the oracle's, until the first field that differs -- which is usually a
few steps before the first observation that does.
    python tools/fuzz_world_trace.py "{'n_envs': 300, ...}" <env> <steps>"""
import ast, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import Config
from crowdnav.env import VecEnv
from oracle import oracle
kw = ast.literal_eval(sys.argv[1]); E = int(sys.argv[2]); T = int(sys.argv[3])
np.set_printoptions(precision=17, linewidth=220)
cfg = Config(**kw); env = VecEnv(cfg); env.enable_f64_obs(); orc = oracle.Oracle(cfg.as_dict())
rng = np.random.default_rng(int(kw["seed"]) ^ 0x5eed); N = cfg.n_envs
env.reset(); torch.cuda.synchronize(); orc.reset()
Tc = int(rng.choice([1, 5, 12, 20]))
keys = ["n_tracks", "n_confirmed", "n_entries", "status", "bb", "collision_prob", "ego_score", "wp", "track_pose", "track_dist", "track_speed", "track_vel", "track_t", "track_dqlen"]
t = 0
while t < T:
    A_ = np.stack([rng.uniform(0, 0.22, (Tc, N)), rng.uniform(-2, 2, (Tc, N))], 2).astype(np.float32)
    for i in range(Tc):
        env.step(torch.from_numpy(A_[i]).cuda(), auto_reset="next"); torch.cuda.synchronize()
        oc, rc, dc, ic = orc.step(A_[i].astype(np.float64), auto_reset="next")
        g, c = env.debug_env(E), orc.debug(E)
        diffs = [k for k in keys if not np.array_equal(np.asarray(g[k]), np.asarray(c[k]))]
        soft = [k for k in diffs if np.asarray(g[k]).shape == np.asarray(c[k]).shape and np.allclose(np.asarray(g[k], dtype=float), np.asarray(c[k], dtype=float), rtol=1e-12, atol=1e-14)]
        if soft: print("   (within 1e-12: %s)" % soft)
        diffs = [k for k in diffs if k not in soft]
        od = np.nonzero(env.obs_f64.cpu().numpy()[E] != oc[E])[0]
        print("step %d: debug fields differing %s; obs cols differing %s; done %d/%d idx %s / %s" % (t, diffs, od[:20], env.done[E].item(), dc[E], env.topk_idx[E].cpu().numpy(), ic[E]))
        if diffs or len(od):
            for k in diffs:
                print("  ", k, "\n     gpu", np.asarray(g[k]).tolist(), "\n     cpu", np.asarray(c[k]).tolist())
            print("   cpu entry_cp", c["entry_cp"].tolist(), "\n   cpu entry_ego", c["entry_ego"].tolist())
            print("   gpu obs tail", env.obs_f64.cpu().numpy()[E][-(7 + 4 * cfg.k_obstacles):].tolist())
            print("   cpu obs tail", oc[E][-(7 + 4 * cfg.k_obstacles):].tolist())
            print("   gpu cprob/ego", g["collision_prob"], g["ego_score"], " cpu", c["collision_prob"], c["ego_score"])
            print("   track_pose gpu", g["track_pose"].tolist()); print("   track_vel gpu", g["track_vel"].tolist()); print("   track_dist gpu", g["track_dist"].tolist())
            sys.exit(0)
        t += 1
print("no difference")
