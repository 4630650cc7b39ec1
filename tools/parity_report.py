#!/usr/bin/env python
"""GPU-vs-oracle parity report (diagnostic; the assertions live in tests/test_gpu_parity.py).
Usage: python tools/parity_report.py [--envs 64] [--steps 200] [--peds 20] [--room 1.4] [--seed 7]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--peds", type=int, default=20)
    ap.add_argument("--rays", type=int, default=360)
    ap.add_argument("--room", type=float, default=1.4)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--max-steps", type=int, default=120)
    ap.add_argument("--min-scan", type=float, default=0.12)
    ap.add_argument("--verbose", type=int, default=5)
    ap.add_argument("--reset-mode", default="same", choices=["same", "next"])
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--risk-mode", type=int, default=0)
    ap.add_argument("--contact", type=int, default=0)
    ap.add_argument("--geos", type=int, default=0)
    ap.add_argument("--layout", type=int, default=0)
    ap.add_argument("--vmax", type=float, default=0.2)
    ap.add_argument("--dt-ms", type=int, default=150)
    ap.add_argument("--py2", type=int, default=0, help="cn_config.py2_round")
    ap.add_argument("--ped-mode", type=int, default=0, help="2 = social-force pedestrians")
    ap.add_argument("--sf-tick", type=int, default=0, help="cn_config.sf_tick_ms")
    ap.add_argument("--scan-f32", type=int, default=0, help="cn_config.scan_f32")
    ap.add_argument("--wheel-accel", type=float, default=0.0, help="cn_config.wheel_accel")
    ap.add_argument("--waypoint-reward", type=int, default=200, help="cn_config.waypoint_reward")
    ap.add_argument("--policy", type=int, default=0, help="T > 0: closed loop -- cn_rollout_policy launches of T periods (a random-init TD3 actor "
                    "inside the step kernel, sigma = 1 exploration); the oracle replays the actions the kernel recorded")
    a = ap.parse_args()
    import torch
    from crowdnav import Config
    from crowdnav.env import VecEnv
    from oracle import oracle

    cfg = Config(n_envs=a.envs, n_peds=a.peds, n_rays=a.rays, room_half=a.room, seed=a.seed, max_steps=a.max_steps,
                 min_scan_range=a.min_scan, k_obstacles=a.k, risk_mode=a.risk_mode, ped_contact=a.contact,
                 geos_untyped_empty=a.geos, obs_layout=a.layout, ped_vmax=a.vmax, dt_ms=a.dt_ms, py2_round=a.py2, ped_mode=a.ped_mode, sf_tick_ms=a.sf_tick,
                 scan_f32=a.scan_f32, wheel_accel=a.wheel_accel, waypoint_reward=a.waypoint_reward)
    print("== parity_report", " ".join(sys.argv[1:]))
    env = VecEnv(cfg)
    env.enable_f64_obs()
    print("kernels: step %s, same-call %s" % (env.kernel_name("step"), env.kernel_name("same")))
    orc = oracle.Oracle(cfg.as_dict())
    oracle.set_num_threads()
    n = a.rays - 1
    if a.layout:
        n = a.rays - 1 - 3          # layouts 1 / 2 have their own tails; the column split below is only indicative there
    o_g = env.reset(); torch.cuda.synchronize()
    o_c = orc.reset()
    g64 = env.obs_f64.cpu().numpy()
    print("reset: max |obs diff| = %.3g, exact rows %d/%d" % (np.abs(g64 - o_c).max(), int((g64 == o_c).all(1).sum()), a.envs))
    rng = np.random.default_rng(a.seed)
    tot = dict(obs_rows=0, scan=0, tail=0, feat=0, reward=0, done=0, idx=0, rows=0)
    shown = 0
    if a.policy:
        # closed loop: the policy kernel decides, the oracle is fed what it decided (float32 observations: the kernel's own output width)
        from crowdnav.td3 import Agent
        assert a.reset_mode == "next"
        print("kernel: policy %s" % env.kernel_name("policy"))
        agent = Agent(obs_dim=cfg.obs_dim, device="cuda:0", seed=a.seed, memory_size=16)
        T, N, D, K = a.policy, a.envs, env.D, env.K
        traj = dict(action=torch.zeros((T, N, 2), device="cuda"), obs=torch.zeros((T, N, D), device="cuda"), reward=torch.zeros((T, N), device="cuda"),
                    done=torch.zeros((T, N), dtype=torch.uint8, device="cuda"), topk_idx=torch.zeros((T, N, K), dtype=torch.int32, device="cuda"))
        moved = 0.0
        for t0 in range(0, a.steps, T):
            env.rollout_policy(agent, T, traj=traj)
            torch.cuda.synchronize()
            A = traj["action"].cpu().numpy(); O = traj["obs"].cpu().numpy(); R = traj["reward"].cpu().numpy(); Dn = traj["done"].cpu().numpy(); I = traj["topk_idx"].cpu().numpy()
            moved += float(np.abs(A[:, :, 1]).mean())
            for t in range(T):
                oc, rc, dc, ic = orc.step(A[t].astype(np.float64), auto_reset="next")
                oc = oc.astype(np.float32)
                bad_rows = ~(O[t] == oc).all(1)
                tot["rows"] += N; tot["obs_rows"] += int(bad_rows.sum())
                tot["scan"] += int((O[t][:, :n] != oc[:, :n]).any(1).sum()); tot["tail"] += int((O[t][:, n:n + 7] != oc[:, n:n + 7]).any(1).sum())
                tot["feat"] += int((O[t][:, n + 7:] != oc[:, n + 7:]).any(1).sum()); tot["reward"] += int((R[t] != rc.astype(np.float32)).sum())
                tot["done"] += int((Dn[t] != dc).sum()); tot["idx"] += int((I[t] != ic).any(1).sum())
                if bad_rows.any() and shown < a.verbose:
                    shown += 1
                    e = int(np.nonzero(bad_rows)[0][0]); cols = np.nonzero(O[t][e] != oc[e])[0]
                    print("period %d env %d: bad cols %s gpu %s cpu %s" % (t0 + t, e, cols[:12], O[t][e][cols[:8]], oc[e][cols[:8]]))
        print("TOTAL over %d env-steps: %s" % (tot["rows"], tot))
        cg = env.counters().cpu().numpy(); cc = orc.counters()
        print("counters equal:", np.array_equal(cg[:, :6], cc), " episodes finished:", int(cg[:, 8].sum()), " mean |w| commanded %.3f" % (moved / max(1, a.steps // T)))
        return
    for t in range(a.steps):
        act = np.stack([rng.uniform(0, 0.22, a.envs), rng.uniform(-2, 2, a.envs)], 1).astype(np.float32)
        env.step(torch.from_numpy(act).cuda(), auto_reset=a.reset_mode, want_final=True)
        torch.cuda.synchronize()
        oc, rc, dc, ic, fc = orc.step(act.astype(np.float64), auto_reset=a.reset_mode, want_final=True)
        og = env.obs_f64.cpu().numpy(); rg = env.reward.cpu().numpy(); dg = env.done.cpu().numpy(); ig = env.topk_idx.cpu().numpy()
        fg = env.final_obs.cpu().numpy()
        bad_rows = ~(og == oc).all(1)
        tot["rows"] += a.envs
        tot["obs_rows"] += int(bad_rows.sum())
        tot["scan"] += int((og[:, :n] != oc[:, :n]).any(1).sum())
        tot["tail"] += int((og[:, n:n + 7] != oc[:, n:n + 7]).any(1).sum())
        tot["feat"] += int((og[:, n + 7:] != oc[:, n + 7:]).any(1).sum())
        tot["reward"] += int((rg != rc.astype(np.float32)).sum())
        tot["done"] += int((dg != dc).sum())
        tot["idx"] += int((ig != ic).any(1).sum())
        fin_bad = int((fg != fc.astype(np.float32)).any(1).sum()) if a.reset_mode == "same" else 0
        if (bad_rows.any() or (dg != dc).any() or fin_bad) and shown < a.verbose:
            shown += 1
            e = int(np.nonzero(bad_rows | (dg != dc))[0][0]) if (bad_rows.any() or (dg != dc).any()) else 0
            cols = np.nonzero(og[e] != oc[e])[0]
            print("step %d env %d: bad cols %s\n   gpu %s\n   cpu %s  reward %s/%s done %s/%s idx %s/%s final_bad=%d" % (
                t, e, cols[:12], og[e][cols[:8]], oc[e][cols[:8]], rg[e], rc[e], dg[e], dc[e], ig[e], ic[e], fin_bad))
            dgb = env.debug_env(e); dcb = orc.debug(e)
            print("   tracks gpu %d cpu %d | nconf %d/%d | bb %r/%r | wp %s/%s | ego %r/%r | status %d/%d" % (
                dgb["n_tracks"], dcb["n_tracks"], dgb["n_confirmed"], dcb["n_confirmed"], dgb["bb"], dcb["bb"],
                dgb["wp"], dcb["wp"], dgb["ego_score"], dcb["ego_score"], dgb["status"], dcb["status"]))
            k = min(dgb["n_tracks"], dcb["n_tracks"])
            if k and not np.array_equal(dgb["track_pose"][:k], dcb["track_pose"][:k]):
                print("   track poses differ:\n", dgb["track_pose"][:k], "\n", dcb["track_pose"][:k])
            rs_g, rs_c = dgb["robot"], orc.sim_state(e)[0]
            print("   robot gpu %s\n   robot cpu %s" % (rs_g, rs_c))
    print("TOTAL over %d env-steps: %s" % (tot["rows"], tot))
    cg = env.counters().cpu().numpy(); cc = orc.counters()
    print("counters equal:", np.array_equal(cg[:, :6], cc), " gpu status bits:", np.unique(cg[:, 6]))
    print("tracks: gpu mean %.2f max %d" % (cg[:, 7].mean(), cg[:, 7].max()))


if __name__ == "__main__":
    main()
