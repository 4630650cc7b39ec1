#!/usr/bin/env python
"""Does the return rise?  The as-logged training run (16 envs, one update of 128 per env-step, cn_td3_update, waypoint_reward 0,
seed 0) for a fixed number of launches; prints the summary line.  Run twice to check that the run is reproducible."""
import argparse, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
from crowdnav import train as T
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 18000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
out = tempfile.mkdtemp()
a = argparse.Namespace(scenario="training_as_logged", envs=16, launches=launches, max_steps=1000, updates=16, batch=128, memory=1_000_000,
                       checkpoint_every=10 ** 9, log_every=1000, ped_vmax=None, seed=seed, device=0, out=out, csv=False, load=None,
                       load_episode=0, evaluate=False, episodes_per_env=1, graphs=1, waypoint_reward=0, scan_f32=None, wheel_accel=None,
                       reset_mode="next", max_csv_rows=100000, time_limit=0.0, learner="fused")
agent, episodes = T.train(a)
import torch
print("episodes", episodes, "actor checksum %.9e" % float(sum(p.double().abs().sum() for p in agent.actor.parameters())))
