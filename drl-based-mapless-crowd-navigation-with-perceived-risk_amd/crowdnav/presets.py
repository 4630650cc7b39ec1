"""Scenario presets: the reference's training world and its scripted evaluation crowds, as data.

`presets_data.json` holds only numbers extracted from the reference tree (oracle/extract_presets.py):
  * the (vx, vy) table each crowd_behaviors/simulate_{crossing,towards,ahead}_{4,8,12,20}[_fast].py publishes
    (e.g. simulate_crossing_20.py:112-140), or the U(-vmax, vmax) bound of the random crowds
    (simulate_crowd.py:101-102, simulate_random_*.py)
  * the obstacle cylinders' initial poses of worlds/test_environment/turtlebot3_obstacle_N.world and of the
    training world turtlebot3_crowd_dense.world (WORLD:87-867)
README "Start testing" gives the evaluation settings: 5 x 5 m room, goal (-2, 2), start (1, 0),
min_scan_range 0.0."""
import json
import os

import numpy as np

from .config import Config

_DATA = None


def data():
    global _DATA
    if _DATA is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "presets_data.json")) as f:
            _DATA = json.load(f)
    return _DATA


def names():
    return sorted(data()["scripts"])


def training(n_envs=1, **kw):
    """The training set-up (launch/start_td3_training.launch): turtlebot3_crowd_dense.world, 14 obstacles driven
    by simulate_crowd.py, 2.8 m room.  Returns (Config, ped_init[N,14,2])."""
    poses = np.asarray(data()["worlds"]["turtlebot3_crowd_dense"], dtype=np.float64)
    cfg = Config(n_envs=n_envs, n_peds=len(poses), ped_mode=0, ped_vmax=data()["scripts"]["simulate_crowd"]["random"],
                 ped_cycle_ms=100 * len(poses), **kw)
    return cfg, np.broadcast_to(poses, (n_envs,) + poses.shape).copy()


def evaluation(kind, n, variant="", n_envs=1, **kw):
    """A scripted evaluation scenario: kind in {crossing, towards, ahead, random}, n in {4, 8, 12, 20},
    variant in {"", "fast", "highspeed"}.  Returns (Config, ped_init[N,n,2], ped_vel[N,n,2] or None)."""
    key = "simulate_%s_%d%s" % (kind, n, "_" + variant if variant else "")
    sc = data()["scripts"][key]
    poses = np.asarray(data()["worlds"]["test_%d" % n], dtype=np.float64)
    base = dict(n_envs=n_envs, n_peds=n, room_half=2.40, goal_x=-2.0, goal_y=2.0, start_x=1.0, start_y=0.0,
                spawn_x=1.0, spawn_y=0.0, min_scan_range=0.0, ped_cycle_ms=100 * n)
    base.update(kw)
    init = np.broadcast_to(poses, (n_envs,) + poses.shape).copy()
    if "vel" in sc:
        vel = np.asarray(sc["vel"], dtype=np.float64)
        return Config(ped_mode=1, **base), init, np.broadcast_to(vel, (n_envs,) + vel.shape).copy()
    return Config(ped_mode=0, ped_vmax=float(sc["random"]), **base), init, None
