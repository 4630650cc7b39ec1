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


def training(n_envs=1, drop_cospawned=False, **kw):
    """The training set-up (launch/start_td3_training.launch): turtlebot3_crowd_dense.world, 14 obstacles driven
    by simulate_crowd.py, 2.8 m room.  Returns (Config, ped_init[N,P,2]).

    drop_cospawned: the world file creates obstacles 7-14 at ONE point (turtlebot3_crowd_dense.world:447-867, all at
    (0.22, 0.54)), and gazebo/reset_simulation puts them back there every episode: eight interpenetrating rigid bodies that the
    physics engine throws apart.  What happens to them cannot be restated (no reference source), but the published training
    log is what this build's simulator gives with ONLY obstacles 1-6 in the room: the published top_8 actor (ep 2500, training
    noise sigma = 1) succeeds in 0.60 of the episodes of the log's last 500, and here 0.203 / 0.359 / 0.426 / 0.508 / 0.598 /
    0.660 with 14 / 10 / 8 / 7 / 6 / 4 walkers (profiles/r03/reference_policy_eval.txt).  True = those six walkers; the crowd
    node's round still takes 1.4 s (it publishes to all fourteen names, 0.1 s each)."""
    poses = np.asarray(data()["worlds"]["turtlebot3_crowd_dense"], dtype=np.float64)
    cycle = 100 * len(poses)
    if drop_cospawned:
        first = {}
        for i, xy in enumerate(map(tuple, poses)):
            first.setdefault(xy, []).append(i)
        poses = poses[[ix[0] for ix in first.values() if len(ix) == 1]]
    cfg = Config(n_envs=n_envs, n_peds=len(poses), ped_mode=0, ped_vmax=data()["scripts"]["simulate_crowd"]["random"],
                 ped_cycle_ms=cycle, **kw)
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
