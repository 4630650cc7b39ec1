#!/usr/bin/env python
"""Extract the reference's scripted-crowd scenarios as DATA (container-only; needs /root/reference).

  velocity tables  crowd_behaviors/simulate_{crossing,towards,ahead,random}_{4,8,12,20}[_fast|_highspeed].py:
                   the (vx, vy) each `move_model('obstacle_i', pose, vx, vy, 0)` call publishes, evaluated
                   with the script's own `speed` constant (e.g. simulate_crossing_20.py:112-140)
  initial poses    turtlebot3_gazebo/worlds/test_environment/turtlebot3_obstacle_N.world, the first
                   <pose> of each <model name='obstacle_i'> (e.g. turtlebot3_obstacle_20.world:85-86)
  training world   worlds/turtlebot3_crowd_dense.world obstacle poses (WORLD:87-867)

Output: drl-..._amd/crowdnav/presets_data.json (numbers only).  Scripts whose velocities are random draws
(simulate_random_*: random.uniform) are recorded as {"random": vmax} instead of a table."""
import ast
import glob
import json
import os
import re

REF = "/root/reference"
CB = os.path.join(REF, "turtlebot3_rl_sim/src/crowd_behaviors")
WORLDS = os.path.join(REF, "turtlebot3_simulations/turtlebot3_gazebo/worlds")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd", "crowdnav", "presets_data.json")


def eval_num(node, env):
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
        return float(node.value)
    if isinstance(node, ast.Name) and node.id in env:
        return env[node.id]
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
        v = eval_num(node.operand, env)
        return None if v is None else -v
    if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Mult, ast.Div, ast.Add, ast.Sub)):
        a, b = eval_num(node.left, env), eval_num(node.right, env)
        if a is None or b is None:
            return None
        return {ast.Mult: a * b, ast.Div: a / b if b else None, ast.Add: a + b, ast.Sub: a - b}[type(node.op)]
    return None


def velocity_table(path):
    src = open(path).read()
    tree = ast.parse(src)
    table, env, vmax, hold = {}, {}, None, None
    for fn in ast.walk(tree):
        if isinstance(fn, ast.FunctionDef) and fn.name == "moving_1":
            for node in ast.walk(fn):
                if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
                    v = eval_num(node.value, env)
                    if v is not None:
                        env[node.targets[0].id] = v
                if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "uniform":
                    a = eval_num(node.args[1], env)
                    vmax = a if vmax is None else max(vmax, a)
                if isinstance(node, ast.Compare) and isinstance(node.left, ast.Name) and node.left.id == "elapsed_time":
                    hold = eval_num(node.comparators[0], env)
            for node in ast.walk(fn):
                if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "move_model":
                    name = node.args[0].value
                    i = int(name.split("_")[1])
                    vx, vy = eval_num(node.args[2], env), eval_num(node.args[3], env)
                    if vx is None or vy is None:
                        table = None
                        break
                    table[i] = [vx, vy]
    n = len(re.findall(r"model_(\d+)_index = ", src))
    if table is None or not table:
        return dict(n=n, random=vmax, hold_s=hold)
    return dict(n=len(table), vel=[table[i] for i in sorted(table)], speed=env.get("speed"), hold_s=hold)


def world_poses(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"<model name=['\"]obstacle_(\d+)['\"]>\s*<pose(?: frame='')?>([^<]+)</pose>", txt):
        i = int(m.group(1))
        if i not in out:
            x, y = m.group(2).split()[:2]
            out[i] = [float(x), float(y)]
    return [out[i] for i in sorted(out)]


def main():
    data = dict(scripts={}, worlds={})
    for p in sorted(glob.glob(os.path.join(CB, "simulate_*.py"))):
        data["scripts"][os.path.basename(p)[:-3]] = velocity_table(p)
    for n in (4, 8, 12, 20):
        data["worlds"]["test_%d" % n] = world_poses(os.path.join(WORLDS, "test_environment", "turtlebot3_obstacle_%d.world" % n))
    for name in ("turtlebot3_crowd_dense", "turtlebot3_crowd_sparse"):
        p = os.path.join(WORLDS, name + ".world")
        if os.path.exists(p):
            data["worlds"][name] = world_poses(p)
    json.dump(data, open(OUT, "w"), indent=0, sort_keys=True)
    for k, v in data["scripts"].items():
        print("%-36s n=%-3s %s" % (k, v.get("n"), "table speed=%s" % v.get("speed") if "vel" in v else "random vmax=%s" % v.get("random")))
    for k, v in data["worlds"].items():
        print("%-36s %d poses" % (k, len(v)))


if __name__ == "__main__":
    main()
