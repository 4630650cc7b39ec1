"""Runs the reference's OWN Python (environment_stage_1_nobonus.py + utils.py, loaded unmodified
from /root/reference) without ROS, Gazebo, shapely or Python 2.

TEST INFRASTRUCTURE, container-only: /root/reference does not exist on the GPU box, so nothing
under tests/ -m gpu, smoke() or bench.py imports this module.  It exists to (a) pin
oracle/cn_oracle.c against the real reference and (b) emit the golden vectors under tests/golden/
(oracle/make_goldens.py).

Ingredients:
  1. sys.modules stubs for rospy / *_msgs / std_srvs / tf / shapely (shapely_shim.py)
  2. Python-2 semantics: every `a / b` in the reference source is rewritten to floor-divide when
     both operands are ints (utils.py:113 `360/359 == 1`, environment_stage_1_nobonus.py:577)
  3. virtual clock: time.time() / time.sleep() / wait_for_message advance simulated time and the
     2-D simulator (the simulator is oracle/cn_oracle.c's, reached through ctypes)
  4. math.hypot -> libm hypot (what Python 2.7's math.hypot calls)
  5. the call pattern of start_td3_training.py:106-166 (reset, sleep 0.1, done=False, step...)
"""
import ast
import contextlib
import ctypes
import ctypes.util
import io
import math
import os
import sys
import types

import numpy as np
import yaml

REF_SRC = "/root/reference/turtlebot3_rl_sim/src"


# ------------------------------------------------------------------ message stubs
class _Obj(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Vector3(_Obj):
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = x, y, z

    def __repr__(self):
        return "x: %r\ny: %r\nz: %r" % (self.x, self.y, self.z)


class Point(Vector3):
    pass


class Quaternion(_Obj):
    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.x, self.y, self.z, self.w = x, y, z, w


class Pose(_Obj):
    def __init__(self):
        self.position = Point()
        self.orientation = Quaternion()


class Twist(_Obj):
    def __init__(self):
        self.linear = Vector3()
        self.angular = Vector3()


class PointStamped(_Obj):
    def __init__(self):
        self.point = Point()


class LaserScan(_Obj):
    def __init__(self, ranges=()):
        self.ranges = list(ranges)


class Odometry(_Obj):
    def __init__(self):
        self.pose = _Obj(pose=Pose())
        self.twist = _Obj(twist=Twist())


class Marker(_Obj):
    def __init__(self):
        self.header = _Obj(frame_id="", stamp=None)
        self.type = 0
        self.id = 0
        self.scale = Vector3()
        self.color = _Obj(r=0.0, g=0.0, b=0.0, a=0.0)
        self.pose = Pose()
        self.text = ""


class Empty(object):
    pass


def euler_from_quaternion(q):
    """tf.transformations.euler_from_quaternion for axes='sxyz' restricted to what the reference
    reads (roll, pitch, yaw of a quaternion [x, y, z, w]); standard ZYX formulas."""
    x, y, z, w = q
    roll = math.atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))
    sp = 2.0 * (w * y - z * x)
    sp = max(-1.0, min(1.0, sp))
    pitch = math.asin(sp)
    yaw = math.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return roll, pitch, yaw


# ------------------------------------------------------------------ py2 loader
def _py2div(a, b):
    if isinstance(a, (int, np.integer)) and not isinstance(a, bool) and isinstance(b, (int, np.integer)) \
            and not isinstance(b, bool):
        return a // b
    return a / b


def py2_round(x, ndigits=0):
    """Python 2.7's builtin round (floatobject.c `_Py_double_round`): the argument is taken as a C double (so np.float64 goes
    the same way as float -- Python 3 would dispatch to numpy's multiply / rint / divide), rounded CORRECTLY to `ndigits`
    decimals with an exact tie going away from zero, and returned as a float.  decimal does the exact arithmetic:
    Decimal(float) is exact and ROUND_HALF_UP is "ties away from zero"."""
    import decimal
    x = float(x)
    if x != x or x in (float("inf"), float("-inf")):
        return x
    q = decimal.Decimal(1).scaleb(-int(ndigits))
    with decimal.localcontext() as ctx:
        ctx.prec = 400
        return float(decimal.Decimal(x).quantize(q, rounding=decimal.ROUND_HALF_UP))


class _DivRewriter(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(
                ast.Call(func=ast.Name(id="_py2div", ctx=ast.Load()), args=[node.left, node.right], keywords=[]),
                node)
        return node


def _load_py2(name, path, extra_globals):
    with open(path, "r") as f:
        src = f.read()
    tree = ast.parse(src, filename=path)
    tree = _DivRewriter().visit(tree)
    ast.fix_missing_locations(tree)
    mod = types.ModuleType(name)
    mod.__file__ = path
    mod.__dict__["_py2div"] = _py2div
    mod.__dict__.update(extra_globals)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        code = compile(tree, path, "exec")
        sys.modules[name] = mod
        exec(code, mod.__dict__)
    return mod


# ------------------------------------------------------------------ the harness
class _MathProxy(types.ModuleType):
    """`math` as Python 2.7 had it: hypot is libm's (CPython >= 3.8 uses its own algorithm)."""

    def __init__(self):
        super().__init__("math")
        self.__dict__.update(math.__dict__)
        libm = ctypes.CDLL(ctypes.util.find_library("m"))
        libm.hypot.argtypes = [ctypes.c_double, ctypes.c_double]
        libm.hypot.restype = ctypes.c_double
        self.hypot = lambda a, b: libm.hypot(float(a), float(b))


class Harness(object):
    """Plays ROS + Gazebo for one reference `Env`.  `sim` is an oracle.Oracle with n_envs == 1
    whose simulator half (cno_hsim_*) stands in for Gazebo."""

    ENV_MODULE = "environment_stage_1_nobonus"

    def __init__(self, sim, params=None, quiet=True):
        self.sim = sim
        c = sim.cfg
        self.latency_ms = c.scan_latency_ms
        self.clock = 0.0
        self.cmd = (0.0, 0.0)
        self.quiet = quiet
        self.odom_cb = None
        self.trace = []          # captured get_state inputs
        self._last_sleep_pos = None
        self.params = {
            "/turtlebot3/starting_pose/x": c.start_x, "/turtlebot3/starting_pose/y": c.start_y,
            "/turtlebot3/starting_pose/z": 0.0,
            "/turtlebot3/desired_pose/x": c.goal_x, "/turtlebot3/desired_pose/y": c.goal_y,
            "/turtlebot3/desired_pose/z": 0.0,
            "/turtlebot3/linear_forward_speed": 0.5, "/turtlebot3/linear_turn_speed": 0.05,
            "/turtlebot3/angular_speed": 0.3, "/turtlebot3/scan_ranges": c.n_rays,
            "/turtlebot3/max_scan_range": c.max_scan_range, "/turtlebot3/min_scan_range": c.min_scan_range,
            "/turtlebot3/nsteps": c.max_steps,
        }
        if params:
            self.params.update(params)
        self._install()
        from . import shapely_shim
        shapely_shim.UNTYPED_EMPTY = bool(getattr(c, "geos_untyped_empty", 0))    # cn_config.geos_untyped_empty
        # cn_config.py2_round: the reference's platform is Python 2.7 (README.md:108-110); its round() differs from Python 3's
        # on exact ties and on np.float64 arguments.  A module-level `round` shadows the builtin inside the reference's modules.
        extra = {"round": py2_round} if getattr(c, "py2_round", 0) else {}
        self.utils = _load_py2("utils", os.path.join(REF_SRC, "utils.py"), dict(extra))
        self.envmod = _load_py2(self.ENV_MODULE, os.path.join(REF_SRC, self.ENV_MODULE + ".py"), dict(extra))
        for m in (self.utils, self.envmod):
            m.time = self.time_mod
            m.math = self.math_mod
        self._before_env()
        with self._silence():
            self.env = self.envmod.Env(action_dim=2, max_step=c.max_steps)
        self._after_env(c)
        self._push_odom()

    def _before_env(self):
        pass

    def _after_env(self, c):
        self.env.k_obstacle_count = c.k_obstacles  # ENV:55 is a source-edit switch
        self._wrap_get_state()

    # -- stubs ---------------------------------------------------------------------------
    def _install(self):
        h = self
        rospy = types.ModuleType("rospy")

        class Publisher(object):
            def __init__(self, topic, *a, **k):
                self.topic = topic

            def publish(self, msg):
                if self.topic == "cmd_vel":
                    h.cmd = (float(msg.linear.x), float(msg.angular.z))

        class Subscriber(object):
            def __init__(self, topic, typ, cb, *a, **k):
                if topic == "odom":
                    h.odom_cb = cb

        class ServiceProxy(object):
            def __init__(self, name, *a, **k):
                self.name = name

            def __call__(self, *a, **k):
                if self.name == "gazebo/reset_simulation":
                    h.sim.hsim_reset()
                    h.cmd = (0.0, 0.0)
                    h._push_odom()

        class ServiceException(Exception):
            pass

        def wait_for_message(topic, typ, timeout=None):
            h._advance(h.latency_ms)
            return LaserScan([float(r) for r in h.sim.hsim_scan()])

        rospy.Publisher = Publisher
        rospy.Subscriber = Subscriber
        rospy.ServiceProxy = ServiceProxy
        rospy.ServiceException = ServiceException
        rospy.wait_for_message = wait_for_message
        rospy.wait_for_service = lambda *a, **k: None
        rospy.on_shutdown = lambda *a, **k: None
        rospy.get_param = lambda name, default=None: h.params[name]
        rospy.loginfo = rospy.logwarn = rospy.logerr = lambda *a, **k: None
        rospy.Time = _Obj(now=lambda: h.clock)
        rospy.init_node = lambda *a, **k: None

        def mod(name, **attrs):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
            return m

        sys.modules["rospy"] = rospy
        mod("geometry_msgs"); mod("geometry_msgs.msg", Twist=Twist, Pose=Pose, Point=Point, PointStamped=PointStamped)
        mod("sensor_msgs"); mod("sensor_msgs.msg", LaserScan=LaserScan)
        mod("nav_msgs"); mod("nav_msgs.msg", Odometry=Odometry)
        mod("std_srvs"); mod("std_srvs.srv", Empty=Empty)
        mod("visualization_msgs"); mod("visualization_msgs.msg", Marker=Marker)
        mod("tf"); mod("tf.transformations", euler_from_quaternion=euler_from_quaternion)
        mod("rospkg", RosPack=lambda: _Obj(get_path=lambda name: "/tmp"))   # environment_stage_1_original.py:35,127
        from . import shapely_shim
        mod("shapely")
        mod("shapely.geometry", Point=shapely_shim.Point, LineString=shapely_shim.LineString,
            polygon=None)
        mod("shapely.geometry.polygon", Polygon=shapely_shim.Polygon)

        tm = types.ModuleType("time")
        tm.time = lambda: h.clock
        tm.sleep = lambda d: h._sleep(d)
        self.time_mod = tm
        self.math_mod = _MathProxy()

    def _silence(self):
        return contextlib.redirect_stdout(io.StringIO()) if self.quiet else contextlib.nullcontext()

    # -- simulator coupling ---------------------------------------------------------------
    def _advance(self, ms, seconds=None):
        self.clock += (ms / 1000.0) if seconds is None else seconds
        self.sim.hsim_advance(ms, self.cmd[0], self.cmd[1])
        self._push_odom()

    def _sleep(self, d):
        self._advance(int(round(d * 1000.0)), d)
        robot, _, _, _ = self.sim.sim_state()
        self._deque_pos = (float(robot[0]), float(robot[1]))  # what ENV:1208 is about to read

    def _push_odom(self):
        robot, _, _, _ = self.sim.sim_state()
        od = Odometry()
        od.pose.pose.position = Point(float(robot[0]), float(robot[1]), 0.0)
        yaw = float(robot[2])
        od.pose.pose.orientation = Quaternion(0.0, 0.0, math.sin(yaw / 2.0), math.cos(yaw / 2.0))
        if getattr(self.sim.cfg, "wheel_accel", 0.0) > 0.0:   # /odom reports the twist the wheels have, not the command (cn_config.wheel_accel)
            od.twist.twist.linear = Vector3(float(robot[3]), 0.0, 0.0)
            od.twist.twist.angular = Vector3(0.0, 0.0, float(robot[4]))
        else:
            od.twist.twist.linear = Vector3(self.cmd[0], 0.0, 0.0)
            od.twist.twist.angular = Vector3(0.0, 0.0, self.cmd[1])
        if self.odom_cb is not None:
            self.odom_cb(od)

    def _wrap_get_state(self):
        h = self
        orig = self.env.get_state

        def get_state(scan, step_counter=0, action=[0, 0]):
            e = h.env
            rec = dict(ranges=np.array(scan.ranges, dtype=np.float64), px=e.position.x, py=e.position.y,
                       v=e.linear_twist.x, w=e.angular_twist.z, now=h.clock, step_counter=int(step_counter))
            out = orig(scan, step_counter, action)
            rec["yaw"] = float(e.robot_yaw)
            h.trace.append(rec)
            return out

        self.env.get_state = get_state

    # -- TRAIN:106-166 call pattern ---------------------------------------------------------
    def reset(self):
        with self._silence():
            obs = self.env.reset()                    # TRAIN:113
            rec = self.trace[-1]
            rec.update(is_reset=1, deque_x=0.0, deque_y=0.0, end_timestep=0.0)
            self.time_mod.sleep(0.1)                  # TRAIN:114
            self.env.done = False                     # TRAIN:116
        return np.asarray(obs, dtype=np.float64)

    def step(self, action, step_counter):
        with self._silence():
            obs, reward, done = self.env.step([float(action[0]), float(action[1])], int(step_counter),
                                              mode="continuous")  # TRAIN:125
        rec = self.trace[-1]
        rec.update(is_reset=0, end_timestep=float(self.env.agent_vel_timestep))
        rec["deque_x"], rec["deque_y"] = self._deque_pos
        return np.asarray(obs, dtype=np.float64), float(reward), bool(done)

    def snapshot(self):
        e = self.env
        tr = list(e.tracked_obstacles.values())
        return dict(
            n_tracks=len(tr),
            track_pose=np.array([t[1] for t in tr], dtype=np.float64).reshape(-1, 2),
            track_dist=np.array([t[2] for t in tr], dtype=np.float64),
            track_speed=np.array([t[5] for t in tr], dtype=np.float64),
            track_vel=np.array([t[6] for t in tr], dtype=np.float64).reshape(-1, 2),
            track_dqlen=np.array([len(t[3]) for t in tr], dtype=np.int32),
            collision_prob=float(e.collision_prob) if e.collision_prob is not None else 0.0,
            ego_score=float(e.ego_score_collision_prob),
            wp=(float(e.waypoint_desired_point.x), float(e.waypoint_desired_point.y)),
            counters=(int(e.ego_safety_violation_count), int(e.social_safety_violation_count),
                      int(e.obstacle_present_step_counts)),
            bb=float(e.bounding_box_size) if e.bounding_box_size is not None else 0.0,
            status=(bool(e.episode_success), bool(e.episode_failure)),
        )


class HarnessOriginal(Harness):
    """The same harness around environment_stage_1_original.py (the 363-input layout of the SAC / DQN / Q-learning
    trainers: 359 rounded ranges + heading + distance + rounded position).  Differences handled here:
    the module imports rospkg (stubbed in _install), and get_state appends a trajectory row to a results CSV on every call
    (ORIG:286 utils.record_data) -- file output only, replaced by a no-op."""
    ENV_MODULE = "environment_stage_1_original"

    def _before_env(self):
        self.utils.record_data = lambda *a, **k: None

    def _after_env(self, c):
        h = self
        orig = self.env.get_state

        def get_state(scan, step_counter=0, action=[0, 0]):
            e = h.env
            rec = dict(ranges=np.array(scan.ranges, dtype=np.float64), px=e.position.x, py=e.position.y,
                       v=h.cmd[0], w=h.cmd[1], now=h.clock, step_counter=int(step_counter))
            out = orig(scan, step_counter, action)
            rec["yaw"] = float(e.robot_odometry[2])
            h.trace.append(rec)
            return out

        self.env.get_state = get_state

    def step(self, action, step_counter):
        with self._silence():
            obs, reward, done = self.env.step([float(action[0]), float(action[1])], int(step_counter),
                                              mode="continuous")  # SAC:116
        self.trace[-1].update(is_reset=0)
        return np.asarray(obs, dtype=np.float64), float(reward), bool(done)

    def reset(self):
        with self._silence():
            obs = self.env.reset()                    # SAC:105
            self.trace[-1].update(is_reset=1)
            self.time_mod.sleep(0.1)                  # SAC:106
            self.env.done = False                     # SAC:107
        return np.asarray(obs, dtype=np.float64)

    def snapshot(self):
        e = self.env
        return dict(status=(bool(e.episode_success), bool(e.episode_failure)),
                    prev=(float(e.previous_distance), float(e.previous_heading)))


class HarnessRealworld(Harness):
    """The same harness around environment_stage_1_nobonus_realworld.py (the 370-input physical-robot variant).
    One Python-2 semantic is supplied by hand: `self.collision_prob = None` (RW:80) is compared with `> 0.4` at RW:708 before it
    is ever assigned; Python 2 orders None below every number (the comparison is False), Python 3 raises -- so the attribute
    starts as -inf here, which compares the same way and is never read otherwise."""
    ENV_MODULE = "environment_stage_1_nobonus_realworld"

    def _after_env(self, c):
        self.env.collision_prob = float("-inf")
        self._wrap_get_state()

    def snapshot(self):
        e = self.env
        tr = list(e.tracked_obstacles.values())
        return dict(
            n_tracks=len(tr),
            track_pose=np.array([t[1] for t in tr], dtype=np.float64).reshape(-1, 2),
            track_dist=np.array([t[2] for t in tr], dtype=np.float64),
            track_speed=np.array([t[5] for t in tr], dtype=np.float64),
            track_vel=np.array([t[6] for t in tr], dtype=np.float64).reshape(-1, 2),
            collision_prob=float(e.collision_prob),
            counters=(int(e.ego_safety_violation_count), int(e.social_safety_violation_count)),
            bb=float(e.bounding_box_size),
            status=(bool(e.episode_success), bool(e.episode_failure)),
            prev=(float(e.previous_distance), float(e.previous_heading)),
        )
