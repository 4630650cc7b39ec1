"""Typed configuration of the batched environment (replaces the rosparam namespace /turtlebot3/*
read at environment_stage_1_nobonus.py:71-90 and the XACRO / world constants).  SURVEY.md
appendix B cites every default."""
import ctypes as C
from dataclasses import asdict, dataclass


class CnConfig(C.Structure):
    """Mirror of `cn_config` in include/crowdnav.h."""
    _fields_ = [
        ("n_envs", C.c_int32), ("n_peds", C.c_int32), ("n_rays", C.c_int32), ("k_obstacles", C.c_int32),
        ("max_steps", C.c_int32), ("ped_mode", C.c_int32), ("dt_ms", C.c_int32), ("scan_latency_ms", C.c_int32),
        ("settle_ms", C.c_int32), ("ped_cycle_ms", C.c_int32), ("ped_stagger_ms", C.c_int32), ("track_capacity", C.c_int32),
        ("obs_layout", C.c_int32), ("geos_untyped_empty", C.c_int32), ("ped_contact", C.c_int32), ("risk_mode", C.c_int32),
        ("py2_round", C.c_int32), ("sf_tick_ms", C.c_int32), ("scan_f32", C.c_int32), ("waypoint_reward", C.c_int32),
        ("env_index_base", C.c_int64), ("seed", C.c_uint64),
        ("room_half", C.c_double), ("ped_radius", C.c_double), ("ped_vmax", C.c_double),
        ("robot_clearance", C.c_double), ("lidar_min", C.c_double), ("lidar_max", C.c_double),
        ("lidar_span", C.c_double), ("lidar_offset_x", C.c_double), ("max_scan_range", C.c_double),
        ("min_scan_range", C.c_double), ("goal_x", C.c_double), ("goal_y", C.c_double),
        ("start_x", C.c_double), ("start_y", C.c_double), ("spawn_x", C.c_double), ("spawn_y", C.c_double),
        ("spawn_yaw", C.c_double), ("waypoint_radius", C.c_double), ("goal_eps", C.c_double),
        ("sf_tau", C.c_double), ("sf_A", C.c_double), ("sf_B", C.c_double), ("sf_wall_A", C.c_double),
        ("sf_wall_B", C.c_double), ("sf_goal_eps", C.c_double), ("wheel_accel", C.c_double), ("wheel_separation", C.c_double),
    ]


@dataclass
class Config:
    n_envs: int = 1
    n_peds: int = 20
    n_rays: int = 360              # XACRO:157
    k_obstacles: int = 8           # ENV:55 / TRAIN:50
    max_steps: int = 1000          # configs/td3.yaml nsteps
    ped_mode: int = 0              # 0 random-velocity walkers (CROWD:98-126), 1 constant preset table, 2 social force (sf_* below)
    dt_ms: int = 150               # ENV:1201
    scan_latency_ms: int = 10      # virtual /scan wait
    settle_ms: int = 100           # TRAIN:114
    ped_cycle_ms: int = 0          # 0 -> 100 ms x n_peds (CROWD:128-144)
    ped_stagger_ms: int = 100      # CROWD:144
    track_capacity: int = 0        # tracker slots per env: 0 = auto (32 up to 40 pedestrians -- up to 32 with risk_mode 1 --, else 64)
    obs_layout: int = 0            # 0: environment_stage_1_nobonus (366 + 4K); 1: environment_stage_1_original (R-1 + 4);
                                   # 2: environment_stage_1_nobonus_realworld (R-1 + 11; use dt_ms=50, RW:880-883)
    geos_untyped_empty: int = 0    # 1: shapely <= 1.7 / GEOS <= 3.8 empty-result semantics at UTL:279,306 (the reference's platform)
    ped_contact: int = 0           # 1: frictionless rigid contact between pedestrians and with the robot (WORLD:86-145)
    risk_mode: int = 0             # 0: lidar segmentation + tracker (the reference); 1: "gt" -- simulator pedestrians
    py2_round: int = 0             # 1: Python-2.7 round() -- exact ties away from zero, round(np.float64) = the builtin (the reference's platform)
    sf_tick_ms: int = 0            # ped_mode 2: physics tick of the social-force integrator in ms (0 -> 10)
    scan_f32: int = 0              # 1: simulated ranges rounded to float32 before get_state (LaserScan.ranges is float32[])
    waypoint_reward: int = 200     # ENV:1116; 0 = the reward the published training log was recorded under (max return 173 < 200)
    env_index_base: int = 0
    seed: int = 1234
    room_half: float = 1.40        # WORLD:926-1108
    ped_radius: float = 0.0505     # WORLD:109
    ped_vmax: float = 0.2          # CROWD:101-102
    robot_clearance: float = 0.09
    lidar_min: float = 0.08        # XACRO:164
    lidar_max: float = 0.60        # XACRO:165
    lidar_span: float = 6.28       # XACRO:159-160
    lidar_offset_x: float = -0.032 # URDF:134-138
    max_scan_range: float = 0.6    # turtlebot3_world.yaml:7
    min_scan_range: float = 0.12   # turtlebot3_world.yaml:8
    goal_x: float = -1.0           # turtlebot3_world.yaml:10-13
    goal_y: float = 1.0
    start_x: float = 0.75          # turtlebot3_world.yaml:15-18 (heading offset only, ENV:223-224)
    start_y: float = -0.75
    spawn_x: float = 1.0           # put_robot_in_world_training.launch
    spawn_y: float = -1.0
    spawn_yaw: float = 3.14
    waypoint_radius: float = 0.3   # ENV:250
    goal_eps: float = 0.20         # ENV:1285
    sf_tau: float = 0.5            # ped_mode 2: relaxation time (Helbing & Molnar 1995)
    sf_A: float = 0.8              # pedestrian / robot repulsion strength, m/s^2
    sf_B: float = 0.10             # ... and range, m
    sf_wall_A: float = 1.0         # wall repulsion strength, m/s^2
    sf_wall_B: float = 0.05        # ... and range, m
    sf_goal_eps: float = 0.10      # goal reached within this distance -> next goal
    wheel_accel: float = 0.0       # XACRO:70 wheelAcceleration (m/s^2): 1.0 = the diff-drive plugin's wheel-speed ramp; 0 = kinematic robot
    wheel_separation: float = 0.160  # XACRO:68

    def resolved(self):
        d = asdict(self)
        if not d["ped_cycle_ms"]:
            d["ped_cycle_ms"] = max(100, 100 * d["n_peds"])
        return d

    def as_dict(self):
        return self.resolved()

    def to_c(self):
        d = self.resolved()
        return CnConfig(**d)

    @property
    def obs_dim(self):
        if self.obs_layout == 1:
            return (self.n_rays - 1) + 4                        # ORIG:315-320 (363 at 360 rays)
        if self.obs_layout == 2:
            return (self.n_rays - 1) + 11                       # RW:730-747 (370 at 360 rays)
        return (self.n_rays - 1) + 7 + 4 * self.k_obstacles   # ENV:1038-1039, TRAIN:88
