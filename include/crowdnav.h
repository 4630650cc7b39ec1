/*
 * crowdnav.h -- C-ABI of libcrowdnav.so, the MI355X-native batched crowd-navigation environment.
 *
 * The reference has no FFI layer: its hot path is the Python object `Env`
 * (turtlebot3_rl_sim/src/environment_stage_1_nobonus.py:42) driven by
 * start_td3_training.py:106-166 and backed by Gazebo + a separate crowd node
 * (crowd_behaviors/simulate_crowd.py).  This header is the boundary a drop-in replacement
 * exports; each entry point cites the reference interface it replaces.
 * The Python binding a maintainer would add is shown in INTEGRATION.md and shipped in
 * drl-based-mapless-crowd-navigation-with-perceived-risk_amd/crowdnav/_abi.py.
 *
 * Conventions
 *   - plain pointers and sizes only; every array is caller-owned
 *   - "dev" pointers are device (HBM) pointers on the handle's GPU, "host" pointers are host memory
 *   - cn_reset / cn_step only enqueue work on `stream` (a hipStream_t; NULL = default stream)
 *     and return immediately; no hidden synchronisation
 *   - return value 0 = OK, negative = error; cn_last_error() gives a thread-local message;
 *     nothing throws across the boundary
 *   - there is NO CPU fallback: cn_create fails (CN_ERR_NO_DEVICE) when no HIP device exists
 */
#ifndef CROWDNAV_H
#define CROWDNAV_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CN_ABI_VERSION 7
#define CN_MAX_TRACKS 64      /* largest per-env capacity of the obstacle tracker (ENV:656-743): one lane per track */
#define CN_MAX_K 16

enum {
    CN_OK = 0,
    CN_ERR_ARG = -1,
    CN_ERR_CONFIG = -2,
    CN_ERR_NO_DEVICE = -3,
    CN_ERR_HIP = -4,
    CN_ERR_SIZE = -5
};

/* per-env status bits (cn_get_counters column 6) */
enum { CN_ST_TRACK_OVERFLOW = 1, CN_ST_TTC_ZERO = 2, CN_ST_DT_ZERO = 4, CN_ST_CONF_OVERFLOW = 8 };

/* Every field of Env.__init__'s rosparam reads (ENV:71-91), the robot/lidar constants of the
 * URDF/XACRO and world files, and the crowd node's constants.  SURVEY.md appendix B cites each. */
enum { CN_LAYOUT_RISK = 0, CN_LAYOUT_ORIGINAL = 1, CN_LAYOUT_REALWORLD = 2 };
/* where the perceived-risk features (rows A21-A24) take their obstacles from */
enum { CN_RISK_LIDAR_TRACKER = 0, CN_RISK_GT = 1 };

typedef struct cn_config {
    int32_t n_envs;          /* N environments in this handle (this GPU's shard) */
    int32_t n_peds;          /* P pedestrians (obstacle cylinders) per env */
    int32_t n_rays;          /* R lidar samples (XACRO:157 -> 360); the observation uses R-1 */
    int32_t k_obstacles;     /* K tracked obstacles in the observation (ENV:55 -> 8) */
    int32_t max_steps;       /* Env(max_step=...) (ENV:43,91) */
    int32_t ped_mode;        /* 0: U(-vmax,vmax) velocity per cycle (CROWD:98-126); 1: constant preset table;
                              * 2: social-force pedestrians (BASELINE north_star; Helbing-Molnar: goal attraction + pedestrian,
                              *    wall and robot repulsion, 10 ms ticks; sf_* below; no reference source -- CROWD:98-126 is a
                              *    random-velocity walker -- so it is pinned by analytic known-answer cases only) */
    int32_t dt_ms;           /* time.sleep(0.15) in Env.step (ENV:1201) -> 150 */
    int32_t scan_latency_ms; /* wait_for_message('scan') (ENV:1218,1238) -> 10 */
    int32_t settle_ms;       /* trainer's time.sleep(0.1) after reset (TRAIN:114) -> 100 */
    int32_t ped_cycle_ms;    /* crowd node cycle: 0.1 s x number of obstacles (CROWD:128-144) */
    int32_t ped_stagger_ms;  /* 0.1 s between consecutive obstacles' updates (CROWD:144) -> 100 */
    int32_t track_capacity;  /* tracker slots per env: 0 = auto (32 for <= 40 pedestrians -- <= 32 with risk_mode gt --, else 64), or 32 / 64 */
    int32_t obs_layout;      /* CN_LAYOUT_RISK (0): environment_stage_1_nobonus.py, obs = R-1 + 7 + 4K (TD3 / DDPG trainers);
                              * CN_LAYOUT_ORIGINAL (1): environment_stage_1_original.py:278-402, obs = R-1 + 4 =
                              * rounded ranges + heading + distance + rounded (x, y) (SAC / DQN / Q-learning trainers);
                              * CN_LAYOUT_REALWORLD (2): environment_stage_1_nobonus_realworld.py:208-749, obs = R-1 + 11 = unrounded
                              * ranges + heading + distance + rounded (x, y) + the constant yaw 3.14 + rounded twist features +
                              * pose and velocity of the one obstacle with the highest collision probability (the physical-robot
                              * script: set dt_ms = 50, RW:880-883; normally driven through cn_observe_external) */
    int32_t geos_untyped_empty; /* shapely/GEOS version switch for UTL:279,306 `str(i) != 'LINESTRING EMPTY'`:
                              * 0: GEOS >= 3.9 typed empties (a miss prints 'LINESTRING EMPTY' and is skipped);
                              * 1: GEOS <= 3.8 (the reference's Python-2.7 / shapely <= 1.7 platform): a miss prints
                              *    'GEOMETRYCOLLECTION EMPTY', the comparison is true, `.geoms` of the empty result is
                              *    empty and the [0] raises -> get_collision_point returns None on the FIRST candidate
                              *    that misses, and get_local_goal_waypoints takes its except branch */
    int32_t ped_contact;     /* 0: pedestrians pass through each other and the robot (round-1 simulator);
                              * 1: frictionless rigid contact, disc-disc and disc-robot (WORLD:86-145 mu = 0 cylinders) */
    int32_t risk_mode;       /* CN_RISK_LIDAR_TRACKER (0): the reference's lidar segmentation + tracker (ENV:329-760);
                              * CN_RISK_GT (1): A21-A24 fed with the simulator's own pedestrians (nearest surface point,
                              *    true velocity) within lidar range and line of sight; indices = pedestrian ids */
    int32_t py2_round;       /* 0: Python-3 round() (ties-to-even on the exact binary value; round(np.float64, n) is numpy's
                              *    multiply / rint / divide) -- what the goldens were recorded under;
                              * 1: Python-2.7 round(), the reference's platform (README.md:108-110): exact ties go AWAY from zero,
                              *    and round(np.float64, n) is the builtin's correctly rounded decimal too (ENV:255, ORIG:280, RW:209).
                              *    Every round() site: ENV:1208, 329-346, 1025-1042, UTL:122-123, 460 ... (np.around at ENV:1042 stays numpy) */
    int32_t sf_tick_ms;      /* ped_mode 2: physics tick of the social-force integrator, ms; 0 -> 10 (the contact model's tick).  The model is
                              * an explicit scheme, so the tick is part of its definition: 10 ms resolves a 0.2 m/s crowd to 2 mm per tick,
                              * 50 ms (the usual choice for social-force crowds, tau = 0.5 s) costs a fifth */
    int32_t scan_f32;        /* 0: the simulated lidar hands Env.get_state float64 ranges (rounds 1-3);
                              * 1: every range is rounded to float32 first -- what sensor_msgs/LaserScan.ranges (float32[]) carries from
                              *    gazebo_ros_laser (XACRO:172-175) to ENV:1218's wait_for_message.  Simulator side only: the reference code
                              *    behind it is unchanged, but its exact-equality tests (`round(scan, 3) == 0.6`, ENV:324-346) then see
                              *    float32-representable inputs.  Externally supplied scans (cn_observe_external) are taken as they are */
    int32_t waypoint_reward; /* ENV:1116 `waypoint_reward = 200` -> 200.  The published training log (results/td3/revamped/
                              * new_tracking_cp_gcp_nobonus_corrected_3/td3_training.csv, 3021 episodes) never contains it -- its largest
                              * episode return is 173 < 200 -- while the committed reward pays it whenever the robot comes within
                              * goal_eps of a way-point 0.3 m ahead, i.e. on most steps of a diagonal run:
                              * 0 reproduces the reward the log was recorded under ("nobonus") */
    int64_t env_index_base;  /* global index of env 0: RNG streams are keyed by global index */
    uint64_t seed;
    double room_half;        /* WORLD:926-1108 -> 1.40 */
    double ped_radius;       /* WORLD:109 -> 0.0505 */
    double ped_vmax;         /* CROWD:101-102 -> 0.2 */
    double robot_clearance;  /* robot centre kept this far from the walls -> 0.09 */
    double lidar_min;        /* XACRO:164 -> 0.08 */
    double lidar_max;        /* XACRO:165 -> 0.60 */
    double lidar_span;       /* XACRO:159-160 -> 6.28 rad */
    double lidar_offset_x;   /* URDF:134-138 -> -0.032 */
    double max_scan_range;   /* turtlebot3_world.yaml:7 -> 0.6 */
    double min_scan_range;   /* turtlebot3_world.yaml:8 -> 0.12 (0.0 for evaluation); must be < max_scan_range (ENV:581 divides by the difference) */
    double goal_x, goal_y;   /* desired_pose (turtlebot3_world.yaml:10-13) */
    double start_x, start_y; /* starting_pose: the heading offset of ENV:223-224 only */
    double spawn_x, spawn_y, spawn_yaw; /* launch-file spawn pose (1.0, -1.0, 3.14) */
    double waypoint_radius;  /* ENV:250 -> 0.3 */
    double goal_eps;         /* ENV:1285,1303 -> 0.20 */
    /* ped_mode 2 (social force).  Per pedestrian i: desired speed v0_i = ped_vmax (0.5 + 0.5 u_i), a goal point drawn uniformly
     * in the room (re-drawn when within sf_goal_eps of it), and per 10 ms tick
     *   a = (v0 e_goal - v) / sf_tau + sum_j q(sf_A exp((2 r - d_ij) / sf_B) n_ij)       (other pedestrians; q rounds each component to
     *                                                                                    the nearest multiple of 2^-36 m/s^2, which makes
     *                                                                                    the sum exact and independent of its order -- the
     *                                                                                    kernels evaluate every unordered pair once and
     *                                                                                    scatter +-; needs n_peds sf_A e^(2r/sf_B) < 2^15)
     *     + sum_walls sf_wall_A exp((r - d_w) / sf_wall_B) n_w                            (-x, +x, -y, +y)
     *     + sf_A exp((r + robot_clearance - d_ir) / sf_B) n_ir                            (the robot)
     *   v <- v + a h, |v| capped at 1.3 v0;  x <- clamp(x + v h) into the room   (semi-implicit Euler, h = sf_tick_ms) */
    double sf_tau;           /* relaxation time -> 0.5 s (Helbing & Molnar 1995) */
    double sf_A;             /* pedestrian / robot repulsion strength, m/s^2 -> 0.8 (2.1 at full walking speed, scaled to 0.2 m/s crowds) */
    double sf_B;             /* ... and range, m -> 0.10 */
    double sf_wall_A;        /* wall repulsion strength, m/s^2 -> 1.0 */
    double sf_wall_B;        /* ... and range, m -> 0.05 */
    double sf_goal_eps;      /* a goal counts as reached within this distance -> 0.10 */
    /* Wheel dynamics of the diff-drive plugin (XACRO:57-72: libgazebo_ros_diff_drive.so, updateRate 100, wheelSeparation 0.160,
     * wheelAcceleration 1, wheelTorque 10).  wheel_accel = 0: kinematic robot, the commanded twist is the twist (rounds 1-3).
     * wheel_accel > 0 (XACRO:70 -> 1.0 m/s^2): the plugin's wheel-speed ramp, restated from gazebo_ros_pkgs'
     * gazebo_ros_diff_drive.cpp (UpdateChild / getWheelVelocities / UpdateOdometryEncoder; third-party, not vendored by the
     * reference, so this is its published algorithm and not a pinned parity): on plugin ticks of 10 ms
     *   target wheel speeds  tl = v_cmd - w_cmd * sep / 2,  tr = v_cmd + w_cmd * sep / 2
     *   if |tl - cl| < 0.01 or |tr - cr| < 0.01:  cl = tl, cr = tr          (either wheel within tolerance releases both)
     *   else  cl += clamp(tl - cl, -a h, +a h),  cr += clamp(tr - cr, -a h, +a h)
     *   the tick's twist v = (cl + cr) / 2, w = (cr - cl) / sep moves the robot by the mid-point rule; /odom reports that twist.
     * A commanded 0 <-> 0.22 m/s then takes 0.22 s (1.5 control periods), a full turn command w = 2 rad/s 0.16 s.
     * Requires obs_layout 0 and the plain simulator (ped_contact 0, ped_mode 0 / 1). */
    double wheel_accel;      /* m/s^2 at the wheel rim; 0 = off */
    double wheel_separation; /* XACRO:68 -> 0.160 */
} cn_config;

typedef struct cn_env_s* cn_handle;

/* Replaces Env.step(action, step_counter, mode="continuous") -> (state, reward, done)
 * (ENV:1164-1225) for N envs at once. */
typedef struct cn_step_io {
    const float* action;         /* dev [N,2]  (v, w), already clipped by the caller (TD3:214-215) */
    const int32_t* step_counter; /* dev [N] 1-based (TRAIN:125) or NULL = per-env internal counter */
    float* obs;                  /* dev [N, D], D = cn_obs_dim(): 366+4K (obs_layout 0) or R-1+4 (obs_layout 1);
                                  * with auto_reset: first obs of the new episode where done */
    float* final_obs;            /* dev [N, 366+4K] or NULL: the observation Env.step returned (terminal if done) */
    double* obs_f64;             /* dev [N, 366+4K] or NULL: `obs` in float64 (the reference's dtype) */
    float* reward;               /* dev [N] */
    uint8_t* done;               /* dev [N] */
    int32_t* topk_idx;           /* dev [N,K] or NULL: tracker slot of each feature row, -1 = padding */
    int32_t auto_reset;          /* 0: none.  1: finished envs run Env.reset() (+TRAIN:114-116) inside the same call.
                                  * 2: "next-step" reset -- a finished env spends the NEXT call on Env.reset() (its
                                  *    action is ignored, reward 0, done 0, obs = first observation of the new episode) */
    int32_t reserved;
} cn_step_io;

/* Env.get_state + Env.compute_reward on EXTERNALLY supplied sensor data (ENV:245-1162): what the reference's
 * `Env` does when Gazebo -- or the physical robot of environment_stage_1_nobonus_realworld.py -- delivers the
 * /scan and /odom messages.  The library's own simulator is bypassed; tracker / waypoint / deque state is the
 * handle's.  This is also how the golden runs recorded from the reference are replayed through the kernel. */
typedef struct cn_external_io {
    const double* ranges;        /* dev [N,R] LaserScan.ranges as delivered: finite values >= 0, +inf = no return, NaN (ENV:1218,1238); a negative
                                  * range is not a LaserScan value and is outside the domain (clip before the call) */
    const double* odom;          /* dev [N,10]: position x, y, yaw, linear_twist.x, angular_twist.z (ENV:239-243),
                                  *   time.time() inside get_state, position x, y at the end of time.sleep (ENV:1208),
                                  *   end_timestep (ENV:1202), reserved */
    const int32_t* step_counter; /* dev [N] (ignored for the reset flow) */
    float* obs;                  /* dev [N, 366+4K] */
    double* obs_f64;             /* dev [N, 366+4K] or NULL */
    float* reward;               /* dev [N] */
    uint8_t* done;               /* dev [N] */
    int32_t* topk_idx;           /* dev [N,K] or NULL */
    int32_t is_reset;            /* 1: Env.reset() flow (ENV:1243-1262 + TRAIN:116), 0: Env.step() flow (ENV:1208-1223) */
    int32_t phase;               /* 0: the whole flow selected by is_reset.  Otherwise a mask of the pieces of Env.step, so that
                                  * get_state and compute_reward can be called separately as ENV:1222-1223 does:
                                  *   CN_PHASE_PRE          ENV:1208-1209 agent_pose_deque.append + agent_vel_timestep (odom[6..8])
                                  *   CN_PHASE_GET_STATE    Env.get_state(scan, step_counter, action) -> obs, done (ENV:245-1044)
                                  *   CN_PHASE_REWARD       Env.compute_reward(state, step_counter, done) -> reward, done
                                  *                         (ENV:1046-1162): reads state[R-1], state[R] from obs_f64 (or obs)
                                  *                         and `done` as INPUTS, position from odom[0..1] */
} cn_external_io;
enum { CN_PHASE_ALL = 0, CN_PHASE_PRE = 1, CN_PHASE_GET_STATE = 2, CN_PHASE_REWARD = 4 };

int cn_abi_version(void);
const char* cn_last_error(void);

/* Env.__init__ (ENV:43-168).  device = HIP device ordinal. */
int cn_create(const cn_config* cfg, int device, cn_handle* out);
void cn_destroy(cn_handle h);
int cn_obs_dim(cn_handle h);                       /* 366 + 4K (ENV:1038-1039) */
int cn_config_of(cn_handle h, cn_config* out);

/* World description (WORLD initial poses / scripted-crowd velocity tables), host arrays [N,P,2]. */
int cn_set_ped_init(cn_handle h, const double* xy_host);
int cn_get_ped_init(cn_handle h, double* xy_host);
int cn_set_ped_preset_vel(cn_handle h, const double* vxy_host);

/* Env.reset() (ENV:1227-1263) followed by the trainer's sleep(0.1) and `env.done = False`
 * (TRAIN:114-116).  mask: dev [N] or NULL (= all). obs_f64 may be NULL. */
int cn_reset(cn_handle h, const uint8_t* mask, float* obs, double* obs_f64, void* stream);
int cn_step(cn_handle h, const cn_step_io* io, void* stream);
/* Issue arbitration between the environments that share a SIMD (one environment = one wavefront).
 * The hardware serves a SIMD's OLDEST wavefront first; when the environments on a SIMD start together (a launch that fills
 * the device by itself) the launch then lasts as long as the wavefront that was starved.  CN_ARB_FAIR runs cn_step's kernel
 * with falling s_setprio levels -- whoever is behind gets the issue slots, all finish together: +8 % for one launch of 4096
 * environments per step -- and costs ~5 % when launches of several handles OVERLAP on the device (stream groups), where
 * oldest-first is the better pipeline.  CN_ARB_AUTO (the default): fair when this handle's launch alone puts at least two
 * wavefronts on every SIMD of the device (n_envs >= 8 x compute units) and it is stepped by cn_step or by a cn_step_multi
 * that names no other handle; oldest-first inside a cn_step_multi over several handles.  Changes when instructions issue, never a result.  Only the default configuration's kernel
 * (obs_layout 0, lidar_tracker, no contact / social force, reset on the next step or none) has a fair variant: elsewhere
 * the setting is accepted and ignored.  cn_get_arbitration returns what cn_step would use: CN_ARB_OLDEST_FIRST or CN_ARB_FAIR. */
#define CN_ARB_AUTO 0
#define CN_ARB_OLDEST_FIRST 1
#define CN_ARB_FAIR 2
int cn_set_arbitration(cn_handle h, int mode);
int cn_get_arbitration(cn_handle h);
/* Stream groups (ABI 6): how many environments the caller keeps in flight on this device TOGETHER with this handle's -- the sum over
 * the handles whose launches overlap (crowdnav.env.VecEnvGroups sets it for every group).  0 (the default) = this handle alone.
 * The 360-ray step kernels run four environments per workgroup while that total is resident at once (<= 16 wavefronts per CU) and
 * one per workgroup beyond; results are identical either way. */
int cn_set_group_envs(cn_handle h, int64_t total_envs);
/* Name of the device kernel a call on this handle launches right now (diagnostics: bench.py and the profile summaries key the
 * PMC counters of a run by it).  what: 0 = cn_step with auto_reset 0 / 2 and cn_reset, 1 = cn_step with auto_reset 1 (same-call
 * reset), 2 = cn_step_sequence, 3 = cn_observe_external, 4 = cn_step inside a cn_step_multi over several handles.  Handles whose
 * shape is the headline one (360 rays, 20 pedestrians, K = 8, default tracker slots) get kernels compiled for exactly that
 * shape (`..._s360`: the LDS map, word counts and loop bounds are constants there); results are identical.  NULL on error. */
const char* cn_kernel_name(cn_handle h, int what);
/* Diagnostics (bench.py's sustained leg): enqueues a one-thread kernel on `stream` that lives for `span_us` microseconds of the
 * constant 100 MHz counter (s_memrealtime) and stores in out_dev[0] the shader-clock cycles (s_memtime) and in out_dev[1] the 100 MHz
 * ticks that passed meanwhile: out[0] / (out[1] / 100) is the clock in MHz the chip ran at during that interval under whatever load
 * the other streams put on it.  (One wave, one die: the two counters are per XCD and readings of different kernels do not subtract.)
 * No reference counterpart. */
int cn_device_clock(int64_t* out_dev, int span_us, int device, void* stream);
/* n calls of cn_step in one crossing of the boundary: handle i steps with ios[i] on streams[i] (env batches run as
 * independent stream groups, DESIGN.md section 6: the launches are the same, the host thread pays the foreign-call
 * overhead once per step instead of once per group).  Stops at the first error and returns it. */
int cn_step_multi(int n, const cn_handle* handles, const cn_step_io* ios, void* const* streams);
int cn_observe_external(cn_handle h, const cn_external_io* io, void* stream);

/* The actor's output stage as one launch (no handle needed): action = clip(heads(logits) + N(0, sigma)).
 * Replaces td3.py:103-104 (sigmoid*max_v, tanh*max_w), td3.py:67-78,209-211 (Gaussian exploration) and
 * td3.py:214-215 (clip).  logits, action: dev [n,2] float32; noise is keyed by (seed, counter, row). */
int cn_policy_tail(const float* logits, float* action, int n, float max_v, float max_w, float sigma,
                   uint64_t seed, uint64_t counter, int device, void* stream);   /* device: HIP ordinal, -1 = current */

/* The whole TD3 actor as one launch: Actor.forward (td3.py:96-106: Linear(obs_dim,256)-ReLU-Linear(256,256)-ReLU-
 * Linear(256,2), sigmoid*max_v / tanh*max_w heads) + Agent.act's exploration noise and clip (td3.py:209-215), in
 * float32 on the f32 matrix cores.  Weights are caller-owned device arrays:
 *   w1p = cn_actor_pack_weights(linear1.weight^T [obs_dim_padded][256], zero rows from obs_dim up to a multiple of 32),
 *   w2p = cn_actor_pack_weights(linear2.weight^T [256][256]), w3 [2][256] = linear3.weight, biases as in PyTorch.
 * obs: dev [n, obs_dim] float32; action: dev [n,2].
 * cn_actor_pack_weights reorders a K-major [k_rows][256] float32 matrix (k_rows a multiple of 32) into the order the kernel's
 * wavefronts consume it -- packed[((((b*8 + w)*4 + q)*64 + lane)*4 + j] = wt[32 b + 4 (2 q + (j >> 1)) + (lane >> 4)][32 w +
 * 2 (lane & 15) + (j & 1)] -- so every lane streams 16-byte loads out of L2 and a wavefront's block is 4 KB contiguous
 * (k_rows * 256 floats, same size as the input; in and out must not alias).  Call it once per weight update. */
typedef struct cn_actor_weights {
    const float* w1p; const float* b1; const float* w2p; const float* b2; const float* w3; const float* b3;
    int32_t obs_dim, obs_dim_padded, hidden, reserved;
} cn_actor_weights;
int cn_actor_pack_weights(const float* wt_dev, int k_rows, float* packed_dev, int device, void* stream);
int cn_actor_forward(const cn_actor_weights* w, const float* obs, float* action, int n, float max_v, float max_w,
                     float sigma, uint64_t seed, uint64_t counter, int device, void* stream);

/* The TD3 update -- Agent.learn (td3.py:225-285) with the hyper-parameters of start_td3_training.py:62-72 -- as a short chain
 * of launches on the caller's stream (crowdnav_td3.hip): the forward and backward GEMMs of the six 3-layer networks on the f32
 * matrix cores, weight gradients folded into the Adam step (a gradient never exists in memory), the TD target / MSE gradient /
 * heads evaluated inside those GEMMs, replay sampling and target-policy noise drawn on the device.  7 launches for the critic
 * step, 5 more when the actor and the targets move (every policy_delay-th update); through PyTorch the same update is ~150
 * kernels.
 * The parameters stay the caller's: device pointers to the nn.Linear storages (weight [out][in] row-major float32, bias
 * [out]) of actor / critics and their targets, stepped in place.  Adam's moments and step counters are the handle's (zero at
 * cn_td3_create, like a fresh torch.optim.Adam).  Arithmetic: float32 throughout like the reference; same formulas as
 * torch.optim.Adam (no weight decay, no amsgrad) and F.mse_loss; results agree with the PyTorch update up to summation order. */
typedef struct cn_td3_mlp { float *w1, *b1, *w2, *b2, *w3, *b3; } cn_td3_mlp;   /* Linear(in, H) - ReLU - Linear(H, H) - ReLU - Linear(H, out) */
typedef struct cn_td3_config {
    int32_t obs_dim;         /* actor input width (TRAIN:88 -> 398); the critics take obs_dim + 2 (TD3:114) */
    int32_t hidden;          /* TRAIN:65 -> 256 */
    int32_t batch;           /* TRAIN:62 -> 128 */
    int32_t policy_delay;    /* TRAIN:72 -> 2 (the caller passes do_actor itself; kept for the record) */
    float gamma, tau;        /* td3.yaml -> 0.99, 0.005 */
    float lr_actor, lr_critic, beta1, beta2, eps;   /* 3e-4, 3e-4, torch.optim.Adam's 0.9, 0.999, 1e-8 */
    float noise_std, noise_clip;                    /* TRAIN:70-71 -> 0.2, 0.5 (target-policy smoothing, TD3:240-243) */
    float max_v, max_w;                             /* TRAIN:67-68 -> 0.22, 2.0 (the actor's heads, TD3:103-104) */
    float reserved;
    cn_td3_mlp actor, actor_t, q1, q1_t, q2, q2_t;
    /* the replay ring on the device (rows obs_dim / 2 / 1 / obs_dim / 1 floats wide; done as 0 / 1) and its live size (an int64
     * on the device, read at update time: indices are drawn uniformly below it).  May all be NULL when every update passes
     * an explicit batch. */
    const float *replay_s, *replay_a, *replay_r, *replay_s2, *replay_d;
    const int64_t* replay_size_dev;
    uint64_t seed;           /* keys the replay indices and the target-policy noise with the handle's update counter */
} cn_td3_config;
typedef struct cn_td3_batch {   /* an explicit batch instead of a replay sample (parity tests): [B, obs_dim], [B, 2], [B], [B, obs_dim], [B] */
    const float *s, *a, *r, *s2, *d;
    const float* target_noise;   /* [B, 2] unit-variance noise BEFORE the scale and clip (TD3:240-242), or NULL = drawn on the device */
} cn_td3_batch;
typedef struct cn_td3_s* cn_td3_handle;
int cn_td3_create(const cn_td3_config* cfg, int device, cn_td3_handle* out);
void cn_td3_destroy(cn_td3_handle h);
/* One update.  do_actor != 0: also the actor step and the three soft updates (TD3:264-285).  batch NULL = sample the replay.
 * Enqueues only (capturable into a hipGraph). */
int cn_td3_update(cn_td3_handle h, int do_actor, const cn_td3_batch* batch, void* stream);
const float* cn_td3_loss_dev(cn_td3_handle h);      /* device pointer: the first critic's MSE loss of the last update */
const char* cn_td3_last_error(void);

/* The collection loop's bookkeeping between Env.step and Agent.learn (start_td3_training.py:129-149) for a batch of environments,
 * without a host read: ReplayBuffer.add (td3.py:24-31) into a ring on the device, and the per-episode record TRAIN:139-149 prints
 * and utils.record_data writes.  (crowdnav.td3.DeviceReplay and crowdnav.train.DeviceEpisodeLog do the same through ~35 PyTorch
 * kernels per launch; next to a 0.065 ms update that is a seventh of a training launch.)  Both enqueue only.
 *
 * cn_replay_write: rows i < n with keep[i] != 0 (keep NULL = all) go to consecutive ring slots (*pos_dev + rank) mod capacity in
 * row order; then *pos_dev advances by their number and *size_dev grows up to capacity.  Rows not kept (an environment's reset
 * launch under the next-step reset convention) are not written.  Arrays: s, s2 [capacity][obs_dim], a [capacity][2], r, d
 * [capacity]; done / keep are bytes; n <= capacity.  slot_scratch: n int32 of device scratch. */
typedef struct cn_replay_ring {
    float *s, *a, *r, *s2, *d;
    int64_t capacity;
    int64_t* pos_dev;        /* next write position */
    int64_t* size_dev;       /* fill level (what cn_td3_config.replay_size_dev points at) */
    int32_t obs_dim, reserved;
} cn_replay_ring;
int cn_replay_write(const cn_replay_ring* ring, const float* s, const float* a, const float* r, const float* s2,
                    const uint8_t* done, const uint8_t* keep, int n, int32_t* slot_scratch, int device, void* stream);
/* cn_episode_log_add: one 8-float row per environment with done[i] != 0, appended at *n_dev in row order (rows past max_rows are
 * dropped, *n_dev keeps counting): {success, failure, return, steps, ego violations, social violations, obstacle-present steps,
 * launch} from the counter columns 4, 5, 13, 10, 11, 12 (cn_get_counters) and last_return (cn_get_returns); and the running
 * totals tot_dev[5] += {episodes, successes, sum of returns, sum of steps, number of rows with transitions[i] != 0}. */
typedef struct cn_episode_log {
    float* rows;             /* [max_rows][8] */
    int64_t max_rows;
    int64_t* n_dev;
    double* tot_dev;         /* [5] */
} cn_episode_log;
int cn_episode_log_add(const cn_episode_log* log, const uint8_t* done, const int32_t* counters, int counter_cols,
                       const float* last_return, const uint8_t* transitions, float launch, int n, int device, void* stream);

/* n_steps calls of cn_step (auto_reset = 2, the next-step reset convention) with OPEN-LOOP actions -- scripted or recorded
 * actions, action repeat, the uniform-random warm-up phase of an off-policy learner -- as ONE launch: a wavefront keeps its
 * environment for the whole launch and walks its steps at its own pace (no launch boundary and no device-wide join between
 * steps).  Results are bit-identical to n_steps cn_step calls.  Slot t of a buffer starts `stride` ELEMENTS after slot t - 1;
 * stride 0 = one slot (actions: the same [N, 2] held for every step; outputs: every step overwrites the slot).
 *   action  dev n_steps slots [N, 2] float32;  obs dev n_steps slots [N, D] float32 (slot t = the observation step t returns)
 *   reward / done / topk_idx (or NULL): n_steps slots [N] / [N] / [N, K]
 * Every configuration cn_create accepts has this form (round 5: social force, wheel ramp, both risk modes, the 720-ray shape; round 6:
 * the contact ticks and obs_layout 1 / 2): cn_kernel_name(h, 2) says which kernel runs it. */
typedef struct cn_sequence_io {
    const float* action;
    float* obs;
    float* reward;
    uint8_t* done;
    int32_t* topk_idx;
    int64_t action_stride, obs_stride, reward_stride, done_stride, topk_stride;
    int32_t n_steps, reserved;
} cn_sequence_io;
int cn_step_sequence(cn_handle h, const cn_sequence_io* io, void* stream);

/* n_steps control periods with the POLICY IN THE LOOP, as ONE launch: per period a_t = actor(o_t) + exploration noise, clipped
 * (cn_actor_forward's arithmetic, noise keyed by (seed, counter + t, env row)), then Env.step(a_t) (cn_step, auto_reset = 2) --
 * the collection loop of start_td3_training.py:104-168 (agent.act -> env.step) for every environment of the handle without a
 * launch or a device-wide join between periods: a workgroup = 16 environments joins only with itself, the actor runs on its CU's
 * matrix cores between two steps.  Results are bit-identical to n_steps x (cn_actor_forward, cn_step) with counters counter,
 * counter + 1, ...  The weights are those of `actor` for the whole launch (a learner that updates every period sees a policy
 * lag of at most n_steps periods; n_steps = 1 is the per-period loop as one launch instead of two).
 *   obs0     dev [N, D]: the observation the first action is computed from (cn_reset's / the previous call's last slot; may
 *            alias slot 0 of obs when obs_stride = 0)
 *   action   dev n_steps slots [N, 2] float32, OUTPUT: slot t = the action period t took
 *   obs / reward / done / topk_idx (or NULL): as cn_sequence_io (slot t = what period t's step returned)
 * Requirements: 8 environments of the handle's shape fitting one CU's LDS (16 per workgroup where they fit; obs_layout 2 always runs
 * 8: its observation needs more registers than a 16-wave workgroup leaves a wave), and an actor of cn_actor_pack_weights' layout for
 * this handle's observation width cn_obs_dim(h) -- 366 + 4 K, or 363 / 370 for obs_layout 1 / 2 at 360 rays -- with hidden = 256.
 * Every configuration cn_create accepts has this form as well (cn_kernel_name(h, 5)). */
typedef struct cn_policy_io {
    const float* obs0;
    float* action;
    float* obs;
    float* reward;
    uint8_t* done;
    int32_t* topk_idx;
    int64_t action_stride, obs_stride, reward_stride, done_stride, topk_stride;
    int32_t n_steps, reserved;
    float max_v, max_w, sigma, reserved_f;
    uint64_t seed, counter;
} cn_policy_io;
int cn_rollout_policy(cn_handle h, const cn_actor_weights* actor, const cn_policy_io* io, void* stream);

/* get_episode_status / get_*_safety_violation_status inputs (ENV:1265-1283).
 * out: dev [N,14] = ego_viol, social_viol, obstacle_present_steps, ep_steps, success, failure, status, n_tracks,
 *                  episodes finished since cn_create, reset pending (auto_reset == 2),
 *                  and the LAST FINISHED episode's ego_viol, social_viol, obstacle_present_steps, ep_steps as they stood
 *                  when Env.step returned done (what TRAIN:142-147 reads before the next reset zeroes them) */
#define CN_COUNTER_COLS 14
int cn_get_counters(cn_handle h, int32_t* out, void* stream);
/* return of the last finished episode and running return, dev [N] each (either may be NULL) */
int cn_get_returns(cn_handle h, float* last_return, float* running_return, void* stream);

/* Parity/debug view of one env (synchronises).  host buffers, any may be NULL:
 *   scalars[24] (layout: CN_SD_* below), robot_ped (5 + 4P doubles: x,y,yaw,v,w, ped xy, ped vxy),
 *   tracks [CN_MAX_TRACKS][12] one record per slot (CN_TF_*; the first track_capacity slots are used), ints[16] */
int cn_debug_env(cn_handle h, int env, double* scalars, double* robot_ped, double* tracks, int32_t* ints);

/* Sizing diagnostics (no handle, no GPU): LDS bytes one env of obs_layout 0 needs for n_rays / n_peds / k with max_conf
 * confirmed-object slots ((n_rays - 1) / 4 + 2 inside cn_create) and track_capacity slots; cn_create refuses > 160 KiB.
 * cn_near_separate: 1 when the near-pedestrian list gets its own LDS region at no cost in wavefronts per CU. */
size_t cn_lds_bytes(int n_rays, int n_peds, int k, int max_conf, int track_capacity);
int cn_near_separate(int n_rays, int n_peds, int k, int max_conf, int track_capacity);

/* Whole-state snapshot for deterministic replay and for stepping the CPU oracle from a GPU state.
 * Blob = cn_snapshot_header | sd [N][CN_SD_COUNT] f64 | si [N][CN_SI_COUNT] i32 | ped_p [N][P][2] f64 | ped_v [N][P][2] f64
 *        | trk [N][track_capacity][CN_TF_COUNT] f64 | ped_init [N][P][2] f64 | ped_preset [N][P][2] f64 | ped_aux [N][P][3] f64
 * (ped_aux: goal x, y and goal counter of ped_mode 2; zeros otherwise).  The header carries the ABI version and the FULL cn_config of the handle
 * that wrote it (env_index_base, seed, layout and mode switches included): cn_restore refuses a blob whose header does not
 * match the restoring handle field for field (CN_ERR_CONFIG, the message names the first field that differs) -- a state only
 * means something under the configuration that produced it.  crowdnav.env.VecEnv.save_snapshot / load_snapshot wrap the blob
 * in an .npz with the header spelled out; oracle/ (test infrastructure) loads that file into the CPU oracle. */
#define CN_SNAPSHOT_MAGIC 0x50414E534E43ull       /* "CNSNAP" little-endian */
typedef struct cn_snapshot_header {
    uint64_t magic;
    int32_t abi_version;       /* CN_ABI_VERSION of the writer */
    int32_t header_bytes;      /* sizeof(cn_snapshot_header) */
    int32_t sd_count, si_count, tf_count, track_capacity;   /* CN_SD_COUNT, CN_SI_COUNT, CN_TF_COUNT, resolved tracker slots */
    uint64_t total_bytes;      /* header + payload */
    cn_config config;
} cn_snapshot_header;
size_t cn_snapshot_size(cn_handle h);
int cn_snapshot(cn_handle h, void* host_buf, size_t size);
int cn_restore(cn_handle h, const void* host_buf, size_t size);

/* float64 scalar record per env */
enum {
    CN_SD_RX = 0, CN_SD_RY, CN_SD_RYAW, CN_SD_RV, CN_SD_RW, CN_SD_CLOCK, CN_SD_WPX, CN_SD_WPY,
    CN_SD_PREV_DIST, CN_SD_PREV_HEAD, CN_SD_DQ0X, CN_SD_DQ0Y, CN_SD_DQ1X, CN_SD_DQ1Y, CN_SD_TS,
    CN_SD_BB, CN_SD_EGO, CN_SD_CPROB, CN_SD_EP_RETURN, CN_SD_LAST_RETURN,
    CN_SD_LAST_EGO_VIOL, CN_SD_LAST_SOCIAL_VIOL, CN_SD_LAST_OBST_STEPS, CN_SD_LAST_EP_STEPS, CN_SD_COUNT = 24
};
/* int32 scalar record per env */
enum {
    CN_SI_DONE = 0, CN_SI_DQ_LEN, CN_SI_NTRACKS, CN_SI_EGO_VIOL, CN_SI_SOCIAL_VIOL, CN_SI_OBST_STEPS,
    CN_SI_SUCCESS, CN_SI_FAILURE, CN_SI_EP_STEP, CN_SI_STATUS, CN_SI_NCONF, CN_SI_NENTRIES,
    CN_SI_CROWD_LO, CN_SI_CROWD_HI, CN_SI_PENDING_RESET, CN_SI_EPISODES, CN_SI_COUNT = 16
};
/* track record fields (ENV:663-670: pose, range, deque(<=2), time stamp, speed, velocity) */
enum {
    CN_TF_PX = 0, CN_TF_PY, CN_TF_DIST, CN_TF_D0X, CN_TF_D0Y, CN_TF_D1X, CN_TF_D1Y, CN_TF_T, CN_TF_SPEED,
    CN_TF_VX, CN_TF_VY, CN_TF_DQLEN, CN_TF_COUNT = 12
};

#ifdef __cplusplus
}
#endif
#endif
