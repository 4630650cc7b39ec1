/*
 * cn_oracle.h -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, float64) of the
 * reference hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libcrowdnav.so) never does.
 *
 * What is restated (file:line are relative to /root/reference/turtlebot3_rl_sim/src):
 *   Env.reset / Env.step / Env.get_state / Env.compute_reward
 *                              environment_stage_1_nobonus.py:245-1263
 *   geometry / scan / CP helpers  utils.py:110-126, 227-345, 375-460
 *   pedestrian velocity process   crowd_behaviors/simulate_crowd.py:98-144
 *   obs_layout 1: Env.reset / step / get_state / compute_reward of
 *                              environment_stage_1_original.py:278-486 ("ORIG")
 * What has NO reference source (Gazebo/ODE + gazebo_ros plugins own it) and is
 * therefore DEFINED here and in DESIGN.md ("parity unpinned" for these rows):
 *   pedestrian position integration, diff-drive kinematics, lidar raycast.
 *
 * Pinning: the get_state/compute_reward restatement is checked bit-for-bit against
 * the reference's own Python executed under oracle/harness (see oracle/make_goldens.py
 * and tests/golden/).
 */
#ifndef CN_ORACLE_H
#define CN_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cno_config {
    int32_t n_envs;          /* N */
    int32_t n_peds;          /* P */
    int32_t n_rays;          /* R: lidar samples (360); observation uses R-1 */
    int32_t k_obstacles;     /* K (ENV:55) */
    int32_t max_steps;       /* ENV:91 */
    int32_t ped_mode;        /* 0 = random-velocity walkers (CROWD:98-126), 1 = constant preset velocities, 2 = social force */
    int32_t dt_ms;           /* control period, time.sleep(0.15) ENV:1201 -> 150 */
    int32_t scan_latency_ms; /* virtual /scan wait (harness: 10) */
    int32_t settle_ms;       /* trainer's time.sleep(0.1) after reset, TRAIN:114 -> 100 */
    int32_t ped_cycle_ms;    /* velocity resample period (CROWD: 0.1 s * n_obs) */
    int32_t ped_stagger_ms;  /* per-pedestrian offset (CROWD:144 sleep 0.1) -> 100 */
    int32_t reserved0;
    int32_t obs_layout;      /* 0: environment_stage_1_nobonus.py (366+4K); 1: environment_stage_1_original.py (R-1+4) */
    int32_t geos_untyped_empty; /* 1: GEOS <= 3.8 / shapely <= 1.7 untyped empties at UTL:279,306 (see include/crowdnav.h) */
    int32_t ped_contact;     /* 1: frictionless rigid contact pedestrian-pedestrian and pedestrian-robot */
    int32_t risk_mode;       /* 0: lidar tracker (reference); 1: gt (simulator pedestrians feed A21-A24) */
    int32_t py2_round;       /* 1: Python-2.7 round(): exact ties away from zero, round(np.float64, n) = the builtin (include/crowdnav.h) */
    int32_t sf_tick_ms;      /* ped_mode 2: physics tick in ms (0 -> 10) */
    int32_t scan_f32;        /* 1: simulated ranges rounded to float32 before get_state (sensor_msgs/LaserScan.ranges is float32[]) */
    int32_t waypoint_reward; /* ENV:1116 -> 200; 0 = the reward the published log was recorded under (include/crowdnav.h) */
    int64_t env_index_base;  /* global index of env 0 (multi-GPU sharding) */
    uint64_t seed;
    double room_half;        /* inner half extent of the square room (WORLD:926-1108 -> 1.40) */
    double ped_radius;       /* WORLD:109 -> 0.0505 */
    double ped_vmax;         /* CROWD:101 -> 0.2 */
    double robot_clearance;  /* robot centre kept this far from walls */
    double lidar_min;        /* XACRO:164 -> 0.08 */
    double lidar_max;        /* XACRO:165 -> 0.60 */
    double lidar_span;       /* XACRO:159-160 -> 6.28 */
    double lidar_offset_x;   /* URDF:134-138 -> -0.032 */
    double max_scan_range;   /* turtlebot3_world.yaml:7 -> 0.6 */
    double min_scan_range;   /* turtlebot3_world.yaml:8 -> 0.12 */
    double goal_x, goal_y;   /* desired_pose */
    double start_x, start_y; /* starting_pose: heading offset only (ENV:223-224) */
    double spawn_x, spawn_y, spawn_yaw; /* launch-file spawn pose */
    double waypoint_radius;  /* 0.3 (ENV:250) */
    double goal_eps;         /* 0.20 (ENV:1285,1303) */
    /* ped_mode 2: social-force pedestrians (defined in include/crowdnav.h; no reference source) */
    double sf_tau, sf_A, sf_B, sf_wall_A, sf_wall_B, sf_goal_eps;
    double wheel_accel;      /* XACRO:70 wheelAcceleration, m/s^2; 0 = kinematic robot (include/crowdnav.h states the model) */
    double wheel_separation; /* XACRO:68 -> 0.160 */
} cno_config;

typedef struct cno_sim cno_sim;

/* per-env debugging / parity outputs of one get_state call */
#define CNO_MAX_TRACKS 64
typedef struct cno_debug {
    int32_t n_confirmed;
    int32_t n_tracks;
    int32_t n_entries;          /* CP entries before top-K */
    int32_t status;             /* bit flags: 1 = track overflow, 2 = ttc==0, 4 = dt==0 */
    double bb;
    double collision_prob;
    double ego_score;
    double wpx, wpy;
    double track_pose[CNO_MAX_TRACKS][2];
    double track_dist[CNO_MAX_TRACKS];
    double track_speed[CNO_MAX_TRACKS];
    double track_vel[CNO_MAX_TRACKS][2];
    double track_t[CNO_MAX_TRACKS];
    int32_t track_dqlen[CNO_MAX_TRACKS];
    double entry_cp[CNO_MAX_TRACKS];    /* CP of every entry before the top-K cut (ENV:818-860), n_entries of them */
    double entry_ego[CNO_MAX_TRACKS];   /* ... and its ego score min(1, 0.15 / ttc) (negative for a negative ttc) */
} cno_debug;

int  cno_create(const cno_config* cfg, cno_sim** out);
void cno_destroy(cno_sim* s);
int  cno_obs_dim(const cno_sim* s);
/* initial pedestrian poses [N,P,2] (default: seeded uniform) and constant
 * preset velocities [N,P,2] for ped_mode 1 */
int  cno_set_ped_init(cno_sim* s, const double* xy);
int  cno_set_ped_preset_vel(cno_sim* s, const double* vxy);
int  cno_get_ped_init(const cno_sim* s, double* xy);

/* Full simulated path (physics + sensing + Env logic). mask NULL = all envs.
 * obs is [N, 366+4K] float64. */
int  cno_reset(cno_sim* s, const uint8_t* mask, double* obs);
/* step_counter NULL -> internal 1-based per-episode counter (TRAIN:125 passes step+1).
 * auto_reset != 0: envs that finish are reset inside the call; obs then holds the
 * first observation of the new episode and final_obs (nullable) the terminal one. */
int  cno_step(cno_sim* s, const double* action, const int32_t* step_counter, int auto_reset,
              double* obs, double* final_obs, double* reward, uint8_t* done, int32_t* topk_idx);
int  cno_get_counters(const cno_sim* s, int32_t* out /* [N,6] ego,social,obst_steps,ep_steps,success,failure */);
int  cno_get_returns(const cno_sim* s, double* out /* [N] return of the last finished episode */);
int  cno_get_sim_state(const cno_sim* s, int env, double* robot5, double* ped_p, double* ped_v, double* ranges);
int  cno_get_debug(const cno_sim* s, int env, cno_debug* out);
int  cno_set_num_threads(int n);

/* Externally driven path (golden replay): the caller supplies what Gazebo supplied. */
typedef struct cno_ext_in {
    double deque_x, deque_y;    /* position at the end of time.sleep (ENV:1208), unrounded */
    double end_timestep;        /* ENV:1202 */
    double px, py, yaw;         /* odom at get_state time */
    double v, w;                /* linear_twist.x, angular_twist.z */
    double now;                 /* time.time() inside get_state */
    int32_t step_counter;
    int32_t is_reset;           /* 1: Env.reset() flow, 0: Env.step() flow */
} cno_ext_in;
int  cno_ext_call(cno_sim* s, int env, const cno_ext_in* in, const double* ranges,
                  double* obs, double* reward, uint8_t* done, int32_t* topk_idx);
void cno_ext_set_done(cno_sim* s, int env, int done);

/* function-level entry points (golden vectors) */
double cno_py_round(double x, int ndigits);
double cno_np_around(double x, int ndigits);
void   cno_det_sincos(double x, double* s, double* c);
double cno_hypot(double x, double y);   /* math.hypot as this image's C library computes it, spelled out (cn_oracle.c) */
void   cno_hypot_array(int n, const double* x, const double* y, double* out);
void   cno_scan_sanitize(const double* ranges, int R, double max_range, double* scan);
void   cno_scan_to_points(const double* scan, int R, double px, double py, double yaw, double* pts);
int    cno_waypoint(double ax, double ay, double gx, double gy, double radius, double* wp);
int    cno_collision_point(double a0x, double a0y, double a1x, double a1y,
                           double ox, double oy, double radius, double* dist);
int    cno_collision_point_geos(double a0x, double a0y, double a1x, double a1y, double ox, double oy, double radius,
                                int untyped_empty, double* dist);
double cno_iou(double ax, double ay, double bx, double by, double half);
double cno_collision_prob(double ttc);
double cno_general_collision_prob(double d, double max_range, double min_range);
int    cno_topk(const double* cp, int n, int K, int32_t* idx_out);
double cno_heading_to_goal(cno_sim* s, int env, double wpx, double wpy, double px, double py, double yaw);
double cno_distance_to_goal(double px, double py, double wpx, double wpy);
int    cno_in_box(double x, double y, double gx, double gy, double eps);
double cno_compute_reward(cno_sim* s, int env, double cur_head, double cur_dist, double prev_head, double prev_dist,
                          double wpx, double wpy, double px, double py, int done);
double cno_bbox_size(const double* pts, int n);
int    cno_estimate_num_obs_scans(double d, double max_range, double min_range);
void   cno_raycast(const cno_config* cfg, double rx, double ry, double ryaw,
                   const double* ped_xy, int P, double* ranges);
double cno_rng_u01(uint64_t seed, int64_t env, uint32_t stream, uint32_t a, uint32_t b);

/* state exchange in the product's snapshot layout (include/crowdnav.h CN_SD_* / CN_SI_* / CN_TF_*) */
int    cno_set_state(cno_sim* s, int env, const double* sd, const int32_t* si, const double* ped_p, const double* ped_v,
                     const double* trk, int trk_cap, const double* ped_init, const double* ped_preset, const double* ped_aux);
int    cno_get_state(const cno_sim* s, int env, double* sd, int32_t* si, double* ped_p, double* ped_v, double* trk, int trk_cap,
                     double* ped_aux);
double cno_det_exp(double x);
void   cno_set_py2_round(int on);   /* for the function-level entry points that take no handle */

/* simulator-only entry points used by oracle/harness (it plays Gazebo for the reference) */
int    cno_hsim_reset(cno_sim* s, int env);
int    cno_hsim_advance(cno_sim* s, int env, int32_t ms, double v, double w);
int    cno_hsim_scan(cno_sim* s, int env, double* ranges);
int    cno_set_robot(cno_sim* s, int env, double x, double y, double yaw);

#ifdef __cplusplus
}
#endif
#endif
