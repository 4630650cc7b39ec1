// crowdnav_kernel.h -- kernel argument block shared by the device code and the C-ABI host code.
#pragma once
#include <stdint.h>
#include "../../include/crowdnav.h"

enum { CN_MODE_STEP = 0, CN_MODE_RESET = 1, CN_MODE_EXT_STEP = 2, CN_MODE_EXT_RESET = 3 };
#define CN_MAXW 17         /* 64-ray bit words per env: ceil(1024/64) + 1 */
#define CN_NMASK 13        /* number of bit-word masks kept in LDS (M_COUNT in the kernel) */

struct CnKParams {
    // sizes
    int32_t N, P, R, K;
    int32_t max_steps, ped_mode, dt_ms, scan_latency_ms, settle_ms, ped_cycle_ms, ped_stagger_ms;
    int32_t mode, auto_reset, max_conf, trk_cap;
    int32_t near_sep;        // near-pedestrian list: 1 = own LDS region, 0 = overlaid on region B (cn_near_separate)
    int32_t ablate;          // PROFILING BUILD ONLY (cn_debug_set_ablate): skips stages, results are then invalid
    int32_t ext_phase;       // cn_external_io.phase (CN_PHASE_* mask, 0 = whole flow); CN_MODE_EXT_STEP only
    int32_t geos_untyped_empty, ped_contact, risk_mode, py2_round;   // cn_config switches
    int32_t sf_tick_ms, sf_pair_matrix;   // ped_mode 2: physics tick; 1 = the pair matrix fits in the simulator's LDS scratch
    int32_t sf_pair_cap, sf_reserved;     // dense social force: near pairs (2 bytes each) that fit in the scratch behind next / aux / acc (0: none)
    int32_t lidar_min_positive;           // lidar_min > 0: the simulated sensor never returns a range of exactly 0 (UTL:382's test is then dead)
    int32_t scan_f32, waypoint_reward;    // cn_config: float32 LaserScan.ranges; ENV:1116's way-point reward (200, or 0 = as logged)
    int64_t env_index_base;
    uint64_t seed;
    // constants (cn_config)
    double room_half, ped_radius, ped_vmax, robot_clearance, lidar_min, lidar_max, lidar_offset_x;
    double max_scan_range, min_scan_range, goal_x, goal_y, start_x, start_y, spawn_x, spawn_y, spawn_yaw;
    double waypoint_radius, goal_eps, angle_inc_deg, lidar_step;
    double ped_inv_cycle;    // 1.0 / ped_cycle_ms
    double sf_tau, sf_A, sf_B, sf_wall_A, sf_wall_B, sf_goal_eps2;   // ped_mode 2 (social force): cn_config.sf_*, goal radius squared
    double wheel_accel, wheel_sep;   // cn_config.wheel_accel (0 = kinematic robot), wheel_separation
    double blk_cb, blk_sb;   // cos / sin of the half-width (32.5 lidar steps) of a 64-ray block (near-pedestrian block bits)
    const double* blk_dir;   // [ceil(R/64)][2] robot-frame direction of ray 64 q + 32 (clamped to R - 1)
    double trig[34];         // constants of cn_det_sincos_t / cn_atan2_t (CN_TRIG_TABLE): scalar loads next to the polynomials
    double bb_spawn;         // bounding-box size (UTL:405-419) at the spawn pose, evaluated on the device by cn_create
    int64_t bb_spawn_valid;
    // The association table (see "ENV:448-485" in the kernel) for bounding-box size bb_spawn, built once by cn_create with the
    // arithmetic the kernel would use: every env of a simulated run keeps that size from reset to reset, so a wavefront copies
    // these <= 256 shorts instead of re-deriving them every observation (three float64 divides and ~80 more instructions).
    int32_t assoc_k1, assoc_fast;     // K1; 1 = the integer test is valid for this size (assoc_tab holds K1 + 2 entries)
    const int16_t* assoc_tab;         // device memory (a vector load from the kernel-argument block itself is a trip to host-visible memory)
    // tables (device)
    const double* lidar_c;  // [R] cos(k * span/(R-1)), deterministic sincos
    const double* lidar_s;  // [R]
    const double* ang_s;    // [R-1] sin(radians(j * angle_inc_deg))
    const double* ang_c;    // [R-1]
    const double* poly_c;   // [64] cos(-k*pi/32)
    const double* poly_s;   // [64]
    // env state (device, library-owned).  One record per env, `state_stride` bytes (a multiple of 128) apart:
    //   [ sd: CN_SD_COUNT f64 | si: CN_SI_COUNT i32 | ped_p: 2P f64 | ped_v: 2P f64 | pad ]
    // so a wavefront's load/store of its env touches whole 128-byte lines only.
    char* state;
    int64_t state_stride;
    const double* ped_init; // [N, P, 2]
    const double* ped_preset; // [N, P, 2]
    double* trk;            // [N, trk_cap, CN_TF_COUNT]: one contiguous 96-byte record per track slot
    double* ped_aux;        // [N, P, 3] ped_mode 2: goal x, goal y, goal counter of every pedestrian
    // caller-owned I/O (device)
    const float* action;
    const int32_t* step_counter;
    const uint8_t* mask;
    const double* ext_ranges;  // [N, R] externally supplied lidar ranges (CN_MODE_EXT_*), else NULL
    const double* ext_odom;    // [N, 10]
    float* obs;
    float* final_obs;
    double* obs_f64;
    float* reward;
    uint8_t* done;
    int32_t* topk_idx;
    // cn_step_sequence (cn_env_kernel_seq): steps per launch and the elements between consecutive steps' slots of the action /
    // output buffers (0 = one slot: the same actions every step / every step's outputs in place)
    int64_t roll_steps, roll_action_in_stride, roll_obs_stride, roll_reward_stride, roll_done_stride, roll_topk_stride;
    // cn_rollout_policy (cn_policy_kernel_s360): the TD3 actor of cn_actor_forward inside the step loop.  `action` above is then an
    // OUTPUT (slot t = the action step t took); pol_obs0 = the observation the first action is computed from; pol_wave_lds = bytes
    // between the LDS working sets of a workgroup's 16 environments
    const float *pol_w1p, *pol_b1, *pol_w2p, *pol_b2, *pol_w3, *pol_b3, *pol_obs0;
    uint64_t pol_seed, pol_counter;
    float pol_max_v, pol_max_w, pol_sigma;
    int32_t pol_D, pol_Dp, pol_wave_lds;
    int32_t wave_lds, wave_lds_pad;   // cn_env_kernel*_w4 (4 environments per workgroup): bytes between their LDS working sets
    int32_t pol_envs, pol_act_off;    // environments per workgroup (16 or 8); byte offset of the workgroup's [pol_envs][2] actions in its LDS
    long long* timing;      // profiling build only: [N, 32] s_memtime stamps
};

#define CN_ST_OFF_SD 0
#define CN_ST_OFF_SI (CN_SD_COUNT * 8)
#define CN_ST_OFF_PED_P (CN_SD_COUNT * 8 + CN_SI_COUNT * 4)
#define CN_ST_OFF_PED_V(P) (CN_ST_OFF_PED_P + 16 * (size_t)(P))
#define CN_ST_STRIDE(P) ((CN_ST_OFF_PED_P + 32 * (size_t)(P) + 127) & ~(size_t)127)

#ifdef __cplusplus
extern "C" {
#endif
size_t cn_lds_bytes(int R, int P, int K, int max_conf, int trk_cap);
int cn_near_separate(int R, int P, int K, int max_conf, int trk_cap);
#ifdef __cplusplus
}
#endif
