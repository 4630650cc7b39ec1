"""ctypes binding of libcrowdnav.so (include/crowdnav.h).  This is the binding INTEGRATION.md shows
a maintainer of the reference adding; there is no CPU fallback -- importing works anywhere, but
creating an environment raises unless the HIP library loads and a GPU is present."""
import ctypes as C
import os
import subprocess

from .config import CnConfig

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "lib", "libcrowdnav.so")
BUILD_SH = os.path.join(_PKG, "csrc", "build.sh")

CN_MAX_TRACKS = 64
EXPECTED_ABI = 7       # the version the ctypes structs below were written against (include/crowdnav.h CN_ABI_VERSION)
CN_PHASE_ALL, CN_PHASE_PRE, CN_PHASE_GET_STATE, CN_PHASE_REWARD = 0, 1, 2, 4
CN_SD_COUNT = 24
CN_SI_COUNT = 16
CN_TF_COUNT = 12
CN_ARB_AUTO, CN_ARB_OLDEST_FIRST, CN_ARB_FAIR = 0, 1, 2      # include/crowdnav.h: cn_set_arbitration
CN_COUNTER_COLS = 14
SD = dict(RX=0, RY=1, RYAW=2, RV=3, RW=4, CLOCK=5, WPX=6, WPY=7, PREV_DIST=8, PREV_HEAD=9, DQ0X=10, DQ0Y=11,
          DQ1X=12, DQ1Y=13, TS=14, BB=15, EGO=16, CPROB=17, EP_RETURN=18, LAST_RETURN=19)
SI = dict(DONE=0, DQ_LEN=1, NTRACKS=2, EGO_VIOL=3, SOCIAL_VIOL=4, OBST_STEPS=5, SUCCESS=6, FAILURE=7, EP_STEP=8,
          STATUS=9, NCONF=10, NENTRIES=11, PENDING_RESET=14, EPISODES=15)
TF = dict(PX=0, PY=1, DIST=2, D0X=3, D0Y=4, D1X=5, D1Y=6, T=7, SPEED=8, VX=9, VY=10, DQLEN=11)

EXPORTS = ["cn_abi_version", "cn_last_error", "cn_create", "cn_destroy", "cn_obs_dim", "cn_config_of",
           "cn_set_ped_init", "cn_get_ped_init", "cn_set_ped_preset_vel", "cn_reset", "cn_step", "cn_step_multi", "cn_set_arbitration", "cn_get_arbitration", "cn_set_group_envs", "cn_kernel_name", "cn_device_clock",
           "cn_observe_external",
           "cn_policy_tail", "cn_actor_pack_weights", "cn_actor_forward", "cn_step_sequence", "cn_rollout_policy", "cn_get_counters",
           "cn_get_returns", "cn_debug_env", "cn_lds_bytes", "cn_near_separate", "cn_snapshot_size", "cn_snapshot", "cn_restore",
           "cn_td3_create", "cn_td3_destroy", "cn_td3_update", "cn_td3_loss_dev", "cn_td3_last_error",
           "cn_replay_write", "cn_episode_log_add"]


class CnStepIO(C.Structure):
    _fields_ = [("action", C.c_void_p), ("step_counter", C.c_void_p), ("obs", C.c_void_p), ("final_obs", C.c_void_p),
                ("obs_f64", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("topk_idx", C.c_void_p),
                ("auto_reset", C.c_int32), ("reserved", C.c_int32)]


class CnExternalIO(C.Structure):
    _fields_ = [("ranges", C.c_void_p), ("odom", C.c_void_p), ("step_counter", C.c_void_p), ("obs", C.c_void_p),
                ("obs_f64", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("topk_idx", C.c_void_p),
                ("is_reset", C.c_int32), ("phase", C.c_int32)]


CN_SNAPSHOT_MAGIC = 0x50414E534E43        # "CNSNAP"


class CnSnapshotHeader(C.Structure):
    """Mirror of `cn_snapshot_header` (include/crowdnav.h): what a snapshot blob starts with."""
    _fields_ = [("magic", C.c_uint64), ("abi_version", C.c_int32), ("header_bytes", C.c_int32),
                ("sd_count", C.c_int32), ("si_count", C.c_int32), ("tf_count", C.c_int32), ("track_capacity", C.c_int32),
                ("total_bytes", C.c_uint64), ("config", CnConfig)]


def split_snapshot(buf):
    """A cn_snapshot blob (uint8 array) -> (header, dict of arrays): sd [N,24] f64, si [N,16] i32, ped_p / ped_v [N,P,2],
    trk [N,cap,12], ped_init / ped_preset [N,P,2], ped_aux [N,P,3].  Views into `buf`."""
    import numpy as np
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    hd = CnSnapshotHeader.from_buffer_copy(buf[:C.sizeof(CnSnapshotHeader)].tobytes())
    if hd.magic != CN_SNAPSHOT_MAGIC or hd.header_bytes != C.sizeof(CnSnapshotHeader):
        raise CrowdNavError("not a libcrowdnav snapshot (magic / header size)")
    N, P, cap = hd.config.n_envs, hd.config.n_peds, hd.track_capacity
    off = [hd.header_bytes]

    def take(shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        a = buf[off[0]:off[0] + n].view(dtype).reshape(shape)
        off[0] += n
        return a
    out = dict(sd=take((N, hd.sd_count), np.float64), si=take((N, hd.si_count), np.int32), ped_p=take((N, P, 2), np.float64),
               ped_v=take((N, P, 2), np.float64), trk=take((N, cap, hd.tf_count), np.float64), ped_init=take((N, P, 2), np.float64),
               ped_preset=take((N, P, 2), np.float64), ped_aux=take((N, P, 3), np.float64))
    if off[0] != hd.total_bytes or off[0] > buf.size:
        raise CrowdNavError("truncated snapshot: %d bytes, header says %d" % (buf.size, hd.total_bytes))
    return hd, out


def join_snapshot(hd, arrays):
    """Inverse of split_snapshot: header + arrays -> blob for cn_restore."""
    import numpy as np
    parts = [np.frombuffer(bytes(hd), dtype=np.uint8)]
    for k, dt in (("sd", np.float64), ("si", np.int32), ("ped_p", np.float64), ("ped_v", np.float64), ("trk", np.float64),
                  ("ped_init", np.float64), ("ped_preset", np.float64), ("ped_aux", np.float64)):
        parts.append(np.ascontiguousarray(arrays[k], dtype=dt).reshape(-1).view(np.uint8))
    return np.concatenate(parts)


def config_to_dict(c):
    return {name: getattr(c, name) for name, _ in CnConfig._fields_}


class CnActorWeights(C.Structure):
    _fields_ = [("w1p", C.c_void_p), ("b1", C.c_void_p), ("w2p", C.c_void_p), ("b2", C.c_void_p), ("w3", C.c_void_p),
                ("b3", C.c_void_p), ("obs_dim", C.c_int32), ("obs_dim_padded", C.c_int32), ("hidden", C.c_int32),
                ("reserved", C.c_int32)]


class CnTd3Mlp(C.Structure):
    """Mirror of `cn_td3_mlp`: device pointers to one network's nn.Linear storages."""
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "w2", "b2", "w3", "b3")]


class CnTd3Config(C.Structure):
    """Mirror of `cn_td3_config` (include/crowdnav.h)."""
    _fields_ = [("obs_dim", C.c_int32), ("hidden", C.c_int32), ("batch", C.c_int32), ("policy_delay", C.c_int32),
                ("gamma", C.c_float), ("tau", C.c_float), ("lr_actor", C.c_float), ("lr_critic", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("noise_std", C.c_float), ("noise_clip", C.c_float), ("max_v", C.c_float),
                ("max_w", C.c_float), ("reserved", C.c_float),
                ("actor", CnTd3Mlp), ("actor_t", CnTd3Mlp), ("q1", CnTd3Mlp), ("q1_t", CnTd3Mlp), ("q2", CnTd3Mlp), ("q2_t", CnTd3Mlp),
                ("replay_s", C.c_void_p), ("replay_a", C.c_void_p), ("replay_r", C.c_void_p), ("replay_s2", C.c_void_p), ("replay_d", C.c_void_p),
                ("replay_size_dev", C.c_void_p), ("seed", C.c_uint64)]


class CnTd3Batch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("s", "a", "r", "s2", "d", "target_noise")]


class CnReplayRing(C.Structure):
    """Mirror of `cn_replay_ring` (include/crowdnav.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("s", "a", "r", "s2", "d")] + [("capacity", C.c_int64), ("pos_dev", C.c_void_p),
                                                                         ("size_dev", C.c_void_p), ("obs_dim", C.c_int32), ("reserved", C.c_int32)]


class CnEpisodeLog(C.Structure):
    """Mirror of `cn_episode_log` (include/crowdnav.h)."""
    _fields_ = [("rows", C.c_void_p), ("max_rows", C.c_int64), ("n_dev", C.c_void_p), ("tot_dev", C.c_void_p)]


class CnSequenceIO(C.Structure):
    """Mirror of `cn_sequence_io` (include/crowdnav.h)."""
    _fields_ = [("action", C.c_void_p), ("obs", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("topk_idx", C.c_void_p),
                ("action_stride", C.c_int64), ("obs_stride", C.c_int64), ("reward_stride", C.c_int64), ("done_stride", C.c_int64),
                ("topk_stride", C.c_int64), ("n_steps", C.c_int32), ("reserved", C.c_int32)]


class CnPolicyIO(C.Structure):
    """Mirror of `cn_policy_io` (include/crowdnav.h)."""
    _fields_ = [("obs0", C.c_void_p), ("action", C.c_void_p), ("obs", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p),
                ("topk_idx", C.c_void_p),
                ("action_stride", C.c_int64), ("obs_stride", C.c_int64), ("reward_stride", C.c_int64), ("done_stride", C.c_int64),
                ("topk_stride", C.c_int64), ("n_steps", C.c_int32), ("reserved", C.c_int32),
                ("max_v", C.c_float), ("max_w", C.c_float), ("sigma", C.c_float), ("reserved_f", C.c_float),
                ("seed", C.c_uint64), ("counter", C.c_uint64)]


class CrowdNavError(RuntimeError):
    pass


def build(force=False):
    """Compile libcrowdnav.so for gfx950 (hipcc cross-compiles without a GPU).  Safe to call from several ranks
    at once: the staleness check and the build run under a file lock and build.sh renames the finished library into
    place, so no process ever maps a half-written file."""
    import fcntl
    srcs = [os.path.join(_PKG, "csrc", f) for f in os.listdir(os.path.join(_PKG, "csrc"))]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "crowdnav.h"))

    def stale():
        return force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)

    if stale():
        os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
        with open(os.path.join(os.path.dirname(LIB_PATH), ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if stale():          # another rank may have built it while we waited
                    subprocess.check_call(["bash", BUILD_SH])
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB_PATH


def build_timing(force=False):
    """Compile the PROFILING build lib/libcrowdnav_timing.so (stage time stamps, ablation mask, PMC calibration kernels:
    tools/stage_timing.py, tools/ablate.py, tools/calib_pmc.py).  Never loaded by the product."""
    tpath = LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so")
    srcs = [os.path.join(_PKG, "csrc", f) for f in os.listdir(os.path.join(_PKG, "csrc"))]
    if force or not os.path.exists(tpath) or os.path.getmtime(tpath) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["bash", BUILD_SH, "timing"])
    return tpath


_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            build()          # no-op when lib/libcrowdnav.so is newer than every source
        except Exception as ex:  # hipcc missing: use what is there, or fail loudly below
            if not os.path.exists(LIB_PATH):
                raise CrowdNavError("libcrowdnav.so is not built (%s) and cannot be built here (%s); there is no CPU "
                                    "fallback" % (LIB_PATH, ex))
            import warnings
            warnings.warn("libcrowdnav.so could not be rebuilt (%s); loading the existing %s -- its ABI version is "
                          "checked below" % (ex, LIB_PATH))
        # PyTorch-ROCm ships its own HIP / ROCr runtime libraries; the caller's tensors live in THAT runtime.  Importing
        # torch first makes libcrowdnav.so's libamdhip64 dependency resolve to the copy that is already loaded.  Loaded
        # the other way round the process ends up with two runtimes and cn_create sees no device (CN_ERR_NO_DEVICE).
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.cn_abi_version.restype = C.c_int
        if L.cn_abi_version() != EXPECTED_ABI:   # a stale library with other struct layouts would corrupt kernel arguments
            raise CrowdNavError("%s reports ABI %d but this binding is written against ABI %d: rebuild it "
                                "(csrc/build.sh)" % (LIB_PATH, L.cn_abi_version(), EXPECTED_ABI))
        L.cn_last_error.restype = C.c_char_p
        L.cn_kernel_name.argtypes = [C.c_void_p, C.c_int]; L.cn_kernel_name.restype = C.c_char_p
        L.cn_set_group_envs.argtypes = [C.c_void_p, C.c_int64]
        L.cn_device_clock.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.cn_create.argtypes = [C.POINTER(CnConfig), C.c_int, C.POINTER(vp)]
        L.cn_destroy.argtypes = [vp]; L.cn_destroy.restype = None
        L.cn_obs_dim.argtypes = [vp]
        L.cn_config_of.argtypes = [vp, C.POINTER(CnConfig)]
        L.cn_set_ped_init.argtypes = [vp, vp]
        L.cn_get_ped_init.argtypes = [vp, vp]
        L.cn_set_ped_preset_vel.argtypes = [vp, vp]
        L.cn_reset.argtypes = [vp, vp, vp, vp, vp]
        L.cn_step.argtypes = [vp, C.POINTER(CnStepIO), vp]
        L.cn_step_multi.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(CnStepIO), C.POINTER(vp)]
        L.cn_set_arbitration.argtypes = [vp, C.c_int]
        L.cn_get_arbitration.argtypes = [vp]
        L.cn_observe_external.argtypes = [vp, C.POINTER(CnExternalIO), vp]
        L.cn_policy_tail.argtypes = [vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_uint64, C.c_int, vp]
        L.cn_actor_pack_weights.argtypes = [vp, C.c_int, vp, C.c_int, vp]
        L.cn_actor_forward.argtypes = [C.POINTER(CnActorWeights), vp, vp, C.c_int, C.c_float, C.c_float, C.c_float,
                                       C.c_uint64, C.c_uint64, C.c_int, vp]
        L.cn_step_sequence.argtypes = [vp, C.POINTER(CnSequenceIO), vp]
        L.cn_rollout_policy.argtypes = [vp, C.POINTER(CnActorWeights), C.POINTER(CnPolicyIO), vp]
        L.cn_get_counters.argtypes = [vp, vp, vp]
        L.cn_get_returns.argtypes = [vp, vp, vp, vp]
        L.cn_debug_env.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.cn_snapshot_size.argtypes = [vp]; L.cn_snapshot_size.restype = C.c_size_t
        L.cn_snapshot.argtypes = [vp, vp, C.c_size_t]
        L.cn_restore.argtypes = [vp, vp, C.c_size_t]
        L.cn_td3_create.argtypes = [C.POINTER(CnTd3Config), C.c_int, C.POINTER(vp)]
        L.cn_td3_destroy.argtypes = [vp]; L.cn_td3_destroy.restype = None
        L.cn_td3_update.argtypes = [vp, C.c_int, C.POINTER(CnTd3Batch), vp]
        L.cn_td3_loss_dev.argtypes = [vp]; L.cn_td3_loss_dev.restype = vp
        L.cn_td3_last_error.restype = C.c_char_p
        L.cn_replay_write.argtypes = [C.POINTER(CnReplayRing), vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, vp]
        L.cn_episode_log_add.argtypes = [C.POINTER(CnEpisodeLog), vp, vp, C.c_int, vp, vp, C.c_float, C.c_int, C.c_int, vp]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise CrowdNavError("libcrowdnav error %d: %s" % (rc, lib().cn_last_error().decode()))
