"""Host-side mirror of the reference's `Env` (environment_stage_1_nobonus.py:42) over the C-ABI.

`VecEnv`  N environments on one MI355X; tensors stay on the device (zero-copy into the actor).
`Env`     the reference's single-robot surface -- Env(action_dim, max_step), reset(), step(action,
          step_counter, mode), get_episode_status(), get_social/ego_safety_violation_status(),
          shutdown(), writable `done`, readable `k_obstacle_count` -- so the loop of
          start_td3_training.py:106-166 runs unchanged.
PyTorch is used only for device memory and streams.
"""
import contextlib
import ctypes as C
import time

import numpy as np
import torch

from . import _abi
from .config import Config


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class VecEnv:
    def __init__(self, cfg=None, device=0, stream=None, out=None, arbitration="auto", **kw):
        """stream: a torch.cuda.Stream every launch of this handle goes to (default: the current stream at
        call time).  out: dict of preallocated output tensors (obs, final_obs, reward, done, topk_idx) --
        VecEnvGroups passes row slices of one [N_total, ...] allocation.  arbitration: "auto" | "oldest_first" | "fair"
        (cn_set_arbitration: how the environments that share a SIMD share its issue slots; timing only, never a result)."""
        self.cfg = cfg if cfg is not None else Config(**kw)
        self.stream = stream
        if not torch.cuda.is_available():
            raise _abi.CrowdNavError("VecEnv needs a HIP device: libcrowdnav.so has no CPU fallback")
        self.L = _abi.lib()
        self.device = torch.device("cuda", device)
        self.h = C.c_void_p()
        ccfg = self.cfg.to_c()
        _abi.check(self.L.cn_create(C.byref(ccfg), int(device), C.byref(self.h)))
        self.N, self.P, self.R, self.K = self.cfg.n_envs, self.cfg.n_peds, self.cfg.n_rays, self.cfg.k_obstacles
        self.set_arbitration(arbitration)
        self.D = self.L.cn_obs_dim(self.h)
        N, D, K, dev = self.N, self.D, self.K, self.device
        out = out or {}

        def buf(name, shape, dtype, fill=0):
            t = out.get(name)
            if t is None:
                return torch.full(shape, fill, dtype=dtype, device=dev)
            assert tuple(t.shape) == tuple(shape) and t.dtype == dtype and t.is_contiguous() and t.device == dev, name
            return t
        self.obs = buf("obs", (N, D), torch.float32)
        self.final_obs = buf("final_obs", (N, D), torch.float32)
        self.obs_f64 = None
        self.reward = buf("reward", (N,), torch.float32)
        self.done = buf("done", (N,), torch.uint8)
        self.topk_idx = buf("topk_idx", (N, K), torch.int32, -1)
        self._counters = torch.zeros((N, _abi.CN_COUNTER_COLS), dtype=torch.int32, device=dev)
        self._ret = torch.zeros(N, dtype=torch.float32, device=dev)
        self._run = torch.zeros(N, dtype=torch.float32, device=dev)

    ARBITRATION = {"auto": _abi.CN_ARB_AUTO, "oldest_first": _abi.CN_ARB_OLDEST_FIRST, "fair": _abi.CN_ARB_FAIR}

    def set_arbitration(self, mode):
        """cn_set_arbitration: "auto" (fair when this handle's launch fills the device on its own), "oldest_first", "fair"."""
        if mode not in self.ARBITRATION:
            raise ValueError("arbitration must be one of %s" % sorted(self.ARBITRATION))
        _abi.check(self.L.cn_set_arbitration(self.h, self.ARBITRATION[mode]))

    @property
    def arbitration(self):
        """What cn_step uses for this handle right now: "oldest_first" or "fair"."""
        rc = self.L.cn_get_arbitration(self.h)
        if rc < 0:
            _abi.check(rc)
        return {_abi.CN_ARB_OLDEST_FIRST: "oldest_first", _abi.CN_ARB_FAIR: "fair"}[rc]

    KERNEL_OF = {"step": 0, "reset": 0, "same": 1, "sequence": 2, "external": 3, "multi": 4, "policy": 5}

    def kernel_name(self, what="step"):
        """cn_kernel_name: the device kernel a call on this handle launches right now -- "step" (cn_step with auto_reset
        "next" / none, cn_reset), "same" (same-call reset), "sequence", "external", "multi" (inside a cn_step_multi over
        several handles), "policy" (cn_rollout_policy).  A handle of the headline shape gets the `_s360` kernels."""
        n = self.L.cn_kernel_name(self.h, self.KERNEL_OF[what])
        if n is None:
            _abi.check(-1)
        return n.decode()

    def close(self):
        if getattr(self, "h", None):
            self.L.cn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        s = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        return C.c_void_p(s.cuda_stream)

    def _on_stream(self):
        """Context in which host->device temporaries of a call are allocated and copied: the stream the kernel is
        launched on, so the copy is ordered before the launch and the caching allocator ties the block to that stream."""
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _dev(self, x, dtype, shape=None):
        """x as a contiguous device tensor of `dtype`; a tensor produced on another stream is ordered before our launch."""
        if isinstance(x, torch.Tensor) and x.device == self.device and x.dtype == dtype and x.is_contiguous():
            if self.stream is not None:
                self.stream.wait_stream(torch.cuda.current_stream(self.device))
                x.record_stream(self.stream)
            return x if shape is None else x.reshape(shape)
        with self._on_stream():
            if not isinstance(x, torch.Tensor):
                x = np.asarray(x, dtype={torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32,
                                         torch.uint8: np.uint8}[dtype])
            t = torch.as_tensor(x, dtype=dtype, device=self.device).contiguous()
        return t if shape is None else t.reshape(shape)

    def enable_f64_obs(self):
        """Also produce the observation in float64 (the reference's dtype) -- used by parity tests."""
        if self.obs_f64 is None:
            self.obs_f64 = torch.zeros((self.N, self.D), dtype=torch.float64, device=self.device)
        return self.obs_f64

    def set_ped_init(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(self.N, self.P, 2)
        _abi.check(self.L.cn_set_ped_init(self.h, xy.ctypes.data))

    def get_ped_init(self):
        xy = np.zeros((self.N, self.P, 2))
        _abi.check(self.L.cn_get_ped_init(self.h, xy.ctypes.data))
        return xy

    def set_ped_preset_vel(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.N, self.P, 2)
        _abi.check(self.L.cn_set_ped_preset_vel(self.h, v.ctypes.data))

    def reset(self, mask=None):
        """Env.reset() for every env (or the masked ones) -> obs [N, D] float32 on the device."""
        m = None
        if mask is not None:
            m = self._dev(mask.to(torch.uint8) if isinstance(mask, torch.Tensor) else np.asarray(mask).astype(np.uint8), torch.uint8)
        _abi.check(self.L.cn_reset(self.h, _ptr(m), _ptr(self.obs), _ptr(self.obs_f64), self._stream()))
        self._keep_reset = m      # alive until the next call: the launch is asynchronous
        return self.obs

    _warned_default_reset = False
    _UNSET = object()          # "auto_reset omitted" (-> "next" with a one-time warning); an explicit None keeps meaning False

    def step(self, action, step_counter=None, auto_reset=_UNSET, want_final=False):
        """Env.step for every env.  action: [N,2] float32 device tensor (v, w).
        auto_reset: "next" (a finished env returns its TERMINAL observation with done = 1 and spends the next call on Env.reset
        -- that call ignores its action and returns the new episode's first observation with reward 0, done 0; one observation
        per wavefront, the fast kernel) | True / "same" (finished envs run Env.reset inside the same call: obs = the new
        episode's first observation, the terminal one in final_obs when want_final; a launch then lasts two observations for
        any env that finishes -- ~55 % of the speed) | False / None (no reset: the caller resets).
        Omitted: "next" (the default since round 4; it was "same" before) with a one-time warning, because a caller written for
        the same-call convention that stores EVERY returned row as a transition would silently record reset launches: under
        "next" the rows of the call AFTER done = 1 are not transitions -- keep `resetting = done.bool().clone()` from one call
        and drop those rows of the next (crowdnav.rollout.rollout does exactly that).
        Returns (obs, reward, done) device tensors (views of internal buffers)."""
        if auto_reset is VecEnv._UNSET:
            auto_reset = "next"
            if not VecEnv._warned_default_reset:
                VecEnv._warned_default_reset = True
                import warnings
                warnings.warn('VecEnv.step(): auto_reset not given -> "next" (next-step reset: the call after done = 1 is the reset, '
                              'its row is not a transition); pass auto_reset="next" or "same" explicitly', stacklevel=2)
        a = action
        if not (isinstance(a, torch.Tensor) and a.device == self.device and a.dtype == torch.float32 and a.is_contiguous()):
            a = self._dev(action, torch.float32)
        sc = None
        if step_counter is not None:
            sc = self._dev(step_counter, torch.int32)
        self._keep_step = (a, sc)     # alive until the next call: the launch is asynchronous
        io = _abi.CnStepIO(action=a.data_ptr(), step_counter=sc.data_ptr() if sc is not None else None,
                           obs=self.obs.data_ptr(), final_obs=self.final_obs.data_ptr() if want_final else None,
                           obs_f64=self.obs_f64.data_ptr() if self.obs_f64 is not None else None,
                           reward=self.reward.data_ptr(), done=self.done.data_ptr(), topk_idx=self.topk_idx.data_ptr(),
                           auto_reset={False: 0, None: 0, True: 1, "same": 1, "next": 2}.get(auto_reset, auto_reset),
                           reserved=0)
        _abi.check(self.L.cn_step(self.h, C.byref(io), self._stream()))
        return self.obs, self.reward, self.done

    def bind_step(self, action, auto_reset="next", want_final=False):
        """Pre-marshalled cn_step for a fixed action buffer: returns a zero-argument callable that only enqueues
        (a fraction of the host time of step(), tools/host_overhead.py).  For rollouts that call several handles per step
        (VecEnvGroups): the action tensor is written in place by the policy, outputs land in self.obs/reward/done.
        Needs a bound stream (VecEnv(stream=...)) or uses the stream current at bind time."""
        assert action.device == self.device and action.dtype == torch.float32 and action.is_contiguous()
        io = _abi.CnStepIO(action=action.data_ptr(), step_counter=None, obs=self.obs.data_ptr(),
                           final_obs=self.final_obs.data_ptr() if want_final else None,
                           obs_f64=self.obs_f64.data_ptr() if self.obs_f64 is not None else None,
                           reward=self.reward.data_ptr(), done=self.done.data_ptr(), topk_idx=self.topk_idx.data_ptr(),
                           auto_reset={False: 0, True: 1, None: 0, "same": 1, "next": 2}.get(auto_reset, auto_reset),
                           reserved=0)
        ref, st, h, fn, check = C.byref(io), self._stream(), self.h, self.L.cn_step, _abi.check
        keep = (io, action)

        def call(_keep=keep):
            rc = fn(h, ref, st)
            if rc:
                check(rc)
        return call

    def step_sequence(self, actions, traj=None):
        """T = len(actions) calls of step(auto_reset="next") with OPEN-LOOP actions as ONE launch (cn_step_sequence): every
        wavefront keeps its env for the whole launch and walks the T steps at its own pace.  actions: [T, N, 2] float32 device
        tensor (or [N, 2] with `traj["steps"]` / an int second argument is not supported -- pass an expanded tensor).
        traj: None -- every step overwrites self.obs / reward / done / topk_idx in place -- or a dict of preallocated device
        tensors obs [T, N, D], reward [T, N], done [T, N] uint8, optionally topk_idx [T, N, K] (slot t = what step t returned).
        Bit-identical to the T calls.  Enqueues only; returns env-steps issued."""
        a = actions
        assert isinstance(a, torch.Tensor) and a.device == self.device and a.dtype == torch.float32 and a.dim() == 3 and a.shape[1:] == (self.N, 2)
        T = int(a.shape[0])
        if T > 1 and a.stride(0) == 0:
            astride = 0                                  # an expanded [N, 2]: the same actions held for T steps
            assert a[0].is_contiguous()
        else:
            assert a.is_contiguous()
            astride = 2 * self.N
        io = _abi.CnSequenceIO()
        io.action, io.action_stride, io.n_steps = a.data_ptr(), astride, T
        if traj is None:
            io.obs, io.reward, io.done, io.topk_idx = self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(), self.topk_idx.data_ptr()
        else:
            N, D, K = self.N, self.D, self.K
            o, r, d = traj["obs"], traj["reward"], traj["done"]
            assert tuple(o.shape) == (T, N, D) and o.dtype == torch.float32 and o.is_contiguous()
            assert tuple(r.shape) == (T, N) and r.dtype == torch.float32 and r.is_contiguous()
            assert tuple(d.shape) == (T, N) and d.dtype == torch.uint8 and d.is_contiguous()
            io.obs, io.reward, io.done = o.data_ptr(), r.data_ptr(), d.data_ptr()
            io.obs_stride, io.reward_stride, io.done_stride = N * D, N, N
            tk = traj.get("topk_idx")
            if tk is not None:
                assert tuple(tk.shape) == (T, N, K) and tk.dtype == torch.int32 and tk.is_contiguous()
                io.topk_idx, io.topk_stride = tk.data_ptr(), N * K
        _abi.check(self.L.cn_step_sequence(self.h, C.byref(io), self._stream()))
        self._keep_seq = (io, a, traj)
        if traj is not None:
            with self._on_stream():
                self.obs.copy_(traj["obs"][T - 1]); self.reward.copy_(traj["reward"][T - 1]); self.done.copy_(traj["done"][T - 1])
                if traj.get("topk_idx") is not None:
                    self.topk_idx.copy_(traj["topk_idx"][T - 1])
                else:
                    self.topk_idx.fill_(-1)      # not produced by this call: never left stale next to the new observation
        return T * self.N

    def _policy_io(self, agent, T, traj, add_noise, noise_seed, obs0):
        """cn_policy_io for T periods of `agent`'s packed actor (Agent.sync_fused_weights) on this handle's buffers."""
        if not hasattr(agent, "_fw_struct"):
            agent.sync_fused_weights()
        N, D, K = self.N, self.D, self.K
        io = _abi.CnPolicyIO()
        o0 = self.obs if obs0 is None else obs0
        assert o0.device == self.device and o0.dtype == torch.float32 and o0.is_contiguous() and tuple(o0.shape) == (N, D)
        io.obs0, io.n_steps = o0.data_ptr(), int(T)
        io.max_v, io.max_w, io.sigma = agent.max_v, agent.max_w, (agent.explore_sigma if add_noise else 0.0)
        io.seed = agent._noise_seed if noise_seed is None else int(noise_seed)
        if traj is None:
            if not hasattr(self, "_pol_action"):
                self._pol_action = torch.zeros((N, 2), dtype=torch.float32, device=self.device)
            io.action = self._pol_action.data_ptr()
            io.obs, io.reward, io.done, io.topk_idx = self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(), self.topk_idx.data_ptr()
        else:
            a, o, r, d = traj["action"], traj["obs"], traj["reward"], traj["done"]
            assert tuple(a.shape) == (T, N, 2) and a.dtype == torch.float32 and a.is_contiguous()
            assert tuple(o.shape) == (T, N, D) and o.dtype == torch.float32 and o.is_contiguous()
            assert tuple(r.shape) == (T, N) and r.dtype == torch.float32 and r.is_contiguous()
            assert tuple(d.shape) == (T, N) and d.dtype == torch.uint8 and d.is_contiguous()
            io.action, io.obs, io.reward, io.done = a.data_ptr(), o.data_ptr(), r.data_ptr(), d.data_ptr()
            io.action_stride, io.obs_stride, io.reward_stride, io.done_stride = 2 * N, N * D, N, N
            tk = traj.get("topk_idx")
            if tk is not None:
                assert tuple(tk.shape) == (T, N, K) and tk.dtype == torch.int32 and tk.is_contiguous()
                io.topk_idx, io.topk_stride = tk.data_ptr(), N * K
        return io, (o0, traj, agent._fw_struct, agent._fw)

    def rollout_policy(self, agent, n_steps, traj=None, add_noise=True, noise_seed=None, obs0=None):
        """n_steps periods of (agent.act_mfma -> step(auto_reset="next")) as ONE launch (cn_rollout_policy): the packed actor
        runs inside the step kernel, a workgroup of 16 envs joins only with itself, nothing returns to the host in between.
        The first action is computed from `obs0` (default: self.obs, i.e. what reset() / the previous call left).
        traj: None -- every period overwrites self.obs / reward / done / topk_idx in place (the last action taken is in
        self.last_policy_action) -- or a dict of preallocated device tensors action [T, N, 2], obs [T, N, D], reward [T, N],
        done [T, N] uint8, optionally topk_idx [T, N, K] (slot t: the action period t took and what its step returned; the
        transition of period t is (obs[t - 1] or obs0, action[t], reward[t], obs[t], done[t]), to be skipped where the env
        spent the period on its reset, i.e. where done[t - 1] was set).
        Bit-identical to the n_steps pairs of calls; advances the agent's noise counter by n_steps.  Enqueues only."""
        T = int(n_steps)
        io, keep = self._policy_io(agent, T, traj, add_noise, noise_seed, obs0)
        io.counter = agent._fused_calls + 1
        agent._fused_calls += T
        _abi.check(self.L.cn_rollout_policy(self.h, C.byref(agent._fw_struct), C.byref(io), self._stream()))
        self._keep_pol = (io, keep)
        if traj is not None:
            with self._on_stream():
                self.obs.copy_(traj["obs"][T - 1]); self.reward.copy_(traj["reward"][T - 1]); self.done.copy_(traj["done"][T - 1])
                if traj.get("topk_idx") is not None:
                    self.topk_idx.copy_(traj["topk_idx"][T - 1])
                else:
                    self.topk_idx.fill_(-1)
        return T * self.N

    @property
    def last_policy_action(self):
        return getattr(self, "_pol_action", None)

    def bind_rollout_policy(self, agent, n_steps, add_noise=True, noise_seed=None):
        """Pre-marshalled rollout_policy (in place): a zero-argument callable that only enqueues."""
        T = int(n_steps)
        io, keep = self._policy_io(agent, T, None, add_noise, noise_seed, None)
        ref, w, st, h, fn, check = C.byref(io), C.byref(agent._fw_struct), self._stream(), self.h, self.L.cn_rollout_policy, _abi.check

        def call(_keep=(io, keep)):
            io.counter = agent._fused_calls + 1
            agent._fused_calls += T
            rc = fn(h, w, ref, st)
            if rc:
                check(rc)
        return call

    def bind_step_sequence(self, actions, traj=None):
        """Pre-marshalled step_sequence for a fixed [T, N, 2] action tensor: a zero-argument callable that only enqueues.  traj None:
        outputs in place (every step overwrites self.obs / reward / done / topk_idx); traj = dict(obs [T, N, D], reward [T, N],
        done [T, N] uint8): step t's outputs land in slot t of the caller's trajectory buffers (self.obs is NOT refreshed)."""
        a = actions
        assert a.device == self.device and a.dtype == torch.float32 and a.is_contiguous() and a.shape[1:] == (self.N, 2)
        io = _abi.CnSequenceIO()
        T = int(a.shape[0])
        io.action, io.action_stride, io.n_steps = a.data_ptr(), 2 * self.N, T
        if traj is None:
            io.obs, io.reward, io.done, io.topk_idx = self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(), self.topk_idx.data_ptr()
        else:
            o, r, d = traj["obs"], traj["reward"], traj["done"]
            assert tuple(o.shape) == (T, self.N, self.D) and o.dtype == torch.float32 and o.is_contiguous()
            assert tuple(r.shape) == (T, self.N) and r.dtype == torch.float32 and r.is_contiguous()
            assert tuple(d.shape) == (T, self.N) and d.dtype == torch.uint8 and d.is_contiguous()
            io.obs, io.reward, io.done = o.data_ptr(), r.data_ptr(), d.data_ptr()
            io.obs_stride, io.reward_stride, io.done_stride = self.N * self.D, self.N, self.N
        ref, st, h, fn, check = C.byref(io), self._stream(), self.h, self.L.cn_step_sequence, _abi.check
        keep = (io, a, traj)

        def call(_keep=keep):
            rc = fn(h, ref, st)
            if rc:
                check(rc)
        return call

    def observe_external(self, ranges, odom, step_counter=None, is_reset=False, phase=0):
        """Env.get_state + Env.compute_reward on externally supplied /scan and /odom (Gazebo, a physical robot,
        or a recorded run): ranges [N,R] float64, odom [N,10] float64 = x, y, yaw, v, w, time.time(),
        x, y at the end of the sleep, end_timestep, 0.  Returns (obs, reward, done) device tensors.
        phase: 0 = the whole step / reset flow, or a mask of _abi.CN_PHASE_PRE | CN_PHASE_GET_STATE | CN_PHASE_REWARD
        to run the pieces of Env.step separately (ENV:1208-1209, get_state, compute_reward).  CN_PHASE_REWARD
        without CN_PHASE_GET_STATE reads state[R-1], state[R] from self.obs_f64 and `done` from self.done."""
        rg = self._dev(ranges, torch.float64, (self.N, self.R))
        od = self._dev(odom, torch.float64, (self.N, 10))
        sc = None
        if step_counter is not None:
            sc = self._dev(step_counter, torch.int32)
        io = _abi.CnExternalIO(ranges=rg.data_ptr(), odom=od.data_ptr(), step_counter=sc.data_ptr() if sc is not None else None,
                               obs=self.obs.data_ptr(), obs_f64=self.obs_f64.data_ptr() if self.obs_f64 is not None else None,
                               reward=self.reward.data_ptr(), done=self.done.data_ptr(), topk_idx=self.topk_idx.data_ptr(),
                               is_reset=int(bool(is_reset)), phase=int(phase))
        _abi.check(self.L.cn_observe_external(self.h, C.byref(io), self._stream()))
        self._keep = (rg, od, sc)
        return self.obs, self.reward, self.done

    def device_clock(self, span_us=2000, stream=None, out=None):
        """Enqueue cn_device_clock on `stream` (a torch stream; default: this env's): a one-thread kernel that lives `span_us`
        microseconds and leaves (shader-clock cycles, 100 MHz ticks) of that interval in a 2-element int64 device tensor
        (returned; read it after a synchronisation).  cycles / (ticks / 100) = the clock in MHz under the load of that moment."""
        if out is None:
            out = torch.zeros(2, dtype=torch.int64, device=self.device)
        st = self._stream() if stream is None else C.c_void_p(stream.cuda_stream)
        _abi.check(self.L.cn_device_clock(C.c_void_p(out.data_ptr()), int(span_us), self.device.index, st))
        return out

    def counters(self):
        """[N,14] int32: ego_viol, social_viol, obstacle_present_steps, ep_steps, success, failure, status, n_tracks,
        episodes finished, reset pending, and the last finished episode's ego_viol, social_viol,
        obstacle_present_steps, ep_steps at the moment Env.step returned done."""
        _abi.check(self.L.cn_get_counters(self.h, _ptr(self._counters), self._stream()))
        return self._counters

    def returns(self):
        """(return of the last finished episode, running return) float32 [N] device tensors."""
        _abi.check(self.L.cn_get_returns(self.h, _ptr(self._ret), _ptr(self._run), self._stream()))
        return self._ret, self._run

    STATUS_BITS = {"track_overflow": 1, "ttc_zero": 2, "dt_zero": 4, "conf_overflow": 8}      # include/crowdnav.h CN_ST_*

    def status_counts(self):
        """How many environments carry each bit of the per-env status word (sticky until the handle is re-created): a host read.
        track_overflow / conf_overflow: the env outgrew track_capacity (32 / 64 tracks) or the confirmed-object table -- the
        reference's Python lists are unbounded (its tracker keeps the tracks of earlier episodes and duplicates them at a reset), so
        from that step on the env's risk features are computed from the tracks that fit and no longer equal the reference's;
        ttc_zero / dt_zero: a zero time to collision / a repeated time stamp (the reference divides by zero there too)."""
        st = self.counters()[:, 6]
        return {k: int(((st & b) != 0).sum().item()) for k, b in self.STATUS_BITS.items()}


    def debug_env(self, env=0):
        sd = np.zeros(_abi.CN_SD_COUNT); rp = np.zeros(5 + 4 * self.P)
        tr = np.zeros((_abi.CN_MAX_TRACKS, _abi.CN_TF_COUNT)); si = np.zeros(_abi.CN_SI_COUNT, dtype=np.int32)
        _abi.check(self.L.cn_debug_env(self.h, int(env), sd.ctypes.data, rp.ctypes.data, tr.ctypes.data, si.ctypes.data))
        tr = np.ascontiguousarray(tr.T)   # -> [field, slot]
        n = int(si[_abi.SI["NTRACKS"]])
        T = _abi.TF
        return dict(sd=sd, si=si, robot=rp[:5].copy(), ped_p=rp[5:5 + 2 * self.P].reshape(-1, 2).copy(),
                    ped_v=rp[5 + 2 * self.P:].reshape(-1, 2).copy(), n_tracks=n,
                    track_pose=np.stack([tr[T["PX"], :n], tr[T["PY"], :n]], 1), track_dist=tr[T["DIST"], :n].copy(),
                    track_speed=tr[T["SPEED"], :n].copy(), track_vel=np.stack([tr[T["VX"], :n], tr[T["VY"], :n]], 1),
                    track_t=tr[T["T"], :n].copy(), track_dqlen=tr[T["DQLEN"], :n].astype(np.int32),
                    bb=sd[_abi.SD["BB"]], collision_prob=sd[_abi.SD["CPROB"]], ego_score=sd[_abi.SD["EGO"]],
                    wp=(sd[_abi.SD["WPX"]], sd[_abi.SD["WPY"]]), status=int(si[_abi.SI["STATUS"]]),
                    n_confirmed=int(si[_abi.SI["NCONF"]]), n_entries=int(si[_abi.SI["NENTRIES"]]))

    def snapshot(self):
        n = self.L.cn_snapshot_size(self.h)
        buf = np.zeros(n, dtype=np.uint8)
        _abi.check(self.L.cn_snapshot(self.h, buf.ctypes.data, n))
        return buf

    def restore(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        _abi.check(self.L.cn_restore(self.h, buf.ctypes.data, buf.size))

    def save_snapshot(self, path):
        """The whole state of this handle as an .npz: the header spelled out -- ABI version, every cn_config
        field (env_index_base, seed, layout and mode switches), tracker capacity -- next to the SoA fields of
        cn_snapshot (sd, si, ped_p, ped_v, trk, ped_init, ped_preset, ped_aux) and the raw blob.  load_snapshot() restores
        it (cn_restore refuses a header that does not match this handle); the CPU oracle (test infrastructure) loads the
        same file to continue a GPU run step by step (tools/bisect_divergence.py)."""
        blob = self.snapshot()
        hd, arrs = _abi.split_snapshot(blob)
        cfgd = _abi.config_to_dict(hd.config)
        np.savez_compressed(path, format=np.array("crowdnav-snapshot-1"), abi_version=np.int32(hd.abi_version),
                            track_capacity=np.int32(hd.track_capacity),
                            config_keys=np.array(list(cfgd.keys())), config_vals=np.array([float(v) for v in cfgd.values()]),
                            config_seed=np.uint64(cfgd["seed"]), config_env_index_base=np.int64(cfgd["env_index_base"]),
                            blob=blob, **{k: np.array(v) for k, v in arrs.items()})
        return path

    def load_snapshot(self, path):
        """Restore a state written by save_snapshot.  The SoA arrays of the file are authoritative (they may have been
        edited, e.g. by tools/bisect_divergence.py); the header must match this handle (ABI version and configuration)."""
        z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
        if str(z["format"]) != "crowdnav-snapshot-1":
            raise _abi.CrowdNavError("%s is not a crowdnav snapshot file" % path)
        hd, _ = _abi.split_snapshot(z["blob"])
        self.restore(_abi.join_snapshot(hd, {k: z[k] for k in ("sd", "si", "ped_p", "ped_v", "trk", "ped_init", "ped_preset", "ped_aux")}))


def concurrent_streams(want, device=0, candidates=None):
    """`want` torch streams that really run concurrently with each other on `device`.

    HIP multiplexes streams onto a few hardware queues (4 by default, GPU_MAX_HW_QUEUES) and two streams that
    share a queue serialise; which pool stream lands on which queue is an implementation detail of the runtime
    (tools/queue_probe.py prints the matrix).  So: measure.  Two 64-env probe handles step on candidate pairs;
    a pair is concurrent when both chains together take < 0.75x the same two chains on one stream.  Greedy clique, a few ms.
    Returns a list of `want` streams; if fewer than `want` mutually concurrent ones exist the list is padded by
    cycling through the ones found (groups that share a queue still run correctly, just back to back)."""
    import time
    dev = torch.device("cuda", device)
    if want <= 1:
        return [torch.cuda.Stream(device=dev) for _ in range(max(1, want))], 1
    cands = [torch.cuda.Stream(device=dev) for _ in range(candidates or (3 * want + 4))]
    pa = VecEnv(Config(n_envs=64, ped_cycle_ms=1400), device=device)
    pb = VecEnv(Config(n_envs=64, ped_cycle_ms=1400, env_index_base=64), device=device)
    act = torch.zeros((64, 2), dtype=torch.float32, device=dev)
    pa.stream = pb.stream = cands[0]
    pa.reset(); pb.reset()
    torch.cuda.synchronize(dev)

    def t_chain(sa, sb, k=16):
        pa.stream, pb.stream = sa, sb
        best = float("inf")
        for _ in range(3):            # best of three: host jitter only ever makes a pair look serialised
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(k):
                pa.step(act, auto_reset="next")
                if sb is not None:
                    pb.step(act, auto_reset="next")
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t0)
        return best

    # The yardstick is both probe chains on ONE stream (certainly serialised), re-measured next to every test: the GPU's
    # clock state drifts while this runs (idle -> busy), and a reference taken once at the start, on a cold device, made
    # later pairs look faster than they were -- a serialised pair then passed now and then and cost the groups ~15 %.
    for _ in range(4):
        t_chain(cands[0], cands[0])
    chosen = [cands[0]]
    for c in cands[1:]:
        if len(chosen) >= want:
            break
        ser = t_chain(chosen[0], chosen[0])
        if all(t_chain(ch, c) < 0.75 * ser for ch in chosen):     # concurrent: ~0.5-0.6 of it; sharing a queue: ~1.0
            chosen.append(c)
    pa.close(); pb.close()
    found = len(chosen)
    out = [chosen[i % found] for i in range(want)]
    return out, found


class VecEnvGroups:
    """N environments as G independent groups, each a VecEnv handle with its own HIP stream.

    Every wavefront of one launch walks the same phases (ray cast -> type machine -> tracker -> cone) in step,
    so a single launch alternates between VALU-bound and scalar-bound stretches; launches of different groups
    drift apart and fill each other's idle units (4096 envs: 2 groups step ~20 % faster than one launch,
    DESIGN.md section 6).  Groups are contiguous slices of the global env index, so every env sees exactly
    the trajectory it would see in one VecEnv of N envs.  Outputs are row slices of one [N, ...] allocation.

    There is NO join between groups: `step_group(g, ...)` only enqueues on group g's stream.  A consumer
    either works per group on `streams[g]` (double-buffered sampler: actor of group A overlaps the env step
    of group B) or calls `join()` to make the current stream wait for every group."""

    def __init__(self, cfg=None, groups=2, device=0, streams=None, arbitration=None, **kw):
        """arbitration (default: "oldest_first" for groups > 1, "auto" for one group): the groups' launches overlap on the device
        by design, which is where the hardware's oldest-first issue order is the better pipeline (cn_set_arbitration; 4 groups
        of 1024: 108 M env-steps/s against 102 M with "fair")."""
        if arbitration is None:
            arbitration = "oldest_first" if groups > 1 else "auto"
        cfg = cfg if cfg is not None else Config(**kw)
        assert cfg.n_envs % groups == 0, "n_envs must divide evenly into groups"
        self.cfg, self.G, self.N = cfg, groups, cfg.n_envs
        self.device = torch.device("cuda", device)
        n = self.N // groups
        self.n = n
        if streams is None:
            streams, self.concurrent = concurrent_streams(groups, device)   # streams on distinct hardware queues
        else:
            self.concurrent = None
        assert len(streams) == groups
        D, K, dev = cfg.obs_dim, cfg.k_obstacles, self.device
        self.D, self.K = D, K
        self.obs = torch.zeros((self.N, D), dtype=torch.float32, device=dev)
        self.final_obs = torch.zeros((self.N, D), dtype=torch.float32, device=dev)
        self.reward = torch.zeros(self.N, dtype=torch.float32, device=dev)
        self.done = torch.zeros(self.N, dtype=torch.uint8, device=dev)
        self.topk_idx = torch.full((self.N, K), -1, dtype=torch.int32, device=dev)
        self.envs = []
        for g in range(groups):
            sl = slice(g * n, (g + 1) * n)
            out = dict(obs=self.obs[sl], final_obs=self.final_obs[sl], reward=self.reward[sl], done=self.done[sl],
                       topk_idx=self.topk_idx[sl])
            self.envs.append(VecEnv(self._group_cfg(g), device=device, stream=streams[g], out=out, arbitration=arbitration))
        for e in self.envs:      # the groups' launches overlap: the library sizes its workgroups for the environments in flight together
            _abi.check(e.L.cn_set_group_envs(e.h, self.N))
        self.streams = [e.stream for e in self.envs]

    def _group_cfg(self, g):
        import dataclasses
        n = self.N // self.G
        return dataclasses.replace(self.cfg, n_envs=n, env_index_base=self.cfg.env_index_base + g * n)

    def close(self):
        for e in self.envs:
            e.close()

    def rows(self, g):
        return slice(g * self.n, (g + 1) * self.n)

    def fork(self):
        """Every group stream waits for work already queued on the current stream (e.g. the actions)."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)

    def join(self):
        """The current stream waits for everything queued on the group streams."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def reset(self):
        self.fork()
        for e in self.envs:
            e.reset()
        self.join()
        return self.obs

    def bind_step_all(self, action, auto_reset="next", want_final=False):
        """One pre-marshalled cn_step_multi for all groups: `action` is the [N, 2] device buffer the policy writes in place
        (each group reads its slice).  Returns a zero-argument callable that enqueues one step of every group with ONE
        foreign call; outputs land in each group's obs / reward / done.  No fork / join: callers that need the groups ordered
        against another stream use fork() / join() around their loop."""
        G = self.G
        ios = (_abi.CnStepIO * G)()
        hs = (C.c_void_p * G)()
        sts = (C.c_void_p * G)()
        keep = [action]
        ar = {False: 0, True: 1, None: 0, "same": 1, "next": 2}.get(auto_reset, auto_reset)
        for g, e in enumerate(self.envs):
            a = action[self.rows(g)]
            assert a.device == e.device and a.dtype == torch.float32 and a.is_contiguous()
            keep.append(a)
            ios[g] = _abi.CnStepIO(action=a.data_ptr(), step_counter=None, obs=e.obs.data_ptr(),
                                   final_obs=e.final_obs.data_ptr() if want_final else None,
                                   obs_f64=e.obs_f64.data_ptr() if e.obs_f64 is not None else None,
                                   reward=e.reward.data_ptr(), done=e.done.data_ptr(), topk_idx=e.topk_idx.data_ptr(),
                                   auto_reset=ar, reserved=0)
            hs[g] = e.h.value
            sts[g] = e._stream().value
        fn, check = self.envs[0].L.cn_step_multi, _abi.check

        def call(_keep=(keep, ios, hs, sts)):
            rc = fn(G, hs, ios, sts)
            if rc:
                check(rc)
        return call

    def bind_step_sequence(self, actions, auto_reset="next"):
        """K open-loop steps of every group behind ONE foreign call (cn_step_multi with K x G entries, step-major: the same
        launches in the same order as K calls of bind_step_all's callable, enqueued by a C loop instead of a Python one).
        actions: a sequence of K [N, 2] float32 device tensors (entries may repeat).  Returns a zero-argument callable."""
        G, K = self.G, len(actions)
        n = G * K
        ios = (_abi.CnStepIO * n)()
        hs = (C.c_void_p * n)()
        sts = (C.c_void_p * n)()
        keep = [actions]
        ar = {False: 0, True: 1, None: 0, "same": 1, "next": 2}.get(auto_reset, auto_reset)
        for i, action in enumerate(actions):
            for g, e in enumerate(self.envs):
                a = action[self.rows(g)]
                assert a.device == e.device and a.dtype == torch.float32 and a.is_contiguous()
                keep.append(a)
                j = i * G + g
                ios[j] = _abi.CnStepIO(action=a.data_ptr(), step_counter=None, obs=e.obs.data_ptr(), final_obs=None,
                                       obs_f64=e.obs_f64.data_ptr() if e.obs_f64 is not None else None,
                                       reward=e.reward.data_ptr(), done=e.done.data_ptr(), topk_idx=e.topk_idx.data_ptr(),
                                       auto_reset=ar, reserved=0)
                hs[j] = e.h.value
                sts[j] = e._stream().value
        fn, check = self.envs[0].L.cn_step_multi, _abi.check

        def call(_keep=(keep, ios, hs, sts)):
            rc = fn(n, hs, ios, sts)
            if rc:
                check(rc)
        return call

    def step_group(self, g, action, **kw):
        """Env.step for group g on its own stream; `action` is that group's [n, 2] slice."""
        return self.envs[g].step(action, **kw)

    def step(self, action, **kw):
        """Step every group once (action: [N, 2]); fork/join against the current stream, so this is a drop-in
        for VecEnv.step -- the groups still overlap each other inside the call."""
        self.fork()
        for g, e in enumerate(self.envs):
            e.step(action[self.rows(g)], **kw)
        self.join()
        return self.obs, self.reward, self.done

    def counters(self):
        cs = [e.counters() for e in self.envs]
        self.join()
        return torch.cat(cs, 0)

    def status_counts(self):
        """VecEnv.status_counts summed over the groups."""
        out = {}
        for e in self.envs:
            for k, v in e.status_counts().items():
                out[k] = out.get(k, 0) + v
        return out

    def returns(self):
        """(return of the last finished episode, running return) float32 [N], gathered over the groups."""
        rs = [e.returns() for e in self.envs]
        self.join()
        return torch.cat([r[0] for r in rs]), torch.cat([r[1] for r in rs])

    def set_ped_init(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(self.N, self.cfg.n_peds, 2)
        for g, e in enumerate(self.envs):
            e.set_ped_init(xy[self.rows(g)])

    def set_ped_preset_vel(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.N, self.cfg.n_peds, 2)
        for g, e in enumerate(self.envs):
            e.set_ped_preset_vel(v[self.rows(g)])

    def snapshot(self):
        return [e.snapshot() for e in self.envs]

    def restore(self, bufs):
        for e, b in zip(self.envs, bufs):
            e.restore(b)

    def episodes(self):
        """Total finished episodes over all groups (host int; synchronises)."""
        tot = 0
        for e in self.envs:
            with torch.cuda.stream(e.stream):
                tot += int(e.counters()[:, 8].sum().item())
        return tot


class Env:
    """Drop-in for the reference's `Env` (ENV:42): same constructor, methods, return types."""

    def __init__(self, action_dim=2, max_step=200, cfg=None, device=0, linear_forward_speed=0.5, linear_turn_speed=0.05,
                 angular_speed=0.3, **kw):
        if cfg is None:
            cfg = Config(n_envs=1, max_steps=max_step, **kw)
        assert cfg.n_envs == 1
        self.action_dim = action_dim
        # configs/turtlebot3_world.yaml:2-4 (ENV:85-87): the three discrete actions of mode="discrete"
        self.linear_forward_speed, self.linear_turn_speed, self.angular_speed = linear_forward_speed, linear_turn_speed, angular_speed
        self.max_steps = cfg.max_steps
        self.k_obstacle_count = cfg.k_obstacles
        self._v = VecEnv(cfg, device=device)
        self._v.enable_f64_obs()
        self._act = torch.zeros((1, 2), dtype=torch.float32, device=self._v.device)
        self.done = False
        self._trainer_done_latch = False

    def reset(self):
        self._v.reset()
        torch.cuda.synchronize(self._v.device)
        self.done = False   # the kernel applies TRAIN:116 (`env.done = False`) as part of reset
        self._ext_odom = None
        return self._v.obs_f64[0].cpu().numpy()

    def step(self, action, step_counter, mode="discrete"):
        """Env.step (ENV:1164-1225), same default mode as the reference.  mode="discrete" (ENV:1165-1177, the DQN /
        Q-learning trainers): action 0 / 1 / 2 = forward / turn left / turn right with the speeds of
        configs/turtlebot3_world.yaml:2-4; anything else is the continuous (v, w) pair every trainer on this path
        passes (TRAIN:125)."""
        if mode == "discrete":
            a = int(action)
            if a not in (0, 1, 2):      # the reference leaves linear_speed unbound here (UnboundLocalError)
                raise ValueError("discrete action must be 0 (forward), 1 (left) or 2 (right)")
            action = ((self.linear_forward_speed, 0.0), (self.linear_turn_speed, self.angular_speed),
                      (self.linear_turn_speed, -1 * self.angular_speed))[a]
        self._act[0, 0] = float(action[0]); self._act[0, 1] = float(action[1])
        self._v.step(self._act, step_counter=[int(step_counter)], auto_reset=False)
        torch.cuda.synchronize(self._v.device)
        self.done = bool(self._v.done[0].item())
        self._ext_odom = None
        return self._v.obs_f64[0].cpu().numpy(), float(self._v.reward[0].item()), self.done

    # ---- the two halves of Env.step, callable on their own like the reference's (ENV:1222-1223) -------------
    def odom_callback(self, x, y, yaw, linear_x, angular_z, now=None):
        """What the /odom subscriber stores (ENV:239-243) plus the clock get_state will read: `now`, or -- as the
        reference does (ENV:666, 710 call time.time()) -- the wall clock at the moment get_state runs.
        Until the next step()/reset(), get_state / compute_reward use THIS pose instead of the library simulator's."""
        self._ext_odom = [float(x), float(y), float(yaw), float(linear_x), float(angular_z),
                          float(now) if now is not None else None]

    def append_agent_pose(self, x, y, end_timestep):
        """ENV:1208-1209 inside Env.step: agent_pose_deque.append([round(x, 3), round(y, 3)]); agent_vel_timestep."""
        od = self._odom(); od[6], od[7], od[8] = float(x), float(y), float(end_timestep)
        self._v.observe_external(np.zeros((1, self._v.R)), [od], step_counter=[0], phase=_abi.CN_PHASE_PRE)
        torch.cuda.synchronize(self._v.device)

    def _odom(self):
        d = self._v.debug_env(0)["sd"]
        od = [d[_abi.SD["RX"]], d[_abi.SD["RY"]], d[_abi.SD["RYAW"]], d[_abi.SD["RV"]], d[_abi.SD["RW"]],
              d[_abi.SD["CLOCK"]], 0.0, 0.0, 0.0, 0.0]
        ext = getattr(self, "_ext_odom", None)
        if ext is not None:
            od[:5] = ext[:5]
            # no explicit clock: the wall clock, read now.  (The simulator's own clock only advances inside cn_step; an
            # external flow that kept reading it would hand the tracker dt = 0: infinite speeds, CN_ST_DT_ZERO.)
            od[5] = ext[5] if ext[5] is not None else time.time()
        return od

    def get_state(self, scan, step_counter=0, action=(0, 0)):
        """Env.get_state(scan, step_counter=0, action=[0, 0]) -> (state list, done) (ENV:245-1044): `scan` is a
        LaserScan-like object with `.ranges` (or a sequence of R ranges); the robot pose is the last /odom (the
        library simulator's, or odom_callback's)."""
        rg = np.asarray(getattr(scan, "ranges", scan), dtype=np.float64).reshape(1, self._v.R)
        self._v.observe_external(rg, [self._odom()], step_counter=[int(step_counter)], phase=_abi.CN_PHASE_GET_STATE)
        torch.cuda.synchronize(self._v.device)
        self.done = bool(self._v.done[0].item())
        return list(self._v.obs_f64[0].cpu().numpy()), self.done

    def compute_reward(self, state, *args):
        """Env.compute_reward(state, step_counter, done) -> (reward, done) (ENV:1046-1162).  The other two layouts'
        scripts take (state, done) (ORIG:324, RW:751); both spellings are accepted for them."""
        if len(args) == 2:
            step_counter, done = args
        elif len(args) == 1 and self._v.cfg.obs_layout in (1, 2):
            step_counter, done = 0, args[0]
        else:
            raise TypeError("compute_reward(state, step_counter, done)" +
                            (" or compute_reward(state, done)" if self._v.cfg.obs_layout in (1, 2) else ""))
        self._v.obs_f64[0].copy_(torch.as_tensor(np.asarray(state, dtype=np.float64)))
        self._v.done[0] = int(bool(done))
        self._v.observe_external(np.zeros((1, self._v.R)), [self._odom()], step_counter=[int(step_counter)],
                                 phase=_abi.CN_PHASE_REWARD)
        torch.cuda.synchronize(self._v.device)
        return float(self._v.reward[0].item()), bool(self._v.done[0].item())

    def get_episode_status(self):
        c = self._v.counters()[0].cpu().numpy()
        return bool(c[4]), bool(c[5])

    def get_social_safety_violation_status(self, step):
        c = self._v.counters()[0].cpu().numpy()
        if self._v.cfg.obs_layout == 2:                  # RW:950-954 divides by the step count it is handed
            return 1.0 - ((int(c[1]) * 1.0) / step)
        return 1.0 - ((int(c[1]) * 1.0) / int(c[2]))   # ZeroDivisionError if no obstacle was ever seen (ENV:1272)

    def get_ego_safety_violation_status(self, step):
        c = self._v.counters()[0].cpu().numpy()
        if self._v.cfg.obs_layout == 2:                  # RW:956-960
            return 1.0 - ((int(c[0]) * 1.0) / step)
        return 1.0 - ((int(c[0]) * 1.0) / int(c[2]))

    def shutdown(self):
        pass
