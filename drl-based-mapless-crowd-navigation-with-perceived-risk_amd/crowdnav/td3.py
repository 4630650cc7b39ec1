"""TD3 pieces the rollout needs, in PyTorch-ROCm (the caller of the hot path).

Mirrors td3.py of the reference: Actor 3 x Linear(256) with sigmoid*max_lin_vel / tanh*max_ang_vel heads
(TD3:81-106), Critic (TD3:109-126), Gaussian exploration sigma = 1.0 (TD3:67-78), clipping to
v in [0, 0.22], w in [-2, 2] (TD3:214-215), and the TD3 update (TD3:225-285) with the hyper-parameters of
start_td3_training.py:62-72.  The replay buffer is a device-resident ring so a vectorised env never
leaves the GPU.  No custom kernels here: these are plain library GEMMs (hipBLASLt)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Actor(nn.Module):
    def __init__(self, num_inputs=398, num_actions=2, hidden_size=256, max_lin_vel=0.22, max_ang_vel=2.0):
        super().__init__()
        self.linear1 = nn.Linear(num_inputs, hidden_size)
        self.linear2 = nn.Linear(hidden_size, hidden_size)
        self.linear3 = nn.Linear(hidden_size, num_actions)
        self.max_lin_vel, self.max_ang_vel = max_lin_vel, max_ang_vel

    def logits(self, state):
        x = F.relu(self.linear1(state))
        x = F.relu(self.linear2(x))
        return self.linear3(x)

    def forward(self, state):
        a = self.logits(state)
        return torch.stack([torch.sigmoid(a[:, 0]) * self.max_lin_vel, torch.tanh(a[:, 1]) * self.max_ang_vel], 1)


class Critic(nn.Module):
    def __init__(self, num_inputs=398, num_actions=2, hidden_size=256):
        super().__init__()
        self.linear1 = nn.Linear(num_inputs + num_actions, hidden_size)
        self.linear2 = nn.Linear(hidden_size, hidden_size)
        self.linear3 = nn.Linear(hidden_size, 1)

    def forward(self, state, action):
        x = torch.cat([state, action], 1)
        x = F.relu(self.linear1(x))
        x = F.relu(self.linear2(x))
        return self.linear3(x)


def _device_scalar_view(ptr, device):
    """A 0-d float32 tensor aliasing one device float owned by libcrowdnav (alive as long as its handle)."""
    class _Arr:
        __cuda_array_interface__ = {"shape": (1,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Arr(), device=device).reshape(())


class DeviceReplay:
    """Ring buffer on the device (ReplayBuffer, TD3:19-37, without the Python list).  ONE source of truth for the write position
    and the fill level: the device scalars `pos_dev` / `size_dev`.  Every write -- add() and add_masked() alike -- indexes from
    `pos_dev` on the device, so the two can be mixed freely; rows that are not transitions go to a spare row past the end of the
    ring.  The host only keeps BOUNDS on the fill level (`_lb` <= size_dev <= `_ub`): add() moves both (every row is kept),
    add_masked() moves the upper one only.  `ready(n)` answers "more than n rows?" from the bounds and reads the device scalar
    only while they straddle n -- never again once the ring holds more than a batch -- and `len()` is exact (it reads the device
    when the bounds differ).  On a HIP device a write is libcrowdnav's cn_replay_write (two launches); `fused=False` keeps the
    PyTorch formulation (~15 kernels: cumulative sum, index arithmetic, five index_copy_), which is also what runs on the CPU
    and what the GPU tests compare the fused write against."""

    def __init__(self, capacity, obs_dim, device, fused=True):
        self.cap = int(capacity)
        self.obs_dim = int(obs_dim)
        self.fused = bool(fused) and torch.device(device).type == "cuda"
        self._ring = None
        cap1 = self.cap + 1                       # row `cap`: where add_masked() drops the rows it does not keep
        self.s = torch.zeros((cap1, obs_dim), dtype=torch.float32, device=device)
        self.s2 = torch.zeros((cap1, obs_dim), dtype=torch.float32, device=device)
        self.a = torch.zeros((cap1, 2), dtype=torch.float32, device=device)
        self.r = torch.zeros((cap1, 1), dtype=torch.float32, device=device)
        self.d = torch.zeros((cap1, 1), dtype=torch.float32, device=device)
        self.pos_dev = torch.zeros((), dtype=torch.int64, device=device)
        self.size_dev = torch.zeros((), dtype=torch.int64, device=device)
        self._lb, self._ub = 0, 0
        self._ar = None

    def add(self, s, a, r, s2, d):
        """ReplayBuffer.add for a batch of transitions (all rows kept).  Same device-side path as add_masked()."""
        n = s.shape[0]
        self._write(s, a, r, s2, d, None)
        self._lb, self._ub = min(self.cap, self._lb + n), min(self.cap, self._ub + n)

    def add_masked(self, s, a, r, s2, d, keep):
        """add() for the rows where `keep` (bool [n]) is set -- the others are not transitions (an env's reset launch under the
        next-step reset convention).  No host synchronisation: kept rows go to consecutive ring slots after `pos_dev`, the
        rest to the spare row."""
        self._write(s, a, r, s2, d, keep)
        self._ub = min(self.cap, self._ub + s.shape[0])

    def _write_fused(self, s, a, r, s2, d, keep):
        import ctypes as C
        from . import _abi
        L = _abi.lib()
        n = s.shape[0]
        dev = self.s.device
        if self._ring is None:
            self._ring = _abi.CnReplayRing(s=self.s.data_ptr(), a=self.a.data_ptr(), r=self.r.data_ptr(), s2=self.s2.data_ptr(), d=self.d.data_ptr(),
                                           capacity=self.cap, pos_dev=self.pos_dev.data_ptr(), size_dev=self.size_dev.data_ptr(),
                                           obs_dim=self.obs_dim, reserved=0)
            self._slot = torch.zeros(0, dtype=torch.int32, device=dev)
        if self._slot.numel() < n:
            self._slot = torch.zeros(n, dtype=torch.int32, device=dev)
        f32 = lambda x, shape: x.reshape(shape).to(torch.float32).contiguous()        # (no-ops for what Env.step and Agent.act hand over)
        s, s2, a, r = f32(s, (n, self.obs_dim)), f32(s2, (n, self.obs_dim)), f32(a, (n, 2)), f32(r, (n,))
        d = d.reshape(n)
        d8 = d.contiguous() if d.dtype in (torch.uint8, torch.bool) else (d != 0)
        k8 = None
        if keep is not None:
            k8 = keep.reshape(n).contiguous() if keep.dtype in (torch.uint8, torch.bool) else (keep.reshape(n) != 0)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = L.cn_replay_write(C.byref(self._ring), C.c_void_p(s.data_ptr()), C.c_void_p(a.data_ptr()), C.c_void_p(r.data_ptr()),
                               C.c_void_p(s2.data_ptr()), C.c_void_p(d8.data_ptr()), C.c_void_p(k8.data_ptr() if k8 is not None else None),
                               n, C.c_void_p(self._slot.data_ptr()), dev.index if dev.index is not None else torch.cuda.current_device(), st)
        if rc != 0:
            raise _abi.CrowdNavError("cn_replay_write: %s" % L.cn_td3_last_error().decode())

    def _write(self, s, a, r, s2, d, keep):
        if self.fused:
            return self._write_fused(s, a, r, s2, d, keep)
        n = s.shape[0]
        if keep is None:
            c = torch.arange(1, n + 1, device=s.device)
            idx = (self.pos_dev + c - 1) % self.cap
        else:
            c = torch.cumsum(keep.to(torch.int64), 0)
            idx = torch.where(keep, (self.pos_dev + c - 1) % self.cap, torch.full_like(c, self.cap))
        self.s.index_copy_(0, idx, s); self.a.index_copy_(0, idx, a); self.s2.index_copy_(0, idx, s2)
        self.r.index_copy_(0, idx, r.reshape(n, 1).float()); self.d.index_copy_(0, idx, d.reshape(n, 1).float())
        tot = c[-1]
        self.pos_dev.copy_((self.pos_dev + tot) % self.cap)
        self.size_dev.copy_(torch.clamp(self.size_dev + tot, max=self.cap))

    def sync_len(self):
        """The exact fill level (one host read); collapses the host-side bounds onto it."""
        self._lb = self._ub = int(self.size_dev.item())
        return self._lb

    def ready(self, n):
        """len() > n, without a host read whenever the bounds already decide it."""
        if self._lb > n:
            return True
        if self._ub <= n:
            return False
        return self.sync_len() > n

    @property
    def size(self):
        return len(self)

    @property
    def pos(self):
        return int(self.pos_dev.item())

    def sample(self, batch):
        """Uniform sample of the filled part; the indices are drawn on the device from the device-side fill level."""
        u = torch.rand(batch, device=self.s.device)
        idx = (u * self.size_dev.clamp(min=1).to(torch.float32)).long().clamp_(max=self.cap - 1)
        idx = torch.minimum(idx, (self.size_dev - 1).clamp(min=0))
        return self.s[idx], self.a[idx], self.r[idx], self.s2[idx], self.d[idx]

    def __len__(self):
        return self._lb if self._lb == self._ub else self.sync_len()


class Agent:
    """TD3 agent (TD3:129-319) acting on batches of observations that stay on the device."""

    def __init__(self, obs_dim=398, hidden=256, actor_lr=3e-4, critic_lr=3e-4, batch_size=128, memory_size=1_000_000,
                 gamma=0.99, tau=0.005, max_v=0.22, max_w=2.0, noise_std=0.2, noise_clip=0.5, policy_delay=2,
                 explore_sigma=1.0, device="cuda", seed=0, actor_final_init=None):
        self.device = torch.device(device)
        g = torch.Generator().manual_seed(seed)
        torch.manual_seed(seed)
        self.actor = Actor(obs_dim, 2, hidden, max_v, max_w).to(self.device)
        self.actor_t = Actor(obs_dim, 2, hidden, max_v, max_w).to(self.device)
        self.q1, self.q2 = Critic(obs_dim, 2, hidden).to(self.device), Critic(obs_dim, 2, hidden).to(self.device)
        self.q1_t, self.q2_t = Critic(obs_dim, 2, hidden).to(self.device), Critic(obs_dim, 2, hidden).to(self.device)
        if actor_final_init:
            # NOT the reference (td3.py:81-95 keeps nn.Linear's default U(+-1/sqrt(256)) everywhere): the DDPG paper's small uniform
            # initialisation of the actor's output layer, an opt-in for the seed sensitivity documented in profiles/r04/train/ (heads
            # that start near the middle of the sigmoid / tanh instead of wherever the default range leaves them)
            with torch.no_grad():
                self.actor.linear3.weight.uniform_(-float(actor_final_init), float(actor_final_init))
                self.actor.linear3.bias.uniform_(-float(actor_final_init), float(actor_final_init))
        for t, s in ((self.actor_t, self.actor), (self.q1_t, self.q1), (self.q2_t, self.q2)):
            t.load_state_dict(s.state_dict())
        fused = self.device.type == "cuda"        # one kernel per optimizer step instead of ~10 per parameter tensor;
        kw = dict(fused=True) if fused else {}
        self.opt_a = torch.optim.Adam(self.actor.parameters(), lr=actor_lr, **kw)
        self.opt_q1 = torch.optim.Adam(self.q1.parameters(), lr=critic_lr, **kw)
        self.opt_q2 = torch.optim.Adam(self.q2.parameters(), lr=critic_lr, **kw)
        self.memory = DeviceReplay(memory_size, obs_dim, self.device)
        self.batch_size, self.gamma, self.tau = batch_size, gamma, tau
        self.max_v, self.max_w = max_v, max_w
        self.noise_std, self.noise_clip, self.policy_delay = noise_std, noise_clip, policy_delay
        self.explore_sigma = explore_sigma
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        # exploration-noise key of the fused actor paths (cn_policy_tail / cn_actor_forward): derived from the Agent seed so that
        # --seed changes the noise; the call counter is part of checkpoints' bookkeeping (noise_state) so a resumed run
        # does not replay the stream
        self._noise_seed = (0x9E3779B97F4A7C15 * (int(seed) + 1) ^ 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        self._fused_calls = 0
        self._dev_index = self.device.index if self.device.type == "cuda" and self.device.index is not None else (
            torch.cuda.current_device() if self.device.type == "cuda" else -1)
        self._lo = torch.tensor([0.0, -max_w], device=self.device)
        self._hi = torch.tensor([max_v, max_w], device=self.device)
        del g

    @torch.no_grad()
    def act(self, obs, add_noise=True):
        """Agent.act (TD3:196-223) for a batch: actor, Gaussian noise sigma=1.0, clip."""
        a = self.actor(obs)
        if add_noise:
            a = a + torch.randn(a.shape, generator=self.gen, device=self.device) * self.explore_sigma
        return torch.max(torch.min(a, self._hi), self._lo).contiguous()

    @torch.no_grad()
    def act_fused(self, obs, out=None, add_noise=True):
        """Agent.act with the output stage (heads + exploration noise + clip) as ONE kernel of libcrowdnav
        (cn_policy_tail) instead of ~10 elementwise launches.  Same distribution as act(); the noise comes from
        a counter-based generator keyed by (seed, call counter, row)."""
        import ctypes as C
        from . import _abi
        lg = self.actor.logits(obs).contiguous()
        if out is None:
            out = torch.empty_like(lg)
        self._fused_calls += 1
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _abi.check(_abi.lib().cn_policy_tail(C.c_void_p(lg.data_ptr()), C.c_void_p(out.data_ptr()), lg.shape[0],
                                             self.max_v, self.max_w, self.explore_sigma if add_noise else 0.0,
                                             self._noise_seed, self._fused_calls, self._dev_index, st))
        return out

    def noise_state(self):
        """(seed, call counter) of the fused exploration noise -- persist it with a checkpoint and hand it back to
        set_noise_state() on resume so the noise stream continues instead of restarting."""
        return self._noise_seed, self._fused_calls

    def set_noise_state(self, seed, calls):
        self._noise_seed, self._fused_calls = int(seed), int(calls)

    def sync_fused_weights(self):
        """(Re)build the packed float32 copies cn_actor_forward / cn_rollout_policy read (cn_actor_pack_weights); call after the
        actor's weights change.  The packed buffers and the cn_actor_weights struct are allocated once and refreshed IN PLACE
        (two small copies + two pack kernels on the current stream), so pre-marshalled calls (bind_act_mfma,
        VecEnv.bind_rollout_policy) keep reading the current weights.  The refresh is ordered on torch's current stream: a caller
        that runs the actor on other streams (VecEnvGroups) orders this call after them (join) and their next launches after it (fork)."""
        import ctypes as C
        from . import _abi
        a = self.actor
        D = a.linear1.in_features
        Dp = (D + 31) // 32 * 32          # zero rows up to the packed layout's block of 32 inputs
        with torch.no_grad():
            st8 = getattr(self, "_fw_stage", None)
            if st8 is None or st8[0].shape != (Dp, 256):
                w1t = torch.zeros((Dp, 256), dtype=torch.float32, device=self.device)      # rows D..Dp stay zero
                w2t = torch.empty((256, 256), dtype=torch.float32, device=self.device)
                self._fw_stage = (w1t, w2t, torch.empty_like(w1t), torch.empty_like(w2t))
            w1t, w2t, w1p, w2p = self._fw_stage
            w1t[:D].copy_(a.linear1.weight.detach().t())
            w2t.copy_(a.linear2.weight.detach().t())
            L = _abi.lib()
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            # K-major -> the order the kernel's wavefronts consume (16-byte loads, 4 KB contiguous per wavefront and block)
            _abi.check(L.cn_actor_pack_weights(C.c_void_p(w1t.data_ptr()), Dp, C.c_void_p(w1p.data_ptr()), self._dev_index, st))
            _abi.check(L.cn_actor_pack_weights(C.c_void_p(w2t.data_ptr()), 256, C.c_void_p(w2p.data_ptr()), self._dev_index, st))
            # biases and linear3 are read where they live (float32, contiguous nn.Linear storages: .float().contiguous() is the tensor itself)
            self._fw = dict(w1p=w1p, b1=a.linear1.bias.detach().float().contiguous(),
                            w2p=w2p, b2=a.linear2.bias.detach().float().contiguous(),
                            w3=a.linear3.weight.detach().float().contiguous(), b3=a.linear3.bias.detach().float().contiguous())
        f = self._fw
        vals = dict(w1p=f["w1p"].data_ptr(), b1=f["b1"].data_ptr(), w2p=f["w2p"].data_ptr(), b2=f["b2"].data_ptr(), w3=f["w3"].data_ptr(),
                    b3=f["b3"].data_ptr(), obs_dim=D, obs_dim_padded=Dp, hidden=256, reserved=0)
        if hasattr(self, "_fw_struct"):
            for k, v in vals.items():
                setattr(self._fw_struct, k, v)          # same object: byref()s taken earlier stay valid
        else:
            self._fw_struct = _abi.CnActorWeights(**vals)

    @torch.no_grad()
    def act_mfma(self, obs, out=None, add_noise=True, stream=None, noise_seed=None):
        """Agent.act as ONE kernel (cn_actor_forward): the three Linear layers on the f32 matrix cores with the
        activations in LDS, plus heads, exploration noise and clip.  float32 throughout, like the reference.
        stream: torch stream to enqueue on (default: current).  The exploration noise is keyed by
        (noise_seed, call counter, row); noise_seed defaults to the Agent's own key, callers that split a batch over
        several calls pass `agent.group_noise_seed(g)` so that the groups draw different noise."""
        import ctypes as C
        from . import _abi
        if not hasattr(self, "_fw_struct"):
            self.sync_fused_weights()
        obs = obs.contiguous()
        if out is None:
            out = torch.empty((obs.shape[0], 2), dtype=torch.float32, device=self.device)
        self._fused_calls += 1
        st = C.c_void_p((stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream)
        seed = self._noise_seed if noise_seed is None else int(noise_seed)
        _abi.check(_abi.lib().cn_actor_forward(C.byref(self._fw_struct), C.c_void_p(obs.data_ptr()), C.c_void_p(out.data_ptr()),
                                               obs.shape[0], self.max_v, self.max_w,
                                               self.explore_sigma if add_noise else 0.0, seed, self._fused_calls,
                                               self._dev_index, st))
        return out

    def group_noise_seed(self, g):
        """Noise key of stream group / rank `g` (distinct per group, derived from the Agent seed)."""
        return (self._noise_seed ^ (0xA0761D6478BD642F * (int(g) + 1))) & 0xFFFFFFFFFFFFFFFF

    def bind_act_mfma(self, obs, out, add_noise=True, stream=None, noise_seed=None):
        """Pre-marshalled act_mfma for fixed obs/out buffers: a zero-argument callable that only enqueues."""
        import ctypes as C
        from . import _abi
        if not hasattr(self, "_fw_struct"):
            self.sync_fused_weights()
        assert obs.is_contiguous() and out.is_contiguous()
        fn, check = _abi.lib().cn_actor_forward, _abi.check
        w = C.byref(self._fw_struct)
        po, pa, n = C.c_void_p(obs.data_ptr()), C.c_void_p(out.data_ptr()), obs.shape[0]
        st = C.c_void_p((stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream)
        sigma = self.explore_sigma if add_noise else 0.0
        mv, mw, seed = self.max_v, self.max_w, self._noise_seed if noise_seed is None else int(noise_seed)
        keep = (obs, out, self._fw_struct, self._fw)
        dv = self._dev_index

        def call(_keep=keep):
            self._fused_calls = c = self._fused_calls + 1
            rc = fn(w, po, pa, n, mv, mw, sigma, seed, c, dv, st)
            if rc:
                check(rc)
        return call

    def _update(self, s, a, r, s2, d, target_noise, do_actor):
        """The arithmetic of one TD3 update (TD3:225-285) on a given batch."""
        with torch.no_grad():
            noise = (target_noise * self.noise_std).clamp(-self.noise_clip, self.noise_clip)
            a2 = self.actor_t(s2) + noise                       # not re-clipped to the action bounds (TD3:244-247)
            q_t = torch.min(self.q1_t(s2, a2), self.q2_t(s2, a2))
            y = r + (1.0 - d) * self.gamma * q_t
        l1 = F.mse_loss(self.q1(s, a), y)
        l2 = F.mse_loss(self.q2(s, a), y)
        self.opt_q1.zero_grad(set_to_none=True); l1.backward(); self.opt_q1.step()
        self.opt_q2.zero_grad(set_to_none=True); l2.backward(); self.opt_q2.step()
        if do_actor:
            la = -self.q1(s, self.actor(s)).mean()
            self.opt_a.zero_grad(set_to_none=True); la.backward(); self.opt_a.step()
            with torch.no_grad():
                for t, src in ((self.q1_t, self.q1), (self.q2_t, self.q2), (self.actor_t, self.actor)):
                    pt, ps = list(t.parameters()), list(src.parameters())
                    torch._foreach_mul_(pt, 1.0 - self.tau)                      # TD3:287-299: target*(1-tau) + local*tau
                    torch._foreach_add_(pt, torch._foreach_mul(ps, self.tau))
        return l1.detach()

    # ---- the same update as ONE hipGraph launch (a TD3 update is ~150 small kernels: at batch 128 the eager path is bound by
    # launch overhead, 1.5 ms per update on an MI355X) ------------------------------------------------------------------------
    def enable_graphs(self):
        """Capture the update (replay sampling + target noise + _update) into two hipGraphs -- critics only, and critics +
        actor + soft updates -- which learn() then replays when it is called without an explicit batch.  The arithmetic is
        _update's; what differs from the eager path: the optimizers are rebuilt with capturable=True (step counters on the
        device), and the replay indices and the target noise come from torch's default CUDA generator inside the graph
        (seeded by Agent(seed=...)) instead of torch.randint / the Agent's own generator.  Parameters and optimizer state are
        left exactly as they were (the capture's warm-up runs on a copy of them)."""
        if self.device.type != "cuda":
            raise RuntimeError("enable_graphs needs a HIP device")
        if getattr(self, "_graphs", None):
            return
        B, dev = self.batch_size, self.device
        nets = (self.actor, self.actor_t, self.q1, self.q1_t, self.q2, self.q2_t)
        saved = [[p.detach().clone() for p in m.parameters()] for m in nets]
        old_state = [o.state_dict() for o in (self.opt_a, self.opt_q1, self.opt_q2)]
        lr = [o.param_groups[0]["lr"] for o in (self.opt_a, self.opt_q1, self.opt_q2)]
        self.opt_a = torch.optim.Adam(self.actor.parameters(), lr=lr[0], fused=True, capturable=True)
        self.opt_q1 = torch.optim.Adam(self.q1.parameters(), lr=lr[1], fused=True, capturable=True)
        self.opt_q2 = torch.optim.Adam(self.q2.parameters(), lr=lr[2], fused=True, capturable=True)
        mem = self.memory

        def one(do_actor):
            # indices from the DEVICE-side fill level (DeviceReplay.size_dev): no host value is frozen into the graph, and an
            # index can never reach a row that was not written (float32 rounding of u * size is clamped to size - 1)
            u = torch.rand(B, device=dev)
            idx = torch.minimum((u * mem.size_dev.to(torch.float32)).long(), (mem.size_dev - 1).clamp(min=0))
            noise = torch.randn((B, 2), device=dev)
            return self._update(mem.s[idx], mem.a[idx], mem.r[idx], mem.s2[idx], mem.d[idx], noise, do_actor)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):                       # warm-up: allocator, optimizer state, autograd graphs of both shapes
                one(False); one(True)
        torch.cuda.current_stream(dev).wait_stream(side)
        self._graphs, self._g_loss = {}, {}
        for do_actor in (False, True):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._g_loss[do_actor] = one(do_actor)
            self._graphs[do_actor] = g
        # back to the state enable_graphs() was called in: parameters, and the optimizers' moments / step counters
        with torch.no_grad():
            for m, ps in zip(nets, saved):
                for p_, q_ in zip(m.parameters(), ps):
                    p_.copy_(q_)
            for o, st in zip((self.opt_a, self.opt_q1, self.opt_q2), old_state):
                for i, p_ in enumerate(o.param_groups[0]["params"]):
                    cur = o.state[p_]
                    prev = st["state"].get(i)
                    if prev is None:
                        cur["exp_avg"].zero_(); cur["exp_avg_sq"].zero_(); cur["step"].zero_()
                    else:
                        cur["exp_avg"].copy_(prev["exp_avg"]); cur["exp_avg_sq"].copy_(prev["exp_avg_sq"])
                        cur["step"].fill_(float(prev["step"]))
        torch.cuda.synchronize(dev)

    # ---- the same update as 7 (+ 5) hand-written launches: libcrowdnav's cn_td3_update (csrc/crowdnav_td3.hip) ----------------------
    def enable_fused_update(self):
        """Hand the update to cn_td3_update: forward / backward GEMMs of the six networks on the f32 matrix cores, weight gradients
        and soft updates folded into the Adam step, TD target / heads inside the GEMMs, replay indices and target noise
        drawn on the device -- 7 launches for the critic step, 5 more with the actor and the targets, against ~150 through PyTorch.  The
        networks stay these nn.Modules (the kernels step their parameter storages in place); Adam's moments restart from zero
        inside the library (torch.optim state is not carried over), so call this before training, not in the middle of it."""
        import ctypes as C
        from . import _abi
        if self.device.type != "cuda":
            raise RuntimeError("enable_fused_update needs a HIP device")
        if getattr(self, "_td3_h", None):
            return
        L = _abi.lib()

        def mlp(m):
            ps = [m.linear1.weight, m.linear1.bias, m.linear2.weight, m.linear2.bias, m.linear3.weight, m.linear3.bias]
            assert all(p.is_contiguous() and p.dtype == torch.float32 and p.is_cuda for p in ps)
            return _abi.CnTd3Mlp(*[p.data_ptr() for p in ps])
        mem = self.memory
        og = self.opt_a.param_groups[0]
        cfg = _abi.CnTd3Config(obs_dim=self.actor.linear1.in_features, hidden=self.actor.linear1.out_features, batch=self.batch_size,
                               policy_delay=self.policy_delay, gamma=self.gamma, tau=self.tau, lr_actor=og["lr"],
                               lr_critic=self.opt_q1.param_groups[0]["lr"], beta1=og["betas"][0], beta2=og["betas"][1], eps=og["eps"],
                               noise_std=self.noise_std, noise_clip=self.noise_clip, max_v=self.max_v, max_w=self.max_w, reserved=0.0,
                               actor=mlp(self.actor), actor_t=mlp(self.actor_t), q1=mlp(self.q1), q1_t=mlp(self.q1_t), q2=mlp(self.q2),
                               q2_t=mlp(self.q2_t), replay_s=mem.s.data_ptr(), replay_a=mem.a.data_ptr(), replay_r=mem.r.data_ptr(),
                               replay_s2=mem.s2.data_ptr(), replay_d=mem.d.data_ptr(), replay_size_dev=mem.size_dev.data_ptr(),
                               seed=self._noise_seed)
        h = C.c_void_p()
        rc = L.cn_td3_create(C.byref(cfg), self._dev_index, C.byref(h))
        if rc != 0:
            raise _abi.CrowdNavError("cn_td3_create: %s" % L.cn_td3_last_error().decode())
        self._td3_h, self._td3_cfg = h, cfg
        import numpy as np  # noqa: F401
        self._td3_loss = None

    def _fused_learn(self, step, batch=None, target_noise=None):
        import ctypes as C
        from . import _abi
        L = _abi.lib()
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        bp = None
        if batch is not None:
            s, a, r, s2, d = [t.contiguous().float() for t in batch]
            tn = target_noise.contiguous().float() if target_noise is not None else None
            # the kernels read exactly cn_td3_config.batch rows of every array (fixed at enable_fused_update)
            B, D = self.batch_size, self.actor.linear1.in_features
            if s.shape != (B, D) or s2.shape != (B, D) or a.shape != (B, 2) or r.numel() != B or d.numel() != B or (
                    tn is not None and tn.shape != (B, 2)):
                raise ValueError("cn_td3_update was created for batches of %d x %d; got s %s a %s r %s s2 %s d %s noise %s" % (
                    B, D, tuple(s.shape), tuple(a.shape), tuple(r.shape), tuple(s2.shape), tuple(d.shape),
                    None if tn is None else tuple(tn.shape)))
            self._td3_keep = (s, a, r, s2, d, tn)            # alive until the next call: the launches are asynchronous
            bp = C.byref(_abi.CnTd3Batch(s.data_ptr(), a.data_ptr(), r.data_ptr(), s2.data_ptr(), d.data_ptr(),
                                         tn.data_ptr() if tn is not None else None))
        rc = L.cn_td3_update(self._td3_h, int(step % self.policy_delay == 0), bp, st)
        if rc != 0:
            raise _abi.CrowdNavError("cn_td3_update: %s" % L.cn_td3_last_error().decode())
        # the first critic's loss of this update, where the kernels left it: a fresh 0-d tensor per call, the same contract as the
        # PyTorch learner (no host synchronisation: one 4-byte device-to-device copy on the update's stream).  The view of the
        # handle's device scalar itself is overwritten by the next update and dies with cn_td3_destroy, so it is not handed out.
        if self._td3_loss is None:
            ptr = L.cn_td3_loss_dev(self._td3_h)
            self._td3_loss = _device_scalar_view(ptr, self.device) if ptr else False
        return self._td3_loss.clone() if self._td3_loss is not False else None

    def __del__(self):
        try:
            if getattr(self, "_td3_h", None):
                from . import _abi
                _abi.lib().cn_td3_destroy(self._td3_h)
                self._td3_h = None
        except Exception:
            pass

    def learn(self, step, batch=None, target_noise=None):
        """One TD3 update (TD3:225-285): clipped target-policy noise added to the target actor's action (the
        reference does not re-clip the noisy action to the action bounds, TD3:244-247), min of the two target
        critics, MSE critic losses with one Adam step each, and every `policy_delay` steps the actor step plus
        the three soft updates.  `batch` = (s, a, r[B,1], s2, d[B,1]) and `target_noise` [B,2] (before the clip)
        override the replay sample / the generator -- used by the parity test against the reference's update.
        Returns the first critic's loss as a 0-d tensor (no host synchronisation)."""
        if getattr(self, "_td3_h", None):
            if batch is None and not self.memory.ready(self.batch_size):
                return None
            return self._fused_learn(step, batch, target_noise)
        if batch is None:
            if not self.memory.ready(self.batch_size):
                return None
            if getattr(self, "_graphs", None) and target_noise is None:
                do_actor = step % self.policy_delay == 0
                self._graphs[do_actor].replay()
                return self._g_loss[do_actor]
            batch = self.memory.sample(self.batch_size)
        s, a, r, s2, d = batch
        if target_noise is None:
            target_noise = torch.randn(a.shape, generator=self.gen, device=self.device)
        return self._update(s, a, r, s2, d, target_noise, step % self.policy_delay == 0)

    def load_models(self, actor_path, critic1_path, critic2_path):
        """Agent.load_models (TD3:313-319): the reference's checkpoints are plain state_dicts with the same
        parameter names (linear1/2/3), so its published .pt files load unchanged; targets are hard-copied."""
        self.actor.load_state_dict(torch.load(actor_path, map_location=self.device))
        self.q1.load_state_dict(torch.load(critic1_path, map_location=self.device))
        self.q2.load_state_dict(torch.load(critic2_path, map_location=self.device))
        for t, src in ((self.actor_t, self.actor), (self.q1_t, self.q1), (self.q2_t, self.q2)):
            t.load_state_dict(src.state_dict())
        if hasattr(self, "_fw_struct"):
            self.sync_fused_weights()

    def save(self, outdir, ep):
        """Target-network checkpoints every 100 episodes (TRAIN:150-154, TD3:304-311)."""
        import os
        os.makedirs(outdir, exist_ok=True)
        torch.save(self.actor_t.state_dict(), os.path.join(outdir, "td3_actor_model_ep%d.pt" % ep))
        torch.save(self.q1_t.state_dict(), os.path.join(outdir, "td3_critic1_model_ep%d.pt" % ep))
        torch.save(self.q2_t.state_dict(), os.path.join(outdir, "td3_critic2_model_ep%d.pt" % ep))
