"""The caller of the hot path: crowdnav.td3 against golden vectors produced by the
reference's own TD3 classes (oracle/make_goldens_td3.py imports turtlebot3_rl_sim/src/td3.py unmodified).
CPU-only: these are plain PyTorch modules."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "td3.npz"))


def _agent(**kw):
    from crowdnav.td3 import Agent
    return Agent(device="cpu", memory_size=64, **kw)


def test_actor_forward_matches_reference_actor():
    """Same seed -> same nn.Linear initialisation order (linear1, linear2, linear3) -> same weights; the heads are
    sigmoid * 0.22 and tanh * 2.0 (TD3:96-106)."""
    from crowdnav.td3 import Actor
    torch.manual_seed(int(G["actor_seed"]))
    actor = Actor(398, 2, 256, 0.22, 2.0)
    with torch.no_grad():
        out = actor(torch.from_numpy(G["actor_obs"])).numpy()
    np.testing.assert_allclose(out, G["actor_out"], rtol=0, atol=1e-7)
    # Agent.act without noise: clip to v in [0, 0.22], w in [-2, 2] (TD3:214-215)
    ag = _agent(obs_dim=398)
    ag.actor.load_state_dict(actor.state_dict())
    a = ag.act(torch.from_numpy(G["actor_obs"][:8]), add_noise=False).numpy()
    np.testing.assert_allclose(a, G["act_single"], rtol=0, atol=1e-7)
    assert (a[:, 0] >= 0).all() and (a[:, 0] <= 0.22).all() and (np.abs(a[:, 1]) <= 2.0).all()


def _load(agent, prefix):
    nets = dict(actor=agent.actor, actor_t=agent.actor_t, q1=agent.q1, q1_t=agent.q1_t, q2=agent.q2, q2_t=agent.q2_t)
    for k, m in nets.items():
        sd = {n: torch.from_numpy(G["%s.%s.%s" % (prefix, k, n)]) for n in m.state_dict()}
        m.load_state_dict(sd)
    return nets


def test_four_updates_match_reference_learn():
    """Agent.learn (TD3:225-285) with the replay order and target-policy noise pinned: critic targets without
    re-clipping the noisy action, two Adam critic steps, delayed actor step and soft updates (tau = 0.005)."""
    ag = _agent(obs_dim=46, hidden=32, batch_size=16)
    nets = _load(ag, "init")
    batch = (torch.from_numpy(G["upd_s"]), torch.from_numpy(G["upd_a"]), torch.from_numpy(G["upd_r"])[:, None],
             torch.from_numpy(G["upd_s2"]), torch.from_numpy(G["upd_d"])[:, None])
    for step in range(4):
        ag.learn(step, batch=batch, target_noise=torch.from_numpy(G["upd_noise"][step]))
        for k, m in nets.items():
            for n, v in m.state_dict().items():
                np.testing.assert_allclose(v.numpy(), G["step%d.%s.%s" % (step, k, n)], rtol=2e-5, atol=2e-7,
                                           err_msg="step %d %s.%s" % (step, k, n))
    # step 1 and 3 are not policy steps: targets unchanged between step 0 and 1
    assert np.array_equal(G["step0.actor_t.linear1.weight"], G["step1.actor_t.linear1.weight"])


def test_hyper_parameters_are_the_reference_defaults():
    """start_td3_training.py:62-72 and td3.py:67-78: batch 128, buffer 1e6, hidden 256, noise 0.2 / clip 0.5,
    policy update every 2 steps, exploration sigma 1.0; configs/*.yaml: lr 3e-4... (actor_alpha / critic_alpha), gamma
    0.99, tau 0.005."""
    import inspect
    from crowdnav.td3 import Agent
    d = {k: v.default for k, v in inspect.signature(Agent.__init__).parameters.items() if v.default is not inspect._empty}
    assert d["batch_size"] == 128 and d["memory_size"] == 1_000_000 and d["hidden"] == 256
    assert d["noise_std"] == 0.2 and d["noise_clip"] == 0.5 and d["policy_delay"] == 2 and d["explore_sigma"] == 1.0
    assert d["gamma"] == 0.99 and d["tau"] == 0.005 and d["max_v"] == 0.22 and d["max_w"] == 2.0


def test_checkpoint_round_trip_uses_reference_file_names(tmp_path):
    """TRAIN:150-154 / TD3:304-319: target networks saved as td3_{actor,critic1,critic2}_model_ep<N>.pt state_dicts
    with linear1/2/3 parameter names; load_models restores locals and hard-copies the targets."""
    ag = _agent(obs_dim=46, hidden=32)
    ag.save(str(tmp_path), 100)
    names = sorted(os.listdir(tmp_path))
    assert names == ["td3_actor_model_ep100.pt", "td3_critic1_model_ep100.pt", "td3_critic2_model_ep100.pt"]
    sd = torch.load(os.path.join(tmp_path, names[0]))
    assert list(sd.keys()) == ["linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "linear3.weight", "linear3.bias"]
    other = _agent(obs_dim=46, hidden=32, seed=9)
    other.load_models(*[os.path.join(tmp_path, n) for n in names])
    for a, b in ((ag.actor_t, other.actor), (ag.actor_t, other.actor_t), (ag.q1_t, other.q1), (ag.q2_t, other.q2_t)):
        for x, y in zip(a.parameters(), b.parameters()):
            assert torch.equal(x, y)


@pytest.mark.skipif(not os.path.isdir("/root/reference/turtlebot3_rl_sim/src/models/td3"), reason="reference tree not mounted")
def test_published_reference_checkpoints_load_unchanged():
    """Every 398-input TD3 checkpoint shipped with the reference loads with strict key matching."""
    import glob
    from crowdnav.td3 import Actor
    n = 0
    for f in glob.glob("/root/reference/turtlebot3_rl_sim/src/models/td3/**/td3_actor_model_ep*.pt", recursive=True):
        sd = torch.load(f, map_location="cpu")
        d_in = sd["linear1.weight"].shape[1]
        Actor(d_in, 2, 256).load_state_dict(sd, strict=True)
        n += 1
    assert n > 0


def test_device_replay_ring():
    from crowdnav.td3 import DeviceReplay
    m = DeviceReplay(8, 3, "cpu")
    for i in range(3):
        s = torch.full((4, 3), float(i)); m.add(s, torch.zeros(4, 2), torch.full((4,), float(i)), s + 1, torch.zeros(4, dtype=torch.uint8))
    assert len(m) == 8 and m.pos == 4
    assert m.s[:4, 0].tolist() == [2.0] * 4 and m.s[4:8, 0].tolist() == [1.0] * 4     # oldest rows overwritten (row 8 = the spare row)
    s, a, r, s2, d = m.sample(5)
    assert s.shape == (5, 3) and r.shape == (5, 1) and torch.equal(s2, s + 1)
    # masked add without a host read: kept rows go to consecutive slots after the device-side position, the others to the spare row
    m2 = DeviceReplay(8, 3, "cpu")
    for i in range(3):
        s = torch.full((4, 3), float(10 + i))
        m2.add_masked(s, torch.zeros(4, 2), torch.zeros(4), s + 1, torch.zeros(4), torch.tensor([True, False, True, True]))
    assert m2.sync_len() == 8 and m2.pos == 1                                           # 9 kept rows wrapped once
    assert m2.s[:8, 0].tolist() == [12.0, 10.0, 10.0, 11.0, 11.0, 11.0, 12.0, 12.0]
    assert bool((m2.sample(64)[0][:, 0] >= 10.0).all())


def test_device_replay_mixes_masked_and_plain_adds_and_gates_learning_without_a_sync_len():
    """ONE source of truth for the ring's position and fill level (the device scalars).  add() after add_masked()
    continues where the masked add stopped (it used to index from a stale host position and rewind the device counters), and
    the learn() gate -- DeviceReplay.ready(batch) -- turns true after masked adds alone (collect_policy -> learn with no
    sync_len in between used to return None forever)."""
    from crowdnav.td3 import DeviceReplay, Agent
    m = DeviceReplay(16, 2, "cpu")
    row = lambda v, n: (torch.full((n, 2), float(v)), torch.zeros(n, 2), torch.full((n,), float(v)), torch.full((n, 2), float(v) + .5), torch.zeros(n))
    m.add_masked(*row(1, 4), torch.tensor([True, False, True, False]))        # rows 0, 1
    m.add(*row(2, 3))                                                         # rows 2, 3, 4 -- not 0, 1, 2
    m.add_masked(*row(3, 2), torch.tensor([False, True]))                     # row 5
    m.add(*row(4, 1))                                                         # row 6
    assert m.s[:7, 0].tolist() == [1, 1, 2, 2, 2, 3, 4] and int(m.pos_dev) == 7 and int(m.size_dev) == 7
    assert len(m) == 7 and m.pos == 7 and m.size == 7
    # the bounds: exact after add(), an upper bound after add_masked(); ready() only reads the device while they straddle n
    m = DeviceReplay(64, 2, "cpu")
    assert not m.ready(4) and (m._lb, m._ub) == (0, 0)
    m.add_masked(*row(1, 4), torch.tensor([True, True, False, False]))
    assert (m._lb, m._ub) == (0, 4) and not m.ready(4) and (m._lb, m._ub) == (0, 4)     # ub <= n: decided without a read
    m.add_masked(*row(1, 4), torch.tensor([True, True, True, False]))
    assert (m._lb, m._ub) == (0, 8) and m.ready(4) and (m._lb, m._ub) == (5, 5)         # straddles: one read, exact after it
    m.add(*row(1, 2)); m.add_masked(*row(1, 2), torch.tensor([False, False]))
    assert (m._lb, m._ub) == (7, 9) and m.ready(4) and (m._lb, m._ub) == (7, 9)         # lb > n: no read
    assert len(m) == 7
    # an Agent fed by masked adds only learns as soon as the ring holds more than a batch -- no sync_len() by the caller
    ag = Agent(obs_dim=6, hidden=8, batch_size=4, memory_size=32, device="cpu", seed=0)
    assert ag.learn(1) is None
    for _ in range(2):
        ag.memory.add_masked(torch.randn(4, 6), torch.rand(4, 2), torch.randn(4), torch.randn(4, 6), torch.zeros(4), torch.tensor([True, True, True, False]))
    loss = ag.learn(1)
    assert loss is not None and torch.isfinite(loss)


def test_episode_csv_schema(tmp_path):
    """utils.record_data (UTL:53-64) + TRAIN:157-162: header row, then one row per finished episode
    [episode_number, success, failure, return, steps, ego_safety_score, social_safety_score, timelapse]."""
    import csv
    from crowdnav.rollout import EpisodeStats
    st = EpisodeStats()
    st.add(True, False, 123.5, 41, 0.9, 0.75, 6.15)
    st.add(False, True, -90.0, 12, 1.0, 1.0)
    path = st.write_csv(str(tmp_path), "td3_training_trajectory_test")
    assert os.path.basename(path) == "td3_training_trajectory_test.csv"
    rows = list(csv.reader(open(path)))
    assert rows[0] == ['episode_number', 'success_episode', 'failure_episode', 'episode_reward', 'episode_step',
                       'ego_safety_score', 'social_safety_score', 'timelapse']
    assert rows[1][:5] == ['1', 'True', 'False', '123.5', '41'] and rows[2][0] == '2' and len(rows) == 3
