"""Diagnostic: does the graphed TD3 update (Agent.enable_graphs) learn like the eager one?  Fits q1 to a synthetic reward."""
import sys, torch
sys.path.insert(0, "/root/repo/drl-based-mapless-crowd-navigation-with-perceived-risk_amd")
from crowdnav.td3 import Agent
def build(empty_first):
    ag = Agent(obs_dim=398, device="cuda", seed=0, batch_size=128, memory_size=100000)
    return ag
def fill(ag, n=20000):
    g = torch.Generator(device="cuda").manual_seed(3)
    s = torch.randn((n, 398), generator=g, device="cuda"); a = torch.rand((n, 2), generator=g, device="cuda")
    r = (s[:, 0] + a[:, 0]).contiguous(); s2 = torch.randn((n, 398), generator=g, device="cuda"); d = torch.rand(n, generator=g, device="cuda") < 0.05
    ag.memory.add(s, a, r, s2, d)
def evalq(ag):
    m = ag.memory
    with torch.no_grad():
        return float(((ag.q1(m.s[:2048], m.a[:2048]) - m.r[:2048]) ** 2).mean())
for mode in ("eager", "graph_after_fill", "graph_before_fill"):
    ag = build(False)
    if mode == "graph_before_fill":
        ag.enable_graphs()
    fill(ag)
    if mode == "graph_after_fill":
        ag.enable_graphs()
    p0 = [p.detach().clone() for p in ag.q1.parameters()]
    out = [evalq(ag)]
    for i in range(600):
        ag.learn(i)
        if i % 200 == 199: out.append(evalq(ag))
    torch.cuda.synchronize()
    drift = sum(float((p - q).abs().sum()) for p, q in zip(ag.q1.parameters(), p0))
    fin = all(bool(torch.isfinite(p).all()) for m in (ag.actor, ag.q1, ag.q2, ag.actor_t) for p in m.parameters())
    print(mode, "q-fit error", ["%.4f" % x for x in out], "param drift %.3f finite %s" % (drift, fin), flush=True)
