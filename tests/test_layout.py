"""CPU checks of the boundary: the C-ABI library loads, exports every symbol include/crowdnav.h declares,
fails loudly without a GPU, and the product never touches the oracle."""
import ctypes as C
import os
import re

import pytest

from conftest import PKG, ROOT


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "crowdnav.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cn_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import crowdnav
    crowdnav.build()
    L = C.CDLL(crowdnav._abi.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(L, s), "libcrowdnav.so does not export %s" % s
    assert set(syms) == set(crowdnav._abi.EXPORTS)
    assert L.cn_abi_version() == crowdnav._abi.EXPECTED_ABI == 7


def test_config_struct_matches_header_and_oracle():
    from crowdnav.config import CnConfig, Config
    from oracle import oracle
    assert C.sizeof(CnConfig) == C.sizeof(oracle.CnoConfig) == 20 * 4 + 8 + 8 + 27 * 8
    from crowdnav._abi import CnSnapshotHeader
    assert C.sizeof(CnSnapshotHeader) == 8 + 6 * 4 + 8 + C.sizeof(CnConfig)      # cn_snapshot_header (include/crowdnav.h)
    assert [f[0].replace("track_capacity", "reserved0") for f in CnConfig._fields_] == [f[0] for f in oracle.CnoConfig._fields_]
    d = Config().as_dict()
    for k, v in oracle.DEFAULTS.items():
        if k in ("reserved0", "ped_cycle_ms"):
            continue
        assert d[k] == v, k
    assert Config().obs_dim == 398 and Config(k_obstacles=4).obs_dim == 382  # TRAIN:88, checkpoints


def test_python_mirrors_match_the_header_field_by_field(tmp_path):
    """Every struct that crosses the boundary: sizeof and the offset of EVERY field as gcc lays out include/crowdnav.h, against
    the ctypes mirror the Python host passes (a field added to one side only, or a reordering, shows up here, not as a silently
    misread pointer on the GPU)."""
    import shutil
    import subprocess
    from crowdnav import _abi
    from crowdnav.config import CnConfig
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = [("cn_config", CnConfig), ("cn_step_io", _abi.CnStepIO), ("cn_external_io", _abi.CnExternalIO),
             ("cn_actor_weights", _abi.CnActorWeights), ("cn_td3_mlp", _abi.CnTd3Mlp), ("cn_td3_config", _abi.CnTd3Config),
             ("cn_td3_batch", _abi.CnTd3Batch), ("cn_replay_ring", _abi.CnReplayRing), ("cn_episode_log", _abi.CnEpisodeLog), ("cn_sequence_io", _abi.CnSequenceIO), ("cn_policy_io", _abi.CnPolicyIO),
             ("cn_snapshot_header", _abi.CnSnapshotHeader)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "crowdnav.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f[0], cname, f[0]))
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        a, b, c = ln.split()
        got[(a, b)] = int(c)
    n = 0
    for cname, cls in pairs:
        assert got[(cname, "sizeof")] == C.sizeof(cls), cname
        for f in cls._fields_:
            assert got[(cname, f[0])] == getattr(cls, f[0]).offset, (cname, f[0])
            n += 1
    assert n > 150


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import crowdnav
    from crowdnav.config import Config
    L = crowdnav.lib()
    h = C.c_void_p()
    cfg = Config().to_c()
    rc = L.cn_create(C.byref(cfg), 0, C.byref(h))
    assert rc == -3 and b"no CPU fallback" in L.cn_last_error()
    from crowdnav.env import VecEnv
    with pytest.raises(crowdnav.CrowdNavError):
        VecEnv(Config())


def test_cn_create_rejects_configs_the_reference_cannot_run():
    """Range checks come before any device work, so they are testable without a GPU.  max_scan_range == min_scan_range is a
    ZeroDivisionError in the reference (ENV:581, UTL:322 divide by the difference)."""
    import crowdnav
    from crowdnav.config import Config
    L = crowdnav.lib()
    L.cn_last_error.restype = C.c_char_p
    for bad, msg in ((dict(max_scan_range=0.5, min_scan_range=0.5), b"max_scan_range"),
                     (dict(max_scan_range=0.1, min_scan_range=0.12), b"max_scan_range"),
                     (dict(n_rays=4), b"out of range"), (dict(k_obstacles=0), b"out of range"),
                     (dict(risk_mode=1, obs_layout=1), b"risk_mode gt"), (dict(py2_round=3), b"out of range"),
                     (dict(ped_mode=2, sf_B=0.0), b"social force"), (dict(ped_mode=2, obs_layout=1), b"social force"),
                     (dict(ped_mode=2, n_peds=100, n_rays=360), b"social force"), (dict(scan_f32=2), b"out of range"),
                     (dict(wheel_accel=-1.0), b"wheel_accel"), (dict(wheel_accel=1.0, ped_contact=1), b"wheel_accel"),
                     (dict(wheel_accel=1.0, obs_layout=1), b"wheel_accel"), (dict(wheel_accel=1.0, wheel_separation=0.0), b"wheel_accel")):
        h = C.c_void_p()
        cfg = Config(**bad).to_c()
        rc = L.cn_create(C.byref(cfg), 0, C.byref(h))
        assert rc == -2 and msg in L.cn_last_error(), (bad, rc, L.cn_last_error())
        assert not h.value


def test_product_never_references_the_oracle_or_the_reference_tree():
    bad = []
    for base, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                txt = open(os.path.join(base, f)).read()
                if re.search(r"\bimport oracle\b|from oracle\b|cn_oracle|libcn_oracle", txt):
                    bad.append(os.path.join(base, f))
                if "/root/reference" in txt and f.endswith(".py"):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_lds_budget_of_benchmark_configs():
    import crowdnav
    L = crowdnav.lib()
    L.cn_lds_bytes.restype = C.c_size_t
    L.cn_lds_bytes.argtypes = [C.c_int] * 5
    assert L.cn_lds_bytes(360, 20, 8, 359 // 4 + 2, 32) <= 10 * 1024      # 16 wavefronts per CU
    assert L.cn_lds_bytes(720, 100, 8, 719 // 4 + 2, 64) <= 160 * 1024


def test_training_preset_and_the_cospawned_obstacles():
    """presets.training: the world file's 14 obstacles, or (drop_cospawned) the six that have a spawn pose of their own --
    obstacles 7-14 are all created at (0.22, 0.54) (turtlebot3_crowd_dense.world:447-867); the crowd node's round stays 1.4 s."""
    from crowdnav import presets
    cfg, init = presets.training(n_envs=3)
    assert cfg.n_peds == 14 and cfg.ped_cycle_ms == 1400 and init.shape == (3, 14, 2)
    assert len({tuple(p) for p in init[0]}) == 7 and (init[0, 6:] == init[0, 6]).all()
    cfg6, init6 = presets.training(n_envs=3, drop_cospawned=True)
    assert cfg6.n_peds == 6 and cfg6.ped_cycle_ms == 1400 and init6.shape == (3, 6, 2)
    assert (init6[0] == init[0, :6]).all() and cfg6.ped_vmax == cfg.ped_vmax == 0.2


def test_episode_csv_resumes_instead_of_truncating(tmp_path):
    """utils.record_data (UTL:53-64) appends; a run continued into its own directory must not truncate the rows it wrote before
    (ADVICE r05), and a fresh start must."""
    import csv
    from crowdnav.rollout import EpisodeStats
    a = EpisodeStats()
    a.add(1, 0, 10.0, 50, 1.0, 0.9, 8.0); a.add(0, 1, -200.0, 20, 0.5, 0.5, 3.2)
    a.append_csv(str(tmp_path), "td3_training")
    a.add(1, 0, 12.0, 40, 1.0, 1.0, 6.4)
    path = a.append_csv(str(tmp_path), "td3_training")
    rows = list(csv.reader(open(path)))
    assert rows[0] == EpisodeStats.HEADERS and [r[0] for r in rows[1:]] == ["1", "2", "3"]
    b = EpisodeStats()                                 # the resumed process: its own rows start at 1 again
    b.add(1, 0, 7.0, 30, 1.0, 1.0, 4.8)
    b.append_csv(str(tmp_path), "td3_training", resume=True)
    b.add(0, 1, -1.0, 31, 1.0, 1.0, 4.96)
    b.append_csv(str(tmp_path), "td3_training", resume=True)
    rows = list(csv.reader(open(path)))
    assert rows[0] == EpisodeStats.HEADERS and sum(1 for r in rows if r == EpisodeStats.HEADERS) == 1
    assert [r[0] for r in rows[1:]] == ["1", "2", "3", "4", "5"] and rows[4][3] == "7.0"
    c = EpisodeStats()                                 # a fresh start truncates
    c.add(1, 0, 1.0, 1, 1.0, 1.0, 0.16)
    c.append_csv(str(tmp_path), "td3_training")
    assert [r[0] for r in list(csv.reader(open(path)))[1:]] == ["1"]


def test_load_episode_latest_pointer(tmp_path):
    """Checkpoints carry the live episode count; `--load-episode latest` finds the newest one."""
    import pytest
    from crowdnav import train

    class FakeAgent:
        def save(self, outdir, ep): open(os.path.join(outdir, "td3_actor_model_ep%d.pt" % ep), "w").close()
        def noise_state(self): return (3, 4)

    with pytest.raises(FileNotFoundError):
        train.resolve_load_episode(str(tmp_path), "latest")
    train.save_checkpoint(FakeAgent(), str(tmp_path), 28013)
    train.save_checkpoint(FakeAgent(), str(tmp_path), 28127)
    assert train.resolve_load_episode(str(tmp_path), "latest") == 28127
    assert train.resolve_load_episode(str(tmp_path), "28013") == 28013 and train.resolve_load_episode(str(tmp_path), 5) == 5
    assert open(tmp_path / "noise_state_ep28127.txt").read().split() == ["3", "4"]
    assert os.path.exists(tmp_path / "td3_actor_model_ep28013.pt")
