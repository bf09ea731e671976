"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path through the C-ABI against
  * the host build of the stepper core (oracle/_build/liboracle_port.so) tick by tick and gym step by gym step,
  * the committed golden vectors generated from the real reference / torch (tests/golden/*.npz),
  * the numpy learner oracle (oracle/learner_ref.py),
  * the real reference itself when oracle/_ref/libref_oracle.so travelled with the snapshot,
and size-independent properties at the BASELINE sizes (4096 envs)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import learner_ref as R  # noqa: E402
from rlgymppo_cpp_amd.state import ArenaState, default_arena  # noqa: E402
from simlib import port_gym_cfg, port_gym_reset, port_gym_step  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def sg():
    return np.load(os.path.join(GOLD, "sim_golden.npz"))


@pytest.fixture(scope="module")
def lg():
    return np.load(os.path.join(GOLD, "learner_golden.npz"))


def _vec(s):
    v = list(s.ball.pos) + list(s.ball.vel) + list(s.ball.ang_vel)
    for k in range(s.num_cars):
        c = s.cars[k]
        v += list(c.pos) + list(c.vel) + list(c.ang_vel) + list(c.rot) + [float(c.flags), c.boost, c.jump_time, c.flip_time, c.handbrake_val]
    return np.array(v, np.float64)


def test_upload_download_roundtrip():
    from rlgymppo_cpp_amd.env import BatchedEnv
    env = BatchedEnv(130, 1)
    rng = np.random.RandomState(0)
    states = []
    for i in range(130):
        s = default_arena(2)
        s.ball.pos[:] = list(rng.uniform(-1000, 1000, 3) + [0, 0, 1500]); s.ball.vel[:] = list(rng.uniform(-500, 500, 3))
        s.cars[0].boost = float(rng.uniform(0, 100)); s.cars[1].bh_tick_hit = int(rng.randint(0, 1 << 40)); s.tick_count = int(rng.randint(0, 1 << 40))
        s.pads[i % 34].cooldown = 3.5; s.pads[i % 34].is_active = 0; s.pads[i % 34].prev_locked_car_id = 2
        s.gym.players[1].match_goals = i; s.gym.no_touch_steps = i * 3
        states.append(s)
    env.upload_states(states)
    back = env.download_states()
    for a, b in zip(states, back):
        assert np.allclose(_vec(a), _vec(b), rtol=3e-7, atol=1e-6)   # uu -> BT -> uu: two fp32 roundings
        assert a.tick_count == b.tick_count and a.cars[1].bh_tick_hit == b.cars[1].bh_tick_hit
        assert [p.is_active for p in a.pads] == [p.is_active for p in b.pads] and abs(a.pads[3].cooldown - b.pads[3].cooldown) < 1e-6
        assert a.gym.players[1].match_goals == b.gym.players[1].match_goals and a.gym.no_touch_steps == b.gym.no_touch_steps


def test_hip_free_run_is_bit_identical_to_the_reference(sg):
    """The HIP stepper against the REFERENCE's recorded trajectories, directly: every tape runs as one env of a batch (1v1, 2v2, 3v3), the
    controls of the tape go in through rlgpu_env_set_controls (nothing else of the resident state is touched or rounded) and every 10 ticks
    the states are downloaded and compared with the reference's for EQUALITY of every field of every body: all 31 tapes over their
    whole length (up to 620 ticks; simlib.PHYS_EXACT_UNTIL, the table of tapes that stop being exact at a named tick, is empty)."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import state_vec, PHYS_EXACT_UNTIL
    every = int(sg["phys_every"])
    names = [str(n) for n in sg["phys_names"]]
    compared = 0
    for nc in (2, 4, 6):
        grp = [n for n in names if ArenaState.from_buffer_copy(sg[f"phys/{n}/start_raw"].tobytes()).num_cars == nc]
        if not grp:
            continue
        env = BatchedEnv(len(grp), nc // 2, mesh=(sg["mesh_verts"], sg["mesh_tris"]))
        env.upload_states([ArenaState.from_buffer_copy(sg[f"phys/{n}/start_raw"].tobytes()) for n in grp])
        tapes = [sg[f"phys/{n}/tape"] for n in grp]
        T = max(len(t) for t in tapes)
        ctl = np.zeros((len(grp), nc, 8), np.float32)
        for t in range(T):
            for i, tp in enumerate(tapes):
                if t < len(tp):
                    ctl[i] = tp[t]
            env.set_controls(ctl)
            env.physics_ticks(1)
            if (t + 1) % every == 0:
                cur = env.download_states()
                for i, n in enumerate(grp):
                    if t + 1 <= min(PHYS_EXACT_UNTIL.get(n, len(tapes[i])), len(tapes[i])):
                        assert np.array_equal(state_vec(cur[i]), sg[f"phys/{n}/states"][(t + 1) // every - 1]), f"{n} tick {t + 1}: HIP state is not the reference's"
                        compared += every
        env.close()
    print("HIP free-run ticks bit-identical to the reference:", compared)


def test_physics_ticks_match_host_port_on_golden_scenarios(sg, port_lib):
    """Every golden physics scenario as one env of a batch, free-running on the device next to the host build of the same core: every
    body's position, velocity, angular velocity, rotation, flags, boost and timers are compared after EVERY tick of every tape and
    must be EQUAL, bit for bit.  (+ - * / sqrt are correctly rounded on both sides and contraction is off; the libm calls of the physics
    go through csrc/rl_libm.h, which computes glibc's bits on the host and on the device.  Before that the two builds agreed to 6e-5 per
    tick and a contact decision on the fence could flip between them.)  The device state is never re-synced to the host's: only the
    controls of the tape are written into it each tick."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    names = [str(n) for n in sg["phys_names"] if ArenaState.from_buffer_copy(sg[f"phys/{str(n)}/start"].tobytes()).num_cars == 2]
    env = BatchedEnv(len(names), 1, mesh=(sg["mesh_verts"], sg["mesh_tris"]))
    host = [ArenaState.from_buffer_copy(sg[f"phys/{n}/start"].tobytes()) for n in names]
    tapes = [sg[f"phys/{n}/tape"] for n in names]
    T = max(len(t) for t in tapes)
    env.upload_states(host)
    hists = [(C.c_uint16 * 8)() for _ in host]     # an env slot keeps its broadphase history across uploads (an upload is a SetState): so does the host side
    compared = 0
    for t in range(T):
        cur = env.download_states()
        for i, s in enumerate(host):
            if t < len(tapes[i]):
                for k in range(2):
                    s.cars[k].controls[:] = list(tapes[i][t, k])
                    cur[i].cars[k].controls[:] = list(tapes[i][t, k])
        env.upload_states(cur)
        env.physics_ticks(1)
        cur = env.download_states()
        for i, s in enumerate(host):
            if t < len(tapes[i]):
                port_lib.step(s, 1, hist=hists[i])
                a, b = _vec(s), _vec(cur[i])
                assert np.array_equal(a, b), f"{names[i]} tick {t + 1}: HIP and host build differ, max |diff| {np.abs(a - b).max()} at {int(np.abs(a - b).argmax())}"
                compared += 1
    print("ticks compared bit for bit, HIP vs host build:", compared)


def test_gym_step_matches_host_port(port_lib):
    """64 random envs, 48 gym steps (incl. auto-resets): obs / reward / done of the HIP path vs the host build of the same source, EQUAL
    bit for bit at every step (both start each step from the same bits)."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from rlgymppo_cpp_amd import _lib
    n = 64
    cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 20
    env = BatchedEnv(n, 1, cfg=cfg)
    pcfg = port_gym_cfg(no_touch_max_steps=20)
    obs = env.reset(True)
    env.sync()
    states = env.download_states()
    # the host port resets from the same seed/stream -> identical states and obs
    hs = [default_arena(2) for _ in range(n)]
    hs, hobs = port_gym_reset(port_lib, hs, pcfg, run_setter=True)
    assert np.abs(obs.cpu().numpy() - hobs).max() < 1e-5
    env.upload_states(hs)     # both sides continue from the same bits (the host build hands its states over in uu)
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(5)
    nobs = torch.empty_like(obs); rew = torch.empty(n * 2, device=dev); done = torch.empty(n * 2, dtype=torch.int32, device=dev)
    n_done = 0
    hist = np.zeros((n, 8), np.uint16)     # the host side keeps each arena's broadphase history across the hand-overs, as the env slots do
    for step in range(48):
        acts = rng.randint(0, 90, size=n * 2).astype(np.int32)
        env.step(torch.from_numpy(acts).to(dev), nobs, rew, done)
        env.sync()
        hs, ho, hr, hd = port_gym_step(port_lib, hs, pcfg, acts, hist)
        d = done.cpu().numpy()
        assert (d == hd).all(), f"done flags differ at step {step}"
        n_done += int(hd.sum())
        assert np.abs(rew.cpu().numpy() - hr).max() <= 2.4e-7, f"rewards differ at step {step}: max |diff| {np.abs(rew.cpu().numpy() - hr).max()}"   # (the host build converts a state to uu and back once more: two ulps of a reward)
        assert np.array_equal(nobs.cpu().numpy(), ho), f"obs differ at step {step}: max |diff| {np.abs(nobs.cpu().numpy() - ho).max()}"
        # the host build hands its states over in uu: the device continues from those bits (a gym step from equal bits gives EQUAL rows)
        env.upload_states(hs)
    assert n_done > 0   # auto-reset path exercised


@pytest.mark.parametrize("team_size,max_players", [(2, 2), (3, 3), (1, 2), (2, 4)])
def test_gym_step_matches_host_port_team_modes(port_lib, team_size, max_players):
    """BASELINE configs[3]/[4] shapes: 2v2 and 3v3 envs (4 / 6 cars, obs width 127 / 165) with the zero-sum reward wrapper,
    HIP path vs host port over 24 gym steps incl. auto-resets, every observation row and reward EQUAL.  Also the only GPU coverage of the 2- and 1-env-per-wavefront
    lane maps (32 / 64 lanes per env in the candidate, pair and item phases), and of the padded-shuffled obs."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from rlgymppo_cpp_amd import _lib
    n, nc = 40, 2 * team_size
    cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 12; cfg.zero_sum = 1; cfg.team_spirit = 0.3; cfg.opp_scale = 1.0
    cfg.obs_max_players = max_players    # DefaultOBSPadded(maxPlayers): teammate / opponent blocks padded with zero blocks and shuffled
    env = BatchedEnv(n, team_size, cfg=cfg)
    assert env.obs_size == 51 + 38 * max_players and env.n_agents == n * nc
    pcfg = port_gym_cfg(no_touch_max_steps=12, zero_sum=1, team_spirit=0.3, opp_scale=1.0, obs_max_players=max_players)
    obs = env.reset(True)
    env.sync()
    hs, hobs = port_gym_reset(port_lib, [default_arena(nc) for _ in range(n)], pcfg, run_setter=True)
    assert np.abs(obs.cpu().numpy() - hobs).max() < 1e-5
    env.upload_states(hs)
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(11 + team_size)
    nobs = torch.empty_like(obs); rew = torch.empty(n * nc, device=dev); done = torch.empty(n * nc, dtype=torch.int32, device=dev)
    n_done = 0
    hist = np.zeros((n, 8), np.uint16)
    for step in range(24):
        acts = rng.randint(0, 90, size=n * nc).astype(np.int32)
        env.step(torch.from_numpy(acts).to(dev), nobs, rew, done)
        env.sync()
        hs, ho, hr, hd = port_gym_step(port_lib, hs, pcfg, acts, hist)
        assert (done.cpu().numpy() == hd).all(), f"done flags differ at step {step}"
        n_done += int(hd.sum())
        assert np.abs(rew.cpu().numpy() - hr).max() <= 2.4e-7, f"rewards differ at step {step}: max |diff| {np.abs(rew.cpu().numpy() - hr).max()}"   # (the host build converts a state to uu and back once more: two ulps of a reward)
        assert np.array_equal(nobs.cpu().numpy(), ho), f"obs differ at step {step}: max |diff| {np.abs(nobs.cpu().numpy() - ho).max()}"
        env.upload_states(hs)
    assert n_done > 0


# (The .cmf directory loader -- rlgpu_env_load_cmf_dir, RS/RocketSim.cpp:70-212 -- is held to EQUALITY with the reference by the two-file, sixteen-file
# and round-robin fixtures below (test_hip_mesh_of_two_files..., ..._tessellated_mesh_of_sixteen_files..., ..._wedge_fixture...); a looser test that
# compared it with the procedural mesh through a uu round trip of the vertices, 97 % of envs within 2e-3, was deleted in round 5.)


def _gym_cfg(team, tick_skip, omp, rk, nts):
    """simlib.gym_cfg_for_case as the C-ABI's config struct (same layout)."""
    from simlib import gym_cfg_for_case
    return gym_cfg_for_case(team, tick_skip, omp, rk, nts)


def test_hip_one_team_gym_rollouts_vs_reference_fixtures():
    """Match(..., spawnOpponents = false) on the HIP path against the reference's one-team gyms (tests/golden/sim_golden_one_team.npz):
    agent rows = the blue cars only, no opponent blocks in DefaultOBS, zero blocks in DefaultOBSPadded, ZeroSumReward with an empty team."""
    test_hip_gym_rollouts_vs_reference_fixtures(np.load(os.path.join(GOLD, "sim_golden_one_team.npz")))


def test_hip_gym_rollouts_under_mutators_vs_reference_fixtures():
    """The two Gym rollouts the reference made under MutatorConfig M1 (tests/golden/mutator_golden.npz: goal line 5000 and gravity -325 as GameEventTracker sees
    them through the arena, next to GoalScoreCondition's own constant) on the HIP path: the block arrives with the uploaded start state."""
    test_hip_gym_rollouts_vs_reference_fixtures(np.load(os.path.join(GOLD, "mutator_golden.npz")))


def test_hip_gym_rollouts_vs_reference_fixtures(sg):
    """Every committed rollout of the REAL reference Gym replayed on the HIP path, no port in between: 1v1 example stack (full 160
    steps, the NoTouch timeout, a goal), 2v2 with every CommonRewards term (goal + assist + shot pass; shot + save + bump + demo),
    zero-sum, DefaultOBSPadded(3), 3v3 DefaultOBS(165): done exactly, rewards, observation rows, and the event counters.  ALL of them
    (simlib.GYM_EXACT: up to 160 gym steps = 1280 ticks of random play in 1v1, 100 steps of random 2v2, 90 of random 3v3, a chase, the
    timeout, goals, assists, a save, a demolition, every reward term, zero-sum) are reproduced EXACTLY: every observation row and every
    reward bit-equal to the reference's.  (The one-team test below runs the same body over the four spawnOpponents = false rollouts.)"""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import gym_compare_obs, GYM_OBS_TOL, GYM_HORIZON, GYM_EXACT, GYM_EXACT_OBS
    dev = torch.device("cuda", 0)
    for case in sg["gym_names"]:
        case = str(case)
        exact = case in GYM_EXACT      # observations and rewards EQUAL to the reference's, bit for bit
        team, tick_skip, omp, rk, nts = [int(x) for x in sg[f"gym/{case}/cfg"][:5]]
        one_team = len(sg[f"gym/{case}/cfg"]) > 5 and int(sg[f"gym/{case}/cfg"][5]) == 0      # spawnOpponents = false
        nc = 2 * team
        gcfg = _gym_cfg(team, tick_skip, omp, rk, nts); gcfg.one_team = 1 if one_team else 0
        env = BatchedEnv(1, team, cfg=gcfg, mesh=(sg["mesh_verts"], sg["mesh_tris"]))
        rows = env.n_agents
        assert rows == (team if one_team else nc)
        st = ArenaState.from_buffer_copy(sg[f"gym/{case}/start_raw" if f"gym/{case}/start_raw" in sg.files else f"gym/{case}/start"].tobytes())
        env.upload_states([st])
        obs0 = env.reset(False)
        gym_compare_obs(obs0.cpu().numpy(), sg[f"gym/{case}/obs0"], nc, omp, [int(x) for x in sg[f"gym/{case}/player_order"][0]], 1e-5, f"{case} reset", one_team)
        acts = sg[f"gym/{case}/actions"]; obs = sg[f"gym/{case}/obs"]; rew = sg[f"gym/{case}/rew"]; done = sg[f"gym/{case}/done"]
        nobs = torch.empty((rows, env.obs_size), device=dev); r = torch.empty(rows, device=dev); d = torch.empty(rows, dtype=torch.int32, device=dev)
        for t in range(min(len(acts), GYM_HORIZON.get(case, len(acts)))):
            env.step(torch.from_numpy(acts[t].astype(np.int32)).to(dev), nobs, r, d)
            env.sync()
            assert int(d[0]) == int(done[t]), f"{case}: done differs at step {t}"
            rr = r.cpu().numpy()
            assert np.abs(rr - rew[t]).max() < 2e-3 * max(1.0, np.abs(rew[t]).max()), f"{case}: reward differs at step {t}: {rr} vs {rew[t]}"
            assert not exact or np.array_equal(rr, rew[t]), f"{case}: reward not bit-equal to the reference at step {t}: {rr} vs {rew[t]}"
            if done[t]:
                break
            gym_compare_obs(nobs.cpu().numpy(), obs[t], nc, omp, [int(x) for x in sg[f"gym/{case}/player_order"][t]], 1e-30 if (exact or case in GYM_EXACT_OBS) else GYM_OBS_TOL.get(case, 2e-3), f"{case} step {t}", one_team)
        if not (done[-1] or case in GYM_HORIZON):
            fin = ArenaState.from_buffer_copy(sg[f"gym/{case}/final"].tobytes())
            got = env.download_states()[0]
            for k in range(0, nc, 2 if one_team else 1):
                a, b = got.gym.players[k], fin.gym.players[k]
                assert (a.match_goals, a.match_assists, a.match_shots, a.match_saves, a.match_shot_passes, a.match_bumps, a.match_demos, a.boost_pickups) == \
                       (b.match_goals, b.match_assists, b.match_shots, b.match_saves, b.match_shot_passes, b.match_bumps, b.match_demos, b.boost_pickups), f"{case}: counters of player {k}"
        env.close()


def test_hip_gameinst_episode_boundaries_vs_reference(sg):
    """SURVEY A9 on the HIP path: GameInst::Step ACROSS episode ends against recordings of the reference's own GameInst.cpp (30 episode ends over
    four cases: NoTouch timeouts and goals; 1v1 example stack, 2v2 every reward term inside ZeroSumReward, 3v3 DefaultOBSPadded(3)) with a
    replayable user state setter: the k-th reset installs the fixture's k-th state through the boundary the C++ host's user state setters use
    (rlgpu_env_upload_states + rlgpu_env_reset_envs with run_setter = 0, Learner::Impl::HostResetEnvs).  EQUALITY: done, every reward, every
    row of curObs -- after an end the first observation of the new episode -- and curEpRew / avgEpRew / avgStepRew / totalSteps."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import gameinst_replay, with_pads_of
    gg = np.load(os.path.join(GOLD, "gameinst_golden.npz"))
    dev = torch.device("cuda", 0)
    total = 0
    for case in gg["names"]:
        case = str(case)
        team, tick_skip, omp, rk, nts = [int(x) for x in gg[f"gi/{case}/cfg"]]
        gcfg = _gym_cfg(team, tick_skip, omp, rk, nts)
        gcfg.host_resets = 1      # the state setter is the test's (a user StateSetter): an env whose episode ended stays as it ended until the masked reset
        env = BatchedEnv(1, team, cfg=gcfg, mesh=(gg["mesh_verts"], gg["mesh_tris"]))
        rows = env.n_agents
        nobs = torch.empty((rows, env.obs_size), device=dev); r = torch.empty(rows, device=dev); d = torch.empty(rows, dtype=torch.int32, device=dev)
        def reset_to(state, first):
            if first:
                env.upload_states([state])
                return env.reset(False).cpu().numpy()
            env.upload_states([with_pads_of(state, env.download_states()[0])])     # what Learner::Impl::HostResetEnvs does around the user's ResetState(Arena*)
            env.reset_envs([0], run_setter=False, obs=nobs)      # the masked reset of an env whose episode ended (GameInst.cpp:27-32)
            env.sync()
            return nobs.cpu().numpy()
        def step(a):
            env.step(torch.from_numpy(a.astype(np.int32)).to(dev), nobs, r, d)
            env.sync()
            return nobs.cpu().numpy(), r.cpu().numpy(), int(d[0])
        total += gameinst_replay(gg, case, reset_to, step, 1e-30, True, "HIP")
        env.close()
    assert total >= 12


def test_hip_random_state_resets_pads_before_first_observation():
    """ADVICE r04 (high) on the HIP path: with the built-in RandomState every new episode's first observation shows all 34 pads active, as the
    reference's does (RandomState.cpp:11 -> Arena.cpp:209-210; tests/golden/padreset_golden.npz recorded from the reference's GameInst with its
    own RandomState: the cars empty their pads in the first episode, the pad columns are compared for equality up to the second episode's start)."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from test_oracle_golden import padreset_replay
    pg = np.load(os.path.join(GOLD, "padreset_golden.npz"))
    dev = torch.device("cuda", 0)
    total = 0
    for case in pg["names"]:
        case = str(case)
        team, tick_skip, nts = [int(x) for x in pg[f"pr/{case}/cfg"]]
        env = BatchedEnv(1, team, cfg=_gym_cfg(team, tick_skip, 0, 0, nts), mesh=(pg["mesh_verts"], pg["mesh_tris"]))
        rows = env.n_agents
        nobs = torch.empty((rows, env.obs_size), device=dev); r = torch.empty(rows, device=dev); d = torch.empty(rows, dtype=torch.int32, device=dev)
        def reset_first(state):
            env.upload_states([state])
            return env.reset(False).cpu().numpy()
        def step(a):
            env.step(torch.from_numpy(a.astype(np.int32)).to(dev), nobs, r, d)
            env.sync()
            return nobs.cpu().numpy(), int(d[0])
        total += padreset_replay(pg, case, reset_first, step, "HIP")
        env.close()
    assert total >= 6


def test_hip_physics_free_run_vs_reference_fixtures(sg):
    """The 31 physics scenarios stepped by the HIP kernel from the reference's start state under the recorded control tape and compared
    with the REFERENCE's states every 10 ticks -- position, velocity, angular velocity, rotation of the ball and every car, flags of
    every car exactly -- with no re-sync to anything (tolerances and the three horizons: simlib.PHYS_FREE_RUN; the 1v1 tapes that are
    bit-identical to the reference are asserted so in test_hip_free_run_is_bit_identical_to_the_reference)."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import PHYS_FREE_RUN, state_vec, phys_errors
    every = int(sg["phys_every"])
    names = [str(n) for n in sg["phys_names"]]
    for nc in (2, 4, 6):
        grp = [n for n in names if ArenaState.from_buffer_copy(sg[f"phys/{n}/start"].tobytes()).num_cars == nc]
        if not grp:
            continue
        env = BatchedEnv(len(grp), nc // 2, mesh=(sg["mesh_verts"], sg["mesh_tris"]))
        cur = [ArenaState.from_buffer_copy(sg[f"phys/{n}/start"].tobytes()) for n in grp]
        tapes = [sg[f"phys/{n}/tape"] for n in grp]
        T = max(len(t) for t in tapes)
        env.upload_states(cur)
        ctl = np.zeros((len(grp), nc, 8), np.float32)
        for t in range(T):
            for i in range(len(grp)):
                if t < len(tapes[i]):
                    ctl[i] = tapes[i][t]
            env.set_controls(ctl)           # only the controls change; the resident state is not rounded to uu and back between ticks
            env.physics_ticks(1)
            if (t + 1) % every == 0:
                got = env.download_states()
                for i, n in enumerate(grp):
                    tol = PHYS_FREE_RUN[n]; until = tol.get("until") or len(tapes[i])
                    if t + 1 > min(until, len(tapes[i])):
                        continue
                    pos, vel, ang, rot, fl = phys_errors(state_vec(got[i]), sg[f"phys/{n}/states"][(t + 1) // every - 1], nc)
                    assert pos <= 1.5 * tol["pos"] and vel <= 1.5 * tol["vel"] and ang <= 1.5 * tol["ang"] and rot <= 1.5 * tol["rot"], \
                        f"{n} tick {t + 1}: pos {pos:.4f} vel {vel:.4f} ang {ang:.5f} rot {rot:.6f} (tol {tol})"
                    assert not fl, f"{n} tick {t + 1}: car flags differ from the reference"
        env.close()


def test_hip_one_tick_vs_reference_states(sg):
    """All 1722 recorded (reference state, reference state one tick later) pairs: upload, one tick of the HIP kernel, compare for EQUALITY
    of every field of every body -- every pair of all 31 scenarios is bit-equal to the reference (1580 1v1 pairs, every tick with a
    narrowphase contact among them; deep contacts through the penetration-depth solver; the six-car heap with its two demolitions).
    One env slot per scenario, its pairs uploaded and ticked one after the other as the reference recorded them in one arena
    (make_sim_golden.py): an upload is a SetState, the broadphase's arrival order passes from pair to pair."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import ONE_TICK_TOL, ONE_TICK_NOT_EXACT_MAX, state_vec, phys_errors
    ss = np.load(os.path.join(GOLD, "sim_steps.npz"))
    names = [str(x) for x in ss["phys_names"]]
    n_exact = n_all = 0
    for nc in (2, 4, 6):
        B, A, T = ss[f"nc{nc}/before"], ss[f"nc{nc}/after"], ss[f"nc{nc}/tag"]
        scen = sorted(set(int(t[0]) for t in T))
        pairs = {si: [i for i in range(len(B)) if int(T[i][0]) == si] for si in scen}
        env = BatchedEnv(len(scen), nc // 2, mesh=(sg["mesh_verts"], sg["mesh_tris"]))
        got = [None] * len(B)
        for j in range(max(len(v) for v in pairs.values())):
            slots = [e for e, si in enumerate(scen) if j < len(pairs[si])]
            env.upload_states([ArenaState.from_buffer_copy(B[pairs[scen[e]][j]].tobytes()) for e in slots], env_ids=slots)
            env.physics_ticks(1)
            out = env.download_states(env_ids=slots)
            for e, st in zip(slots, out): got[pairs[scen[e]][j]] = st
        for i in range(len(B)):
            want = ArenaState.from_buffer_copy(A[i].tobytes())
            exact = np.array_equal(state_vec(got[i]), state_vec(want))
            if not exact:
                pos, vel, ang, rot, fl = phys_errors(state_vec(got[i]), state_vec(want), nc)
                tol = ONE_TICK_TOL.get(names[T[i][0]], ONE_TICK_TOL["default"])
                assert pos <= tol["pos"] and vel <= tol["vel"], f"{names[T[i][0]]} tick {T[i][1]}: not bit-equal to the reference (pos {pos:.4g} vel {vel:.4g})"
                assert not fl or tol.get("flags_loose"), f"{names[T[i][0]]} tick {T[i][1]}: flags differ"
            n_all += 1; n_exact += exact
        env.close()
    assert n_all - n_exact <= ONE_TICK_NOT_EXACT_MAX, f"only {n_exact} of {n_all} one-tick pairs bit-equal to the reference"
    print(f"HIP one-tick pairs bit-equal to the reference: {n_exact} of {n_all}")


@pytest.mark.parametrize("team", [1, 2, 3])
def test_hip_state_setters_against_reference_samples(sg, team):
    """SURVEY A8 on the GPU: the KERNEL's RandomState(true, true, true) and KickoffState (rlgpu_env_reset with run_setter = 1) against the
    reference's own 4000 / 600 resets (RandomState.cpp:8-61, Arena.cpp:112-216; sim_golden.npz setter/*): kickoff -- exactly the reference's
    spawn set per car; random -- every column inside the reference's support, quantiles within 5 % of the span."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from rlgymppo_cpp_amd import _lib
    from simlib import setter_samples_compare
    nc = 2 * team
    for kname, kind, n in (("random", 0, 4000), ("kickoff", 1, 600)):
        cfg = _lib.default_gym_config(); cfg.setter_kind = kind; cfg.seed_lo = 77 + team
        env = BatchedEnv(n, team, cfg=cfg)
        env.reset(True); env.sync()
        setter_samples_compare(env.download_states(), sg[f"setter/{kname}/team{team}"], kind, nc, f"HIP {kname} team {team}")
        env.close()


def test_hip_tapes_through_respawns_equal_the_reference():
    """VERDICT r05 "next" 2 on the GPU: the seven recorded tapes of tests/golden/respawn_golden.npz (the reference with its thread engine in a known
    state: 2v2 / 3v3 charges with two or three demolitions, two hunts whose mutual demolition brings both cars back in one tick) as envs of a batch,
    each env drawing its respawn slots from that engine state (RlgpuArenaHidden::ref_engine, uploaded with the start state) -- EQUAL to the reference in
    every field of every body every 10 ticks over the whole tape, >= 300 ticks after the last respawn, and the engine's state with it."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import state_vec
    rg = np.load(os.path.join(GOLD, "respawn_golden.npz")); sgm = np.load(os.path.join(GOLD, "sim_golden.npz"))
    every = int(rg["every"]); names = [str(n) for n in rg["phys_names"]]
    compared = 0
    for nc in (4, 6):
        grp = [n for n in names if ArenaState.from_buffer_copy(rg[f"phys/{n}/start_raw"].tobytes()).num_cars == nc]
        env = BatchedEnv(len(grp), nc // 2, mesh=(sgm["mesh_verts"], sgm["mesh_tris"]))
        env.upload_states([ArenaState.from_buffer_copy(rg[f"phys/{n}/start_raw"].tobytes()) for n in grp])
        tapes = [rg[f"phys/{n}/tape"] for n in grp]; T = max(len(t) for t in tapes)
        ctl = np.zeros((len(grp), nc, 8), np.float32)
        for t in range(T):
            for i, tp in enumerate(tapes):
                if t < len(tp): ctl[i] = tp[t]
            env.set_controls(ctl)
            env.physics_ticks(1)
            if (t + 1) % every == 0:
                cur = env.download_states()
                for i, n in enumerate(grp):
                    if t + 1 <= len(tapes[i]):
                        j = (t + 1) // every - 1
                        assert np.array_equal(state_vec(cur[i]), rg[f"phys/{n}/states"][j]), f"{n} tick {t + 1}: HIP state is not the reference's"
                        assert cur[i].hidden.ref_engine == int(rg[f"phys/{n}/engines"][j]), f"{n} tick {t + 1}: the engines parted"
                        compared += every
        env.close()
    assert compared >= 5500, compared


def test_hip_stress_tapes_of_round_6_equal_the_reference():
    """Round 6 on the GPU: tests/golden/edge_golden.npz -- the 3v3 wall-play tape in which a wheel ray ends ON the side wall's plane at tick 360 (the plane's two
    triangles decide such a ray, not an analytic sign test: csrc/arena_world.h ray_planes), and the two aerial tapes in which a wheel gets the convex cast's
    far "hit" on a car it does not touch (RLGPU_MUT_RAY_PROXY_LISTS in their start states: ray_ball_and_cars) -- EQUAL to the reference every 10 ticks over their
    whole length."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import state_vec
    eg = np.load(os.path.join(GOLD, "edge_golden.npz")); sgm = np.load(os.path.join(GOLD, "sim_golden.npz")); every = int(eg["every"])
    for name in [str(x) for x in eg["phys_names"]]:
        st = ArenaState.from_buffer_copy(eg[f"phys/{name}/start_raw"].tobytes()); nc = st.num_cars
        env = BatchedEnv(1, nc // 2, mesh=(sgm["mesh_verts"], sgm["mesh_tris"]))
        env.upload_states([st])
        tape = eg[f"phys/{name}/tape"]
        for t in range(len(tape)):
            env.set_controls(np.ascontiguousarray(tape[t][None], np.float32))
            env.physics_ticks(1)
            if (t + 1) % every == 0:
                cur = env.download_states()[0]; j = (t + 1) // every - 1
                assert np.array_equal(state_vec(cur), eg[f"phys/{name}/states"][j]), f"{name} tick {t + 1}: HIP state is not the reference's"
                assert cur.hidden.ref_engine == int(eg[f"phys/{name}/engines"][j])
        env.close()


def test_hip_tapes_under_mutators_equal_the_reference():
    """VERDICT r05 "next" 6 on the GPU: the ten tapes of tests/golden/mutator_golden.npz (the reference under two MutatorConfigs in which every run-time field is
    off its default) as envs of a batch -- each env's block arrives with its start state (RlgpuArenaState::mutators), and a second batch gets one set through
    rlgpu_env_set_mutators and start states WITHOUT a block.  EQUAL to the reference in every field of every body, every pad's state and cooldown, every 10 ticks
    over the whole tapes, and the engine's state with it."""
    import ctypes as C
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import state_vec
    mg = np.load(os.path.join(GOLD, "mutator_golden.npz")); sgm = np.load(os.path.join(GOLD, "sim_golden.npz"))
    every = int(mg["every"]); names = [str(n) for n in mg["phys_names"]]

    def vec(o): return np.concatenate([state_vec(o), [float(o.pads[p].is_active) for p in range(34)], [o.pads[p].cooldown for p in range(34)]])

    def run(grp, nc, strip_block):
        starts = [ArenaState.from_buffer_copy(mg[f"phys/{n}/start_raw"].tobytes()) for n in grp]
        env = BatchedEnv(len(grp), nc // 2, mesh=(sgm["mesh_verts"], sgm["mesh_tris"]))
        if strip_block:
            m = type(starts[0].mutators).from_buffer_copy(bytes(starts[0].mutators))
            assert env.lib.rlgpu_env_set_mutators(env.h, C.byref(m)) == 0
            for s in starts: s.hidden.valid &= ~8
        env.upload_states(starts)
        tapes = [mg[f"phys/{n}/tape"] for n in grp]; T = max(len(t) for t in tapes)
        ctl = np.zeros((len(grp), nc, 8), np.float32); compared = 0
        for t in range(T):
            for i, tp in enumerate(tapes):
                if t < len(tp): ctl[i] = tp[t]
            env.set_controls(ctl)
            env.physics_ticks(1)
            if (t + 1) % every == 0:
                cur = env.download_states()
                for i, n in enumerate(grp):
                    if t + 1 <= len(tapes[i]):
                        j = (t + 1) // every - 1
                        assert np.array_equal(vec(cur[i]), mg[f"phys/{n}/states"][j]), f"{n} tick {t + 1}: HIP state is not the reference's"
                        assert cur[i].hidden.ref_engine == int(mg[f"phys/{n}/engines"][j]), f"{n} tick {t + 1}: the engines parted"
                        assert cur[i].hidden.valid & 8 and cur[i].mutators.gravity_z == starts[i].mutators.gravity_z
                        compared += every
        env.close()
        return compared

    compared = 0
    for nc in (2, 4, 6):
        grp = [n for n in names if ArenaState.from_buffer_copy(mg[f"phys/{n}/start_raw"].tobytes()).num_cars == nc]
        compared += run(grp, nc, False)                                   # M1 and M2 side by side in one batch
    assert compared >= 14800, compared
    assert run([n for n in names if n.startswith("M1/") and ArenaState.from_buffer_copy(mg[f"phys/{n}/start_raw"].tobytes()).num_cars == 2], 2, True) >= 2800
    # a fresh batch runs RLConst's defaults, and says so
    env = BatchedEnv(2, 1); st = env.download_states()
    assert st[0].hidden.valid & 8 and st[0].mutators.gravity_z == -650.0 and st[0].mutators.ball_max_speed == 6000.0 and st[0].mutators.flags == 0
    bad = type(st[0].mutators).from_buffer_copy(bytes(st[0].mutators)); bad.ball_damp_per_tick = 0.0
    assert env.lib.rlgpu_env_set_mutators(env.h, C.byref(bad)) != 0       # refused loudly
    env.close()


@pytest.mark.parametrize("team", [1, 2, 3])
def test_hip_state_setters_equal_the_reference_draw_for_draw(team):
    """SURVEY A8 on the GPU, exactly: the KERNEL's RandomState / KickoffState (rlgpu_env_reset with run_setter = 1) drawing from the reference's engine
    (tests/golden/setter_golden.npz: 64 resets of one reference arena per setter and car order, its thread engine started from a known state).  Reset i of
    the reference's arena continues the engine where reset i - 1 left it: here env i of a batch starts from the recorded engine state before reset i, so
    the 64 resets run as ONE launch -- every state EQUAL to the reference's (ball, every car's position, velocity, angular velocity, basis, flags, boost),
    and the engine after it."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from rlgymppo_cpp_amd import _lib
    from rlgymppo_cpp_amd.state import default_arena
    from simlib import state_vec
    sg2 = np.load(os.path.join(GOLD, "setter_golden.npz"))
    for label in ("random_air", "random_ground", "kickoff"):
        for rehash in (0, 11):
            key = f"{label}/{team}/{rehash}"
            want = [ArenaState.from_buffer_copy(b.tobytes()) for b in sg2[key + "/states"]]; engines = [int(e) for e in sg2[key + "/engine_after"]]
            flags = int(sg2[key + "/flags"]); kind = int(sg2[key + "/kind"])
            cfg = _lib.default_gym_config(); cfg.setter_kind = kind
            cfg.rand_ball_speed = flags & 1; cfg.rand_car_speed = (flags >> 1) & 1; cfg.cars_on_ground = (flags >> 2) & 1
            env = BatchedEnv(len(want), team, cfg=cfg)
            starts = []
            for i in range(len(want)):
                st = default_arena(2 * team); st.car_order = want[0].car_order
                st.hidden.valid |= 4; st.hidden.ref_engine = int(sg2[key + "/engine0"]) if i == 0 else engines[i - 1]
                starts.append(st)
            env.upload_states(starts)
            env.reset(True); env.sync()
            got = env.download_states()
            for i in range(len(want)):
                assert np.array_equal(state_vec(got[i]), state_vec(want[i])), f"{key} reset {i}: not the reference's state"
                assert got[i].hidden.ref_engine == engines[i], f"{key} reset {i}: the engines parted"
            env.close()


def test_hip_mesh_of_two_files_vs_reference_golden(tmp_path):
    """The HIP path with a mesh loaded from TWO .cmf files (rlgpu_env_load_cmf_dir keeps one collision object per file, as the reference does:
    Arena.cpp:1028-1054) against the reference's recording on the same two files (tests/golden/seam_golden.npz): all 115 one-tick pairs
    EQUAL, the five tapes bit-identical (simlib.SEAM_EXACT_UNTIL), through the seam in the panel above the goal and in the goal roof."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import write_cmf_parts, state_vec, SEAM_EXACT_UNTIL
    sg = np.load(os.path.join(GOLD, "seam_golden.npz"))
    root = write_cmf_parts(sg["mesh_verts"], sg["mesh_tris"], sg["mesh_parts"], str(tmp_path))
    mesh_dir = os.path.join(root, "soccar")
    B, A = sg["pairs/before"], sg["pairs/after"]
    env = BatchedEnv(len(B), 1, mesh=mesh_dir)
    env.upload_states([ArenaState.from_buffer_copy(b.tobytes()) for b in B])
    env.physics_ticks(1)
    got = env.download_states()
    bad = [i for i in range(len(B)) if not np.array_equal(state_vec(got[i]), state_vec(ArenaState.from_buffer_copy(A[i].tobytes())))]
    assert not bad, f"one-tick pairs {bad[:8]} of the two-file mesh are not bit-equal to the reference"
    env.close()
    names = [str(x) for x in sg["phys_names"]]; every = int(sg["phys_every"])
    env = BatchedEnv(len(names), 1, mesh=mesh_dir)
    env.upload_states([ArenaState.from_buffer_copy(sg[f"phys/{n}/start_raw"].tobytes()) for n in names])
    tapes = [sg[f"phys/{n}/tape"] for n in names]; T = max(len(t) for t in tapes)
    ctl = np.zeros((len(names), 2, 8), np.float32)
    for t in range(T):
        for i, tp in enumerate(tapes):
            if t < len(tp): ctl[i] = tp[t]
        env.set_controls(ctl); env.physics_ticks(1)
        if (t + 1) % every == 0:
            cur = env.download_states()
            for i, n in enumerate(names):
                if t + 1 <= min(SEAM_EXACT_UNTIL.get(n, len(tapes[i])), len(tapes[i])):
                    assert np.array_equal(state_vec(cur[i]), sg[f"phys/{n}/states"][(t + 1) // every - 1]), f"{n} tick {t + 1}: HIP state is not the reference's"
    env.close()


def test_hip_tessellated_mesh_of_sixteen_files_vs_reference_golden(tmp_path):
    """VERDICT r03 4b.  The HIP path on the mesh of bench.py's `mesh_tessellated` leg -- 10 084 triangles in 16 .cmf files, one collision object
    and one contact manifold per file (Arena.cpp:1028-1054), loaded by rlgpu_env_load_cmf_dir -- against recordings of the reference on the same
    16 files (tests/golden/tess_golden.npz).  This is the workload that reaches the device-only mesh machinery at size: BVH top levels staged
    in LDS, frontiers of up to 128 nodes, the kept candidate leaves, up to 24 leaves per body, two mesh manifolds per body.  EQUALITY with the
    reference: all 360 one-tick pairs (79 of them touch two or more files at once; a tape's pairs in one env slot one after the other, as
    recorded), and the six kickoff tapes of 1v1 / 2v2 / 3v3 under random controls over their whole 1 200 ticks, every field of every body."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import write_cmf_parts, state_vec
    tg = np.load(os.path.join(GOLD, "tess_golden.npz"))
    root = write_cmf_parts(tg["mesh_verts"], tg["mesh_tris"], tg["mesh_parts"], str(tmp_path))
    mesh_dir = os.path.join(root, "soccar")
    names = [str(x) for x in tg["phys_names"]]; every = int(tg["phys_every"])
    B, A, T = tg["pairs/before"], tg["pairs/after"], tg["pairs/tag"]
    n_pairs = 0
    for team in (1, 2, 3):
        nc = 2 * team
        mine = [i for i, n in enumerate(names) if n.startswith(f"{team}v{team}")]
        # one-tick pairs
        per_tape = {ti: [i for i in range(len(B)) if int(T[i][0]) == ti] for ti in mine}
        env = BatchedEnv(len(mine), team, mesh=mesh_dir)
        for j in range(max(len(v) for v in per_tape.values())):
            slots = [e for e, ti in enumerate(mine) if j < len(per_tape[ti])]
            env.upload_states([ArenaState.from_buffer_copy(B[per_tape[mine[e]][j]].tobytes()) for e in slots], env_ids=slots)
            env.physics_ticks(1)
            got = env.download_states(env_ids=slots)
            for e, st in zip(slots, got):
                i = per_tape[mine[e]][j]
                assert np.array_equal(state_vec(st), state_vec(ArenaState.from_buffer_copy(A[i].tobytes()))), \
                    f"{names[mine[e]]} tick {int(T[i][1])} ({int(T[i][2])} mesh files touched): one tick on the HIP path is not the reference's"
                n_pairs += 1
        env.close()
        # free-running tapes
        env = BatchedEnv(len(mine), team, mesh=mesh_dir)
        env.upload_states([ArenaState.from_buffer_copy(tg[f"phys/{names[ti]}/start_raw"].tobytes()) for ti in mine])
        tapes = [tg[f"phys/{names[ti]}/tape"] for ti in mine]
        ctl = np.zeros((len(mine), nc, 8), np.float32)
        for t in range(len(tapes[0])):
            for e, tp in enumerate(tapes): ctl[e] = tp[t]
            env.set_controls(ctl); env.physics_ticks(1)
            if (t + 1) % every == 0:
                cur = env.download_states()
                for e, ti in enumerate(mine):
                    assert np.array_equal(state_vec(cur[e]), tg[f"phys/{names[ti]}/states"][(t + 1) // every - 1]), f"{names[ti]} tick {t + 1}: HIP state is not the reference's"
        env.close()
    assert n_pairs == len(B) == 360


def test_hip_wedge_fixture_contacts_beyond_the_lds_layout_vs_reference_golden(tmp_path):
    """no contact point is ever dropped.  tests/golden/wedge_golden.npz (make_wedge_golden.py) holds the reference on the
    tessellated arena dealt round-robin into 16 .cmf files -- neighbouring triangles in different files, so a car on a fillet holds points in
    three, four, five (once: fourteen) mesh manifolds at once, where the env's LDS-resident contact layout has two -- 1v1 / 2v2 / 3v3 under random
    controls and two six-car pile-ups.  Round 4's stepper dropped the points beyond the second manifold and left the reference on three of the
    five tapes (ticks 830 / 560 / 570); now such a tick is redone with the big layout (rlgpu_env.hip:tick_world_big): all five tapes EQUAL to
    the reference over their whole length, every field of every body, with > 100 env-ticks redone and nothing lost."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import write_cmf_parts, state_vec
    wg = np.load(os.path.join(GOLD, "wedge_golden.npz"))
    root = write_cmf_parts(wg["mesh_verts"], wg["mesh_tris"], wg["mesh_parts"], str(tmp_path))
    mesh_dir = os.path.join(root, "soccar")
    names = [str(x) for x in wg["phys_names"]]; every = int(wg["phys_every"])
    redone = 0
    for name in names:
        team = int(name[0]); nc = 2 * team
        env = BatchedEnv(1, team, mesh=mesh_dir)
        before = env.big_layout_ticks()   # (process-wide counters: differences, so that a session's totals stay readable)
        env.upload_states([ArenaState.from_buffer_copy(wg[f"phys/{name}/start_raw"].tobytes())])
        tape = wg[f"phys/{name}/tape"]; want = wg[f"phys/{name}/states"]
        for t in range(len(tape)):
            env.set_controls(tape[t][None]); env.physics_ticks(1)
            if (t + 1) % every == 0:
                assert np.array_equal(state_vec(env.download_states()[0]), want[(t + 1) // every - 1]), f"{name} tick {t + 1}: HIP state is not the reference's"
        redone += env.big_layout_ticks() - before; assert env.lost_contact_count() == 0
        env.close()
    print("wedge fixture: env-ticks redone with the big layout:", redone)
    if "tiny" not in os.environ.get("RLGPU_LIB", ""): assert 100 < redone < 200    # (the host build of the same layout: 111)


def test_hip_continues_a_mid_episode_reference_state_with_its_hidden_state():
    """the hidden-state read-out, rebuilt.  tests/golden/midtape_golden.npz: ten mid-episode states of the reference with the
    arena's hidden state (btRSBroadphase's cell and arrival rank of every dynamic proxy, read from its cell lists by oracle/ref_driver.cpp) and
    the reference's continuation.  Uploaded into a FRESH env slot and into a USED one (a slot that has been ticking another game: its own history
    must give way), the HIP path continues bit for bit like the arena the state came from, every field of every body to the end of the tape; the
    same state with hidden.valid = 0 -- a fresh arena set to it, as before round 5 -- leaves the reference on the two cuts inside the six-car heap.
    Then device to device: a state downloaded while a car lies demolished carries the wreck's own basis (it has turned away from the reported
    one); uploaded into another slot and back into its own, both continue alike."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import state_vec
    g = np.load(os.path.join(GOLD, "midtape_golden.npz"))
    every = int(g["every"]); told_apart = []
    for name in [str(x) for x in g["names"]]:
        st = ArenaState.from_buffer_copy(g[f"cut/{name}/state"].tobytes()); nc = st.num_cars
        bare = ArenaState.from_buffer_copy(g[f"cut/{name}/state"].tobytes()); bare.hidden.valid = 0
        tape = g[f"cut/{name}/tape"]; want = g[f"cut/{name}/states"]
        env = BatchedEnv(3, nc // 2)
        # slot 1 plays something else first
        other = default_arena(nc); env.upload_states([other], env_ids=[1])
        ctl = np.zeros((3, nc, 8), np.float32); ctl[1, :, 0] = 1.0; ctl[1, :, 6] = 1.0
        env.set_controls(ctl); env.physics_ticks(90)
        env.upload_states([st, st, bare], env_ids=[0, 1, 2])
        same_bare = True
        for t in range(len(tape)):
            ctl[:] = tape[t][None]; env.set_controls(ctl); env.physics_ticks(1)
            if (t + 1) % every == 0:
                cur = env.download_states(); w = want[(t + 1) // every - 1]
                assert np.array_equal(state_vec(cur[0]), w), f"{name} + {t + 1} ticks, fresh slot: HIP state is not the reference's"
                assert np.array_equal(state_vec(cur[1]), w), f"{name} + {t + 1} ticks, used slot: HIP state is not the reference's"
                same_bare = same_bare and np.array_equal(state_vec(cur[2]), w)
        if not same_bare: told_apart.append(name)
        env.close()
    assert told_apart == ["3v3_kickoff@280", "3v3_kickoff@300"], told_apart
    # a wreck's basis
    sg = np.load(os.path.join(GOLD, "sim_golden.npz"))
    s0 = ArenaState.from_buffer_copy(sg["phys/demo_and_respawn/start_raw"].tobytes()); tape = sg["phys/demo_and_respawn/tape"]; nc = s0.num_cars
    env = BatchedEnv(2, nc // 2); env.upload_states([s0], env_ids=[0])
    ctl = np.zeros((2, nc, 8), np.float32); t = 0; mid = None
    while t < len(tape) and mid is None:
        ctl[0] = tape[t]; env.set_controls(ctl); env.physics_ticks(1); t += 1
        cur = env.download_states(env_ids=[0])[0]
        wrecks = [k for k in range(nc) if cur.cars[k].flags & (1 << 13)]
        if wrecks and cur.cars[wrecks[0]].demo_respawn_timer < 2.5: mid = cur
    assert mid is not None and (mid.hidden.valid & 3) == 3
    k = wrecks[0]
    assert any(abs(mid.hidden.wreck_rot[k][q] - mid.cars[k].rot[q]) > 1e-4 for q in range(9)), "the wreck's body has not turned away from the reported rotation"
    env.upload_states([mid, mid], env_ids=[0, 1])
    for t2 in range(t, len(tape)):
        ctl[0] = tape[t2]; ctl[1] = tape[t2]; env.set_controls(ctl); env.physics_ticks(1)
    a, b = env.download_states()
    assert bytes(a) == bytes(b)
    env.close()


def test_live_reference_rollout(ref_lib, port_lib):
    """When the prebuilt reference .so travelled with the snapshot: step the real RLGymSim_CPP Gym on the host CPU next
    to the GPU env from the same state and action tape."""
    from simlib import RefGym
    from rlgymppo_cpp_amd.env import BatchedEnv
    from rlgymppo_cpp_amd import _lib
    dev = torch.device("cuda", 0)
    g = RefGym(ref_lib, 1, 8)
    s0 = default_arena(2); s0.ball.pos[:] = (500, -800, 93.15); s0.ball.vel[:] = (300, 100, 0); s0.cars[0].boost = 80
    obs_r = g.reset_to(s0)
    st = ref_lib.get_state(g.arena())
    env = BatchedEnv(1, 1, cfg=_lib.default_gym_config(), mesh=port_lib.mesh)
    env.upload_states([st])
    obs0 = env.reset(False)
    assert np.abs(obs0.cpu().numpy() - obs_r).max() < 1e-5
    nobs = torch.empty((2, 89), device=dev); r = torch.empty(2, device=dev); d = torch.empty(2, dtype=torch.int32, device=dev)
    rng = np.random.RandomState(11)
    for t in range(30):
        # A LIVE comparison on whatever CPU the GPU box has: the reference normalises with rsqrtss, whose table belongs to the CPU vendor, and the
        # stepper restates the table of the CPU the fixtures were recorded on (csrc/rl_math.h) -- so this test cannot ask for equality (the
        # bit-exact ones are the recorded fixtures: sim_golden / sim_steps / gym_golden / tess_golden / gameinst_golden, all equal on the HIP path).
        # Ground actions only (table rows 0..23), so that a last-bit difference is not amplified by a landing.
        a = rng.randint(0, 24, size=2).astype(np.int32)
        o_r, r_r, d_r, _ = g.step(a)
        env.step(torch.from_numpy(a).to(dev), nobs, r, d); env.sync()
        assert int(d[0]) == d_r
        assert np.abs(nobs.cpu().numpy() - o_r).max() < 2e-3 and np.abs(r.cpu().numpy() - r_r).max() < 2e-3


def test_full_size_properties():
    """BASELINE size (4096 envs 1v1): determinism, finiteness, bounds, episode accounting."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    dev = torch.device("cuda", 0)
    outs = []
    for rep in range(2):
        env = BatchedEnv(4096, 1)
        obs = env.reset(True)
        nobs = torch.empty_like(obs); rew = torch.empty(8192, device=dev); done = torch.empty(8192, dtype=torch.int32, device=dev)
        g = torch.Generator(device="cpu").manual_seed(3)
        tot_done = 0
        for t in range(40):
            a = torch.randint(0, 90, (8192,), generator=g, dtype=torch.int32).to(dev)
            env.step(a, nobs, rew, done)
            tot_done += int(done.sum().item())
        env.sync()
        o = nobs.cpu().numpy()
        assert np.isfinite(o).all() and np.isfinite(rew.cpu().numpy()).all()
        assert np.abs(o[:, 0:3]).max() < 1.3          # ball position / arena extents
        assert set(np.unique(o[:, 17:51])) <= {0.0, 1.0}  # pads are 0/1
        assert (done.view(4096, 2)[:, 0] == done.view(4096, 2)[:, 1]).all()  # done shared by an env's players
        outs.append((o.copy(), rew.cpu().numpy().copy(), tot_done))
        env.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]  # bitwise deterministic


# ---- learner kernels -----------------------------------------------------------------------------------------------
def _mk_core(lg, bf16=False, max_rows=4096):
    from rlgymppo_cpp_amd.ppo import PPOCore
    D, A, H = int(lg["D"]), int(lg["A"]), int(lg["H"])
    core = PPOCore(D, A, (H, H), (H, H), policy_lr=float(lg["adam_lr"]), critic_lr=float(lg["adam_lr"]), ent_coef=float(lg["ent_coef"]),
                   clip_range=float(lg["clip"]), use_bf16=bf16, max_rows=max_rows)
    core.set_params(lg["pol_params"], 0); core.set_params(lg["cri_params"], 1)
    return core


def test_policy_value_forward_and_sampling(lg):
    dev = torch.device("cuda", 0)
    core = _mk_core(lg)
    obs = torch.from_numpy(lg["obs"]).to(dev)
    p = core.probs(obs).cpu().numpy()
    assert np.abs(p - lg["probs"]).max() < 2e-6
    v = core.value(obs).cpu().numpy()
    assert np.abs(v - lg["values"]).max() < 2e-5
    n = obs.shape[0]
    acts = torch.empty(n, dtype=torch.int32, device=dev); logp = torch.empty(n, device=dev)
    core.act(obs, acts, logp, noise=torch.from_numpy(lg["q"]).to(dev))
    a = acts.cpu().numpy()
    mism = np.nonzero(a != lg["actions"])[0]
    margins = R.top2_margin(lg["probs"], lg["q"])
    # bit-exact action indices on the recorded noise tape; a mismatch is tolerated only inside fp32 summation-order noise
    assert all(margins[i] < 1e-5 for i in mism), f"sampled actions differ at rows {mism} with margins {margins[mism]}"
    ok = a == lg["actions"]
    assert np.abs(logp.cpu().numpy()[ok] - lg["logp"][ok]).max() < 1e-5
    core.act(obs, acts, logp, deterministic=True)
    assert (acts.cpu().numpy() == lg["det_actions"]).all() and (logp.cpu().numpy() == 0).all()
    # own Philox sampler: empirical frequencies follow probs
    big = obs[:1].repeat(4096, 1).contiguous()
    a2 = torch.empty(4096, dtype=torch.int32, device=dev); l2 = torch.empty(4096, device=dev)
    counts = np.zeros(int(lg["A"]))
    for _ in range(8):
        core.act(big, a2, l2)
        counts += np.bincount(a2.cpu().numpy(), minlength=int(lg["A"]))
    freq = counts / counts.sum()
    assert np.abs(freq - lg["probs"][0]).max() < 0.01


def test_gae_kernel(lg):
    dev = torch.device("cuda", 0)
    core = _mk_core(lg)
    # the golden 1-D batch as ONE trajectory (n = 1)
    B = len(lg["gae_rews"])
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    adv, tg, rt = core.gae(t(lg["gae_rews"]).view(B, 1), t(lg["gae_terminal"]).view(B, 1), t(lg["gae_truncated"]).view(B, 1), t(lg["gae_values"]).view(B + 1, 1),
                           float(lg["gae_gamma"]), float(lg["gae_lambda"]), float(lg["gae_ret_std"]), float(lg["gae_clip"]), 0)
    assert np.abs(adv.cpu().numpy()[:, 0] - lg["gae_adv"]).max() <= 1e-4
    assert np.abs(rt.cpu().numpy()[:, 0] - lg["gae_returns"]).max() <= 1e-4
    assert np.abs(tg.cpu().numpy()[:, 0] - lg["gae_targets"]).max() <= 1e-4
    # many trajectories, time-major, vs the serial oracle on the agent-major concatenation (reference quirk Q1 included)
    rng = np.random.RandomState(1); T, n = 32, 300
    rews = rng.randn(T, n).astype(np.float32); dones = (rng.rand(T, n) < 0.05).astype(np.float32); vals = rng.randn(T + 1, n).astype(np.float32)
    truncs = np.zeros((T, n), np.float32); truncs[T - 1] = 1 - dones[T - 1]
    adv, tg, rt = core.gae(t(rews), t(dones), t(truncs), t(vals), 0.99, 0.95, 2.0, 10.0, 0)
    cat = lambda x: x.T.reshape(-1)
    cvals = np.concatenate([cat(vals[:T]), vals[T, n - 1:n]])
    a_o, t_o, r_o = R.compute_gae(cat(rews), cat(dones), cat(truncs), cvals, 0.99, 0.95, 2.0, 10.0)
    assert np.abs(cat(adv.cpu().numpy()) - a_o).max() <= 1e-4 and np.abs(cat(rt.cpu().numpy()) - r_o).max() <= 1e-4 and np.abs(cat(tg.cpu().numpy()) - t_o).max() <= 1e-4


def test_ppo_minibatch_grads_and_adam(lg):
    dev = torch.device("cuda", 0)
    core = _mk_core(lg)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    obs = t(lg["obs"]); acts = t(lg["actions"].astype(np.int32)); olp = t(lg["old_logp"]); adv = t(lg["adv"]); tg = t(lg["targets"])
    metrics = torch.zeros(8, device=dev)
    core.zero_grads()
    core.minibatch(obs, acts, olp, adv, tg, None, obs.shape[0], float(lg["scale"]), metrics)
    core.sync()
    gp, gc = core.get_grads(0), core.get_grads(1)
    assert np.abs(gp - lg["pol_grads"]).max() < 5e-6 * max(1.0, np.abs(lg["pol_grads"]).max() * 100)
    assert np.abs(gc - lg["cri_grads"]).max() < 5e-6 * max(1.0, np.abs(lg["cri_grads"]).max() * 100)
    m = metrics.cpu().numpy(); n = obs.shape[0]
    assert abs(m[0] / n - float(lg["entropy"])) < 1e-4 and abs(m[1] / n - float(lg["kl"])) < 1e-5
    assert abs(m[2] / n - float(lg["clip_fraction"])) < 1e-6 and abs(m[4] / n - float(lg["value_loss"])) < 1e-4
    # gathered minibatch (shuffled indices) gives the same gradients
    perm = np.random.RandomState(0).permutation(n).astype(np.int32)
    core.zero_grads()
    core.minibatch(obs, acts, olp, adv, tg, t(perm), n, float(lg["scale"]), None)
    assert np.abs(core.get_grads(0) - gp).max() < 1e-5
    # accumulation over two half minibatches == one full minibatch (PPOLearner.cpp:127 batchSizeRatio)
    core.zero_grads()
    h = n // 2
    core.minibatch(obs, acts, olp, adv, tg, t(np.arange(0, h, dtype=np.int32)), h, float(lg["scale"]) * h / n, None)
    core.minibatch(obs, acts, olp, adv, tg, t(np.arange(h, n, dtype=np.int32)), n - h, float(lg["scale"]) * (n - h) / n, None)
    assert np.abs(core.get_grads(0) - gp).max() < 1e-5 and np.abs(core.get_grads(1) - gc).max() < 1e-5
    # clip + Adam: three steps with the golden fake gradients
    gt = core.grad_tensor()
    npol = core.num_params(0)
    for s, (gk, pk) in enumerate([("adam_g0", "adam_p1"), ("adam_g1", "adam_p2"), ("adam_g2", "adam_p3")]):
        gt.zero_(); gt[:npol] = t(lg[gk])
        core.clip_adam_step(0.5, 1.0)
        core.sync()
        assert np.abs(core.get_params(0) - lg[pk]).max() < 1e-6, f"adam step {s}"


def test_bf16_path_close_to_fp32(lg):
    dev = torch.device("cuda", 0)
    c32, c16 = _mk_core(lg), _mk_core(lg, bf16=True)
    obs = torch.from_numpy(lg["obs"]).to(dev)
    p32, p16 = c32.probs(obs).cpu().numpy(), c16.probs(obs).cpu().numpy()
    assert np.abs(p32 - p16).max() < 5e-3      # bf16 operands (8-bit mantissa), fp32 accumulation
    assert np.abs(c32.value(obs).cpu().numpy() - c16.value(obs).cpu().numpy()).max() < 5e-2


def test_bf16_fast_path_gradients_and_ragged_shapes():
    """bf16 fast path (NT / transposing-read TN GEMMs, bf16 activations) against the exact fp32 path on the flagship layer
    shapes with ragged edges: K = 89, N = 90 / 1, rows not a multiple of any tile.  Tolerances: bf16 has an 8-bit mantissa;
    operands are rounded once per layer, sums are fp32."""
    from rlgymppo_cpp_amd.ppo import PPOCore
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(3)
    n = 1037
    mk = lambda bf: PPOCore(89, 90, (256, 256, 256), (256, 256, 256), use_bf16=bf, max_rows=1100, seed=7)
    c32, c16 = mk(False), mk(True)
    c16.set_params(c32.get_params(0), 0); c16.set_params(c32.get_params(1), 1)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    obs = t(rng.randn(n, 89).astype(np.float32))
    p32, p16 = c32.probs(obs).cpu().numpy(), c16.probs(obs).cpu().numpy()
    assert np.abs(p32 - p16).max() < 5e-3
    v32, v16 = c32.value(obs).cpu().numpy(), c16.value(obs).cpu().numpy()
    assert np.abs(v32 - v16).max() < 5e-2
    acts = t(rng.randint(0, 90, n).astype(np.int32)); olp = t(np.log(p32[np.arange(n), acts.cpu().numpy()]).astype(np.float32) + 0.05 * rng.randn(n).astype(np.float32))
    adv = t(rng.randn(n).astype(np.float32)); tg = t(rng.randn(n).astype(np.float32))
    perm = t(rng.permutation(n).astype(np.int32))
    grads = []
    for c in (c32, c16):
        m = torch.zeros(8, device=dev)
        c.zero_grads(); c.minibatch(obs, acts, olp, adv, tg, perm, n, 1.0, m); c.sync()
        grads.append((c.get_grads(0), c.get_grads(1), m.cpu().numpy()))
    for k in (0, 1):
        ref, got = grads[0][k], grads[1][k]
        assert np.isfinite(got).all()
        assert np.abs(ref - got).max() < 3e-2 * np.abs(ref).max() + 1e-6, (k, np.abs(ref - got).max(), np.abs(ref).max())
        cos = float(np.dot(ref, got) / (np.linalg.norm(ref) * np.linalg.norm(got) + 1e-30))
        assert cos > 0.999, (k, cos)
    assert np.allclose(grads[0][2][:5], grads[1][2][:5], rtol=2e-2, atol=2e-2 * n)
    # weights change -> shadows refresh: one Adam step keeps the two paths together
    for c in (c32, c16):
        c.clip_adam_step(0.5, 1.0); c.sync()
    assert np.abs(c32.get_params(0) - c16.get_params(0)).max() < 2e-3
    assert np.abs(c32.probs(obs).cpu().numpy() - c16.probs(obs).cpu().numpy()).max() < 1e-2


def test_gemm_shapes_against_numpy():
    """The MFMA GEMM at the flagship layer shapes incl. ragged edges (K=89, N=90, N=1, rows not a tile multiple)."""
    from rlgymppo_cpp_amd.ppo import PPOCore
    dev = torch.device("cuda", 0)
    core = PPOCore(89, 90, (256, 256, 256), (256, 256, 256), max_rows=1100)
    rng = np.random.RandomState(0)
    x = rng.randn(1037, 89).astype(np.float32)
    pshapes, cshapes = core.layer_shapes(0), core.layer_shapes(1)
    out, _ = R.mlp_forward(core.get_params(0), pshapes, x)
    p = core.probs(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.abs(p - R.policy_probs(out)).max() < 1e-5
    v, _ = R.mlp_forward(core.get_params(1), cshapes, x)
    assert np.abs(core.value(torch.from_numpy(x).to(dev)).cpu().numpy() - v[:, 0]).max() < 1e-4
    assert core.num_params(0) == 177754 and core.num_params(1) == 154881   # verified parameter counts


@pytest.mark.gpu
def test_experience_fifo_keeps_older_iterations_resident():
    """Learner with expBufferSize = 2.5 iterations (ExperienceBuffer.cpp:17-68): every live iteration's rows sit bit-exact in their
    device slot, the FIFO size follows min(max, k*B), and each epoch makes curSize // batchSize optimizer steps."""
    from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
    n_envs, T = 32, 8
    B = n_envs * 2 * T
    cap = B * 5 // 2
    cfg = LearnerConfig(numEnvs=n_envs, teamSize=1, timestepsPerIteration=B, expBufferSize=cap, randomSeed=5,
                        ppo=PPOLearnerConfig(batchSize=B // 2, miniBatchSize=B // 4, epochs=2, autocastLearn=False))
    L = Learner(cfg)
    assert L.fifo.num_slots == 4
    kept = {}
    for it in range(5):
        before = L.cumulative_model_updates
        L.collect()
        L.add_new_experience()
        cur = L.fifo.size()
        assert cur == min(cap, (it + 1) * B)
        snap = (L.obs_buf[:T].reshape(B, -1).clone(), L.act_buf.view(-1).clone(), L.logp_buf.view(-1).clone(), L.adv.view(-1).clone(), L.tgt.view(-1).clone())
        # which slot did it go to: the one whose contents now equal this iteration's rows
        hits = [s for s in range(L.fifo.num_slots) if torch.equal(L.ex_obs[s * B:(s + 1) * B], snap[0])]
        assert len(hits) == 1
        kept = {k: v for k, v in kept.items() if v[0] != hits[0]}
        kept[it] = (hits[0], snap)
        L.learn()
        assert L.cumulative_model_updates - before == 2 * (cur // (B // 2))
        # older iterations that are still (partly) alive were not disturbed by this one
        for k, (s, sn) in kept.items():
            if k > it - 3:
                r = slice(s * B, (s + 1) * B)
                assert torch.equal(L.ex_obs[r], sn[0]) and torch.equal(L.ex_act[r], sn[1]) and torch.equal(L.ex_logp[r], sn[2])
                assert torch.equal(L.ex_adv[r], sn[3]) and torch.equal(L.ex_tgt[r], sn[4])
    rep = L.finish_report()
    assert np.isfinite(rep["Policy Entropy"]) and np.isfinite(rep["Value Function Loss"])


@pytest.mark.gpu
def test_collection_during_learn_mode():
    """LearnerConfig.collectionDuringLearn: the PPO epochs run on their own stream; the FIFO slot of the next iteration is not written
    before the epochs reading the old rows are done, every iteration still makes its optimizer steps, parameters stay finite."""
    from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
    n_envs, T = 32, 8
    B = n_envs * 2 * T
    cfg = LearnerConfig(numEnvs=n_envs, teamSize=1, timestepsPerIteration=B, expBufferSize=B, randomSeed=11, collectionDuringLearn=True,
                        ppo=PPOLearnerConfig(batchSize=B, miniBatchSize=B // 2, epochs=2, autocastLearn=True))
    L = Learner(cfg)
    assert L.s_learn is not None
    p0 = L.ppo.get_params(2).copy()
    for it in range(4):
        L.iteration()
        assert L.cumulative_model_updates == 2 * (it + 1)
    rep = L.finish_report()
    torch.cuda.synchronize()
    p1 = L.ppo.get_params(2)
    assert np.isfinite(p1).all() and np.abs(p1 - p0).max() > 0
    assert np.isfinite(rep["Policy Entropy"]) and 0 < rep["Policy Entropy"] < np.log(90) + 1e-3
    # the slot holding the last iteration is intact (nothing wrote into it after its copy)
    assert torch.equal(L.ex_act[:B] if torch.equal(L.ex_act[:B], L.act_buf.view(-1)) else L.ex_act[B:2 * B], L.act_buf.view(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("team_size,n_envs,tess", [(1, 700, False), (2, 300, False), (3, 171, False), (1, 260, True)])
def test_no_kernel_writes_past_a_device_buffer(team_size, n_envs, tess, monkeypatch):
    """RLGPU_REDZONE: 64 KB of guard bytes behind every persistent device buffer of the batch; after resets, steps, physics ticks, uploads, downloads,
    snapshots, lockstep and free-running collection launches every guard byte is what it was.  (The staged word rows of rounds 3-4 wrote
    NC x n_envs words behind the resident state: this test fails on that build with "redzone of 'resident state words' overwritten".)"""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from rlgymppo_cpp_amd.ppo import PPOCore
    from rlgymppo_cpp_amd import _lib
    monkeypatch.setenv("RLGPU_REDZONE", "65536")
    dev = torch.device("cuda", 0); T = 6
    cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 7
    mesh = "procedural"
    if tess:
        import bench
        mesh = os.path.join(bench.make_tessellated_mesh_dir()[0], "soccar")
    env = BatchedEnv(n_envs, team_size, cfg=cfg, mesh=mesh)
    N, D = env.n_agents, env.obs_size
    core = PPOCore(D, env.n_actions, (64, 64), (64, 64), use_bf16=True, max_rows=max(4096, N))
    obs = torch.zeros((T + 1, N, D), device=dev); acts = torch.zeros((T, N), dtype=torch.int32, device=dev); logp = torch.zeros((T, N), device=dev)
    rew = torch.zeros((T, N), device=dev); done = torch.zeros((T, N), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    env.reset(True, obs[0]); env.sync(); env.check_redzones()
    env.enable_snapshots(True)
    for t in range(3):
        core.act(obs[0], acts[0], logp[0]); core.sync(); env.step(acts[0], obs[1], rew[0], done[0]); env.sync()
    env.check_redzones()
    st = env.download_states(); env.upload_states(st); env.physics_ticks(3); env.sync(); env.check_redzones()
    env.enable_snapshots(False)
    for k in range(2):
        assert env.collect(core, T, obs, acts, logp, rew, done); env.sync()
        obs[0].copy_(obs[T]); torch.cuda.synchronize()
    env.check_redzones()
    steps = torch.zeros(n_envs, dtype=torch.int32, device=dev); torch.cuda.synchronize()
    assert env.collect_free(core, T, (T // 2) * N, obs, acts, logp, rew, done, steps); env.sync()
    env.check_redzones()
    # (the counters next to the kernels: a read with reset hands back what was counted and leaves zero)
    lost = env.lost_contact_count(reset=True); assert lost == 0 and env.lost_contact_count() == 0   # (no contact point is ever dropped)
    big = env.big_layout_ticks(reset=True); assert big >= 0 and env.big_layout_ticks() == 0
    ovf = env.overflow_counts(reset=True); assert len(ovf) == 5 and env.overflow_counts() == [0, 0, 0, 0, 0]
    epa = env.epa_counts(reset=True); assert len(epa) == 2 and env.epa_counts() == [0, 0]
    core.check_redzones()
    env.reset(True, obs[0]); env.sync()
    env.check_redzones()
    # and the checker does see a stray store: one byte, 40 000 bytes behind the first buffer
    assert env.lib.rlgpu_env_debug_overrun(env.h, 0, 40000) == 0
    with pytest.raises(_lib.RlgpuError, match="redzone of 'resident state words'.*first at \\+40000"):
        env.check_redzones()
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("team_size,n_envs", [(1, 512), (1, 1024), (2, 256), (3, 171), (1, 4096), (2, 8192), (3, 16384)])
def test_identical_collection_flows_are_identical(team_size, n_envs, monkeypatch):
    """Two env batches under the same flow of fused collection launches (sampler rewound in between): outputs AND the downloaded resident states are
    equal byte for byte after every launch.  Regression for the staged word rows that ran NC x n_envs words past the resident allocation
    (csrc/arena_io.h arena_num_words): with batches this size the rows no longer fit the page slack and overwrote the action table behind it,
    and one of two identical flows ended launches with car controls that are no rows of the table (tools/determinism/).  The last three cases are
    BASELINE configs[1], [3] and [4] at their full env counts (two batches each), with guard bytes behind every device buffer (RLGPU_REDZONE)."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    monkeypatch.setenv("RLGPU_REDZONE", "65536")
    from rlgymppo_cpp_amd.ppo import PPOCore
    from rlgymppo_cpp_amd import _lib
    dev = torch.device("cuda", 0); CAP = 12
    cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 9
    ea, eb = BatchedEnv(n_envs, team_size, cfg=cfg), BatchedEnv(n_envs, team_size, cfg=cfg)
    v, st = C.c_int(0), C.c_int(0)
    assert ea.lib.rlgpu_state_word_counts(team_size, C.byref(v), C.byref(st)) == 0 and v.value == st.value == ea.state_words()
    core = PPOCore(ea.obs_size, ea.n_actions, (64, 64), (64, 64), use_bf16=True, max_rows=max(4096, ea.n_agents))
    N, D = ea.n_agents, ea.obs_size
    def bufs():
        return (torch.zeros((CAP + 1, N, D), device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev), torch.zeros((CAP, N), device=dev),
                torch.full((CAP, N), -777.0, device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev))
    A, B = bufs(), bufs(); torch.cuda.synchronize()
    ea.reset(True, A[0][0]); eb.reset(True, B[0][0]); ea.sync(); eb.sync()
    for k in range(3 if n_envs <= 1024 else 2):
        s, c = core.get_sampler()
        assert ea.collect(core, CAP, *A); ea.sync()
        core.set_sampler(s, c)
        assert eb.collect(core, CAP, *B); eb.sync()
        for name, x, y in zip(("obs", "actions", "logp", "reward", "done"), A, B):
            assert torch.equal(x, y), (k, name)
        sa, sb = ea.download_states(), eb.download_states()
        bad = [e for e in range(n_envs) if bytes(sa[e]) != bytes(sb[e])]
        assert not bad, (k, bad[:8])
        A[0][0].copy_(A[0][CAP]); B[0][0].copy_(B[0][CAP]); torch.cuda.synchronize()
    ea.check_redzones(); eb.check_redzones(); core.check_redzones()
    ea.close(); eb.close(); core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("team_size,n_envs,bf16", [(1, 70, True), (2, 21, True), (2, 20, True), (3, 9, True), (1, 70, False), (2, 21, False), (2, 20, False), (3, 9, False)])
def test_fused_collection_equals_alternating_act_and_step(team_size, n_envs, bf16):
    """rlgpu_collect (T x (inference + gym step) in one launch, every wavefront on its own envs) against T alternations of
    rlgpu_policy_act / rlgpu_env_step from the same seeds: observations, actions, rewards, dones and the final env state are
    bit-identical, log-probs within one ulp.  Env counts that leave the last wavefront partly empty; 1v1 / 2v2 / 3v3 = 4 / 2 / 1 envs per wavefront.
    bf16 = False: the exact-parity mode -- fp32 operands through v_mfma_f32_32x32x2_f32 inside the collection kernel (wave_infer_f32), against
    the fp32 batched path whose sampled action indices are pinned bit-exact to the reference's DiscretePolicy (tests/test_ref_learner.py)."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from rlgymppo_cpp_amd.ppo import PPOCore
    from rlgymppo_cpp_amd import _lib
    T = 12
    dev = torch.device("cuda", 0)
    out = []
    for fused in (False, True):
        cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 5; cfg.seed_lo = 31
        env = BatchedEnv(n_envs, team_size, cfg=cfg)
        N, D = env.n_agents, env.obs_size
        ppo = PPOCore(D, env.n_actions, (256, 256, 256), (64,), use_bf16=bf16, max_rows=max(N, 64), seed=5)
        obs = torch.zeros((T + 1, N, D), device=dev); acts = torch.zeros((T, N), dtype=torch.int32, device=dev)
        logp = torch.zeros((T, N), device=dev); rew = torch.zeros((T, N), device=dev); done = torch.zeros((T, N), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()   # (the zero fills run on torch's stream, the library writes on its own: without this a fill can land on top of obs[0])
        env.reset(True, obs[0])
        if fused:
            assert env.collect(ppo, T, obs, acts, logp, rew, done)
        else:
            for t in range(T):
                # the learner and the env batch work on streams of their own: every hand-over is synchronised.  (When this test failed behind the
                # learner tests in round 4 it was not this: the staged word rows ran past the resident state, csrc/arena_io.h arena_num_words.)
                ppo.act(obs[t], acts[t], logp[t]); ppo.sync()
                env.step(acts[t], obs[t + 1], rew[t], done[t]); env.sync()
        env.sync()
        # one more sequential step from both: the resident env state and the sampler counter moved identically
        a2 = torch.zeros(N, dtype=torch.int32, device=dev); l2 = torch.zeros(N, device=dev); o2 = torch.zeros((N, D), device=dev)
        r2 = torch.zeros(N, device=dev); d2 = torch.zeros(N, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ppo.act(obs[T], a2, l2); ppo.sync(); env.step(a2, o2, r2, d2); env.sync()
        out.append([x.cpu().numpy() for x in (obs, acts, logp, rew, done, a2, l2, o2, r2, d2)])
    names = ("obs", "actions", "logp", "reward", "done", "next actions", "next logp", "next obs", "next reward", "next done")
    for a, b, name in zip(out[0], out[1], names):
        if "logp" in name:
            # the head is compiled into two translation units; the stepper's one forbids fp contraction, and the compiler's expansion of
            # logf / expf honours that -- the log-probs agree to the last bit or the one before
            assert np.abs(a - b).max() < 1e-6, name
        else:
            assert np.array_equal(a, b), name
    assert out[0][4].sum() > 0 and len(np.unique(out[0][1])) > 20      # episodes ended (auto-resets inside the launch) and actions vary


# ---- the BASELINE configs beyond the headline, at their full sizes ------------------------------------------
def test_config3_full_size_2v2_padded_zero_sum():
    """BASELINE config[3]: 8192 envs 2v2, DefaultOBSPadded + ZeroSumReward, through the Learner's own loop (fused collection) --
    size-independent properties: zero-sum rewards add up to zero per env, rows finite and bounded, the padded rows keep the
    DefaultOBS prefix layout, done is shared by the four players of an env, the optimizer steps happen."""
    from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
    from rlgymppo_cpp_amd import _lib
    n_envs, T, team = 8192, 8, 2
    g = _lib.default_gym_config(); g.obs_max_players = team; g.zero_sum = 1; g.team_spirit = 0.3; g.opp_scale = 1.0
    B = n_envs * 2 * team * T
    cfg = LearnerConfig(numEnvs=n_envs, teamSize=team, timestepsPerIteration=B, expBufferSize=B, randomSeed=3,
                        ppo=PPOLearnerConfig(batchSize=B, miniBatchSize=B // 4, epochs=1, autocastLearn=True))
    L = Learner(cfg, gym_cfg=g)
    assert L.obs_size == 51 + 38 * team and L.n_agents == n_envs * 4
    p0 = L.ppo.get_params(2).copy()
    for _ in range(3):
        L.iteration()
    torch.cuda.synchronize()
    rew = L.rew_buf.view(T, n_envs, 4); done = L.done_buf.view(T, n_envs, 4); obs = L.obs_buf
    assert torch.isfinite(rew).all() and torch.isfinite(obs).all()
    # ZeroSumReward with opponent scale 1: r_i (1 - s) + s mean(team) - mean(opponents) sums to zero over the four players
    assert rew.sum(dim=2).abs().max().item() < 1e-3 * max(1.0, rew.abs().max().item())
    assert (done == done[:, :, :1]).all()
    assert obs[..., :9].abs().max().item() < 3.5 and obs[..., 17:51].min().item() >= 0 and obs[..., 17:51].max().item() <= 1   # ball (scaled), pads are 0 / 1
    p1 = L.ppo.get_params(2)
    assert np.isfinite(p1).all() and np.abs(p1 - p0).max() > 0 and L.cumulative_model_updates == 3
    assert L.total_timesteps == 3 * B
    # the narrowphase's fixed-size queues: overflows (-> inline fallback for that env and tick) are visible in the release build, and rare
    ovf = L.env.overflow_counts()
    assert len(ovf) == 5 and sum(ovf) <= 1e-4 * (3 * T * n_envs * 8), ovf


def test_config4_shape_3v3_16384_envs_with_collection_during_learn():
    """BASELINE config[4]'s shape: 16 384 envs 3v3 (98 304 agent rows per step) with collectionDuringLearn -- the PPO epochs on their own
    stream under the next collection.  (The reference's autocast dtype is bf16, FrameworkTorch.h:14; so is this build's.)"""
    from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
    n_envs, T, team = 16384, 4, 3
    B = n_envs * 2 * team * T
    cfg = LearnerConfig(numEnvs=n_envs, teamSize=team, timestepsPerIteration=B, expBufferSize=B, randomSeed=4, collectionDuringLearn=True,
                        ppo=PPOLearnerConfig(batchSize=B, miniBatchSize=B // 4, epochs=1, autocastLearn=True))
    L = Learner(cfg)
    assert L.s_learn is not None and L.obs_size == 165 and L.n_agents == 98304
    p0 = L.ppo.get_params(2).copy()
    for it in range(3):
        L.iteration()
        assert L.cumulative_model_updates == it + 1
    rep = L.finish_report()
    torch.cuda.synchronize()
    p1 = L.ppo.get_params(2)
    assert np.isfinite(p1).all() and np.abs(p1 - p0).max() > 0
    assert torch.isfinite(L.rew_buf).all() and torch.isfinite(L.obs_buf).all()
    assert (L.done_buf.view(T, n_envs, 6) == L.done_buf.view(T, n_envs, 6)[:, :, :1]).all()
    assert 0 < rep["Policy Entropy"] < np.log(90) + 1e-3


def test_config4_as_worded_3v3_16384_envs_overlap_and_fp16_under_redzones(monkeypatch):
    """BASELINE configs[4] AS WORDED, all of it at once: 3v3, 16 384 envs per GPU, collect-during-learn overlap AND fp16 operands
    with the dynamic loss scale, on padded observations + ZeroSumReward, with guard bytes behind every device buffer of the env batch and of the
    learner (RLGPU_REDZONE).  Three iterations; size-independent properties: parameters finite and moved, the optimizer stepped three times, the
    loss scale still the scaler's initial 2^16 with three clean steps counted and none skipped, rewards zero-sum per env, done shared by an env's
    six players, rows finite, pads 0 / 1, no buffer overrun."""
    monkeypatch.setenv("RLGPU_REDZONE", "65536")
    from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
    from rlgymppo_cpp_amd import _lib
    n_envs, T, team = 16384, 4, 3
    g = _lib.default_gym_config(); g.obs_max_players = team; g.zero_sum = 1; g.team_spirit = 0.3; g.opp_scale = 1.0
    B = n_envs * 2 * team * T
    cfg = LearnerConfig(numEnvs=n_envs, teamSize=team, timestepsPerIteration=B, expBufferSize=B, randomSeed=5, collectionDuringLearn=True,
                        ppo=PPOLearnerConfig(batchSize=B, miniBatchSize=B // 4, epochs=1, autocastLearn="fp16"))
    L = Learner(cfg, gym_cfg=g)
    assert L.s_learn is not None and L.obs_size == 51 + 38 * team and L.n_agents == n_envs * 6
    p0 = L.ppo.get_params(2).copy()
    for it in range(3):
        L.iteration()
        assert L.cumulative_model_updates == it + 1
    rep = L.finish_report()
    torch.cuda.synchronize()
    p1 = L.ppo.get_params(2)
    assert np.isfinite(p1).all() and np.abs(p1 - p0).max() > 0
    scale, clean, skipped = L.ppo.loss_scale()
    assert scale == 65536.0 and clean == 3 and skipped == 0, (scale, clean, skipped)
    rew = L.rew_buf.view(T, n_envs, 6); done = L.done_buf.view(T, n_envs, 6); obs = L.obs_buf
    assert torch.isfinite(rew).all() and torch.isfinite(obs).all()
    assert rew.sum(dim=2).abs().max().item() < 1e-3 * max(1.0, rew.abs().max().item())
    assert (done == done[:, :, :1]).all()
    assert obs[..., 17:51].min().item() >= 0 and obs[..., 17:51].max().item() <= 1
    assert 0 < rep["Policy Entropy"] < np.log(90) + 1e-3
    L.env.check_redzones(); L.ppo.check_redzones()


# ---- multi-GPU pieces that one GPU can exercise ----------------------------------------------
def test_fake_ranks_on_one_gpu_equal_single_learner_on_the_union():
    """N = 4 device learners stand in for 4 ranks: identical parameters (same init seed), each takes ITS shard of a batch as a full local
    batch, the flat gradients are summed (what rlgpu_allreduce_grads does between GPUs), every "rank" applies rlgpu_clip_adam_step with
    grad_scale = 1 / N -- and ends up with the parameters of ONE learner that took the union as four minibatches.
    HIP kernels throughout; only the transport is replaced by a tensor sum."""
    from rlgymppo_cpp_amd.ppo import PPOCore
    dev = torch.device("cuda", 0)
    N, rows, D, A = 4, 512, 89, 90
    g = torch.Generator(device="cpu").manual_seed(0)
    obs = torch.randn(N * rows, D, generator=g).to(dev); acts = torch.randint(0, A, (N * rows,), generator=g, dtype=torch.int32).to(dev)
    adv = torch.randn(N * rows, generator=g).to(dev); tgt = torch.randn(N * rows, generator=g).to(dev)
    mk = lambda: PPOCore(D, A, (256, 256, 256), (256, 256, 256), use_bf16=False, max_rows=rows, seed=77)
    single = mk()
    logp = torch.empty(N * rows, device=dev)
    for r in range(N):   # old log-probs from the shared initial policy
        a_tmp = torch.empty(rows, dtype=torch.int32, device=dev)
        single.act(obs[r * rows:(r + 1) * rows], a_tmp, logp[r * rows:(r + 1) * rows])
    logp = logp + 0.05 * torch.randn(N * rows, generator=g).to(dev)
    ranks = [mk() for _ in range(N)]
    assert all(np.array_equal(single.get_params(2), c.get_params(2)) for c in ranks)
    met = torch.zeros(8, device=dev)
    single.zero_grads()
    for r in range(N):
        sl = slice(r * rows, (r + 1) * rows)
        single.minibatch(obs[sl], acts[sl], logp[sl], adv[sl], tgt[sl], None, rows, 1.0 / N, met)
        ranks[r].zero_grads(); ranks[r].minibatch(obs[sl], acts[sl], logp[sl], adv[sl], tgt[sl], None, rows, 1.0, met)
    single.clip_adam_step(0.5, 1.0)
    total = torch.stack([c.grad_tensor() for c in ranks]).sum(0)
    for c in ranks:
        c.grad_tensor().copy_(total); c.clip_adam_step(0.5, 1.0 / N)
    torch.cuda.synchronize()
    want = single.get_params(2)
    for c in ranks:
        assert np.abs(c.get_params(2) - want).max() < 2e-6


def test_rccl_communicator_through_the_cabi_and_rank_keyed_sampler():
    """rlgpu_comm_* with world size 1 (what one GPU can run): unique id, init, all-reduce of the learner's gradient buffer on its stream
    (identity), broadcast, destroy.  And the action sampler's key: two learners with the same init seed but different sampler
    streams (= ranks) draw different actions from identical observations, the same stream reproduces them."""
    import ctypes as C
    from rlgymppo_cpp_amd import _lib
    from rlgymppo_cpp_amd.ppo import PPOCore
    lib = _lib.load(); dev = torch.device("cuda", 0)
    uid = (C.c_ubyte * 128)()
    assert lib.rlgpu_comm_unique_id(uid) == 0
    h = C.c_void_p()
    assert lib.rlgpu_comm_init(C.byref(h), 0, 0, 1, uid) == 0, lib.rlgpu_comm_last_error(None)
    assert lib.rlgpu_comm_rank(h) == 0 and lib.rlgpu_comm_world(h) == 1
    core = PPOCore(89, 90, (256, 256, 256), (256, 256, 256), use_bf16=True, max_rows=256, seed=5)
    gt = core.grad_tensor(); gt.copy_(torch.arange(gt.numel(), device=dev, dtype=torch.float32) % 97)
    before = gt.clone()
    assert lib.rlgpu_allreduce_grads(core.h, h) == 0
    t = torch.arange(16, dtype=torch.float32, device=dev)
    assert lib.rlgpu_comm_broadcast(h, C.c_void_p(t.data_ptr()), 64, 0, None) == 0
    core.sync(); torch.cuda.synchronize()
    assert torch.equal(gt, before) and torch.equal(t, torch.arange(16, dtype=torch.float32, device=dev))
    assert lib.rlgpu_comm_destroy(h) == 0
    # the Python host's wrapper over the same calls, from the launcher's environment (a world of one here): scalar gathers default to the
    # current device (they used to build a CPU tensor), barrier, gradient all-reduce scale
    from rlgymppo_cpp_amd import parallel
    rc = parallel.RcclComm()
    assert (rc.rank, rc.world) == (0, 1) and rc.max_over_ranks(3.5) == 3.5 and rc.sum_over_ranks(2.25) == 2.25
    rc.barrier()
    assert rc.allreduce_gradients(core) == 1.0 and torch.equal(rc.share_from_rank0(t), t)
    rc.close()
    with pytest.raises(ValueError):
        from rlgymppo_cpp_amd.learner import Learner, LearnerConfig
        Learner(LearnerConfig(numEnvs=4), world_size=2)       # several ranks without an exchange object: refused, not silently diverging
    obs = torch.randn(256, 89, device=dev)
    draws = []
    for stream in (0, 1, 0):
        c = PPOCore(89, 90, (256, 256, 256), (256, 256, 256), use_bf16=True, max_rows=256, seed=5)
        c.set_sampler(stream, 0)
        a = torch.empty(256, dtype=torch.int32, device=dev); lp = torch.empty(256, device=dev)
        c.act(obs, a, lp); c.sync()
        draws.append(a.cpu().numpy().copy())
        assert c.get_sampler() == (stream, 1)
    assert np.array_equal(draws[0], draws[2]) and (draws[0] != draws[1]).mean() > 0.5


# ---- the host-plugin boundary of the C-ABI: snapshots, masked reset, host-parsed controls, device step statistics ----
def test_snapshots_masked_reset_controls_step_and_step_stats():
    """rlgpu_env_enable_snapshots / download_snapshots: the snapshot is the arena one tick into the step, and the rewards / dones the kernel
    wrote follow from it; rlgpu_env_step_controls with the table rows of the chosen actions == rlgpu_env_step; rlgpu_env_reset_envs resets
    exactly the listed envs; rlgpu_env_step_stats == the sums over the snapshots."""
    from rlgymppo_cpp_amd import _lib
    from rlgymppo_cpp_amd.env import BatchedEnv, action_table
    n_envs, team, K = 96, 2, 40
    g = _lib.default_gym_config(); g.no_touch_max_steps = 15
    envs = [BatchedEnv(n_envs, team, cfg=g) for _ in range(2)]
    a, b = envs
    a.enable_snapshots(); a.enable_step_stats()
    dev = torch.device("cuda", 0)
    N, D = a.n_agents, a.obs_size
    table = torch.from_numpy(action_table()).to(dev)
    obs = [e.reset(True) for e in envs]
    assert torch.equal(obs[0], obs[1])
    rng = np.random.RandomState(9)
    o = [torch.empty((N, D), device=dev) for _ in envs]; r = [torch.empty(N, device=dev) for _ in envs]; d = [torch.empty(N, dtype=torch.int32, device=dev) for _ in envs]
    tot = np.zeros(4); tick0 = np.array([s.tick_count for s in a.download_states()])
    ended = 0
    for k in range(K):
        acts = torch.from_numpy(rng.randint(0, 90, N).astype(np.int32)).to(dev)
        a.step(acts, o[0], r[0], d[0])
        b.step_controls(table[acts.long()].contiguous(), o[1], r[1], d[1])
        torch.cuda.synchronize()
        assert torch.equal(o[0], o[1]) and torch.equal(r[0], r[1]) and torch.equal(d[0], d[1]), k
        snaps = a.download_snapshots()
        after = a.download_states()
        dn = d[0].view(n_envs, 2 * team)[:, 0].cpu().numpy()
        for e in range(n_envs):
            s = snaps[e]
            assert s.tick_count == tick0[e] + 1                                 # one tick into the step
            for c in range(2 * team):
                car = s.cars[c]
                tot += [1, float(np.linalg.norm(car.vel[:])), float(bool(car.flags & (1 << 14)) and car.bh_tick_hit >= s.tick_count - g.tick_skip),
                        float(not (car.flags & 1))]
            if not dn[e]:
                assert after[e].tick_count == s.tick_count + g.tick_skip - 1
        ended += int(dn.sum())
        tick0 = np.array([s.tick_count for s in after])
    assert ended > 0
    got = a.step_stats(reset=True)
    assert got[0] == tot[0] == K * N
    assert abs(got[1] - tot[1]) <= 1e-4 * tot[1] and abs(got[3] - tot[3]) < 0.5
    assert abs(got[2] - tot[2]) <= max(2, ended)   # the touch window of an episode's first step is 1 tick, not tick_skip (PlayerData.cpp:20-25)
    assert a.step_stats()[0] == 0
    # masked reset: only the listed envs change
    before = a.download_states()
    ids = [3, 17, 64]
    rows = torch.full((N, D), -7.0, device=dev)
    a.reset_envs(ids, run_setter=True, obs=rows)
    torch.cuda.synchronize()
    now = a.download_states()
    for e in range(n_envs):
        same = bytes(before[e]) == bytes(now[e])
        assert same == (e not in ids), e
        block = rows.view(n_envs, 2 * team, D)[e]
        assert bool((block == -7.0).all()) == (e not in ids)
    for e in ids:
        assert now[e].gym.episode_steps == 0 and now[e].gym.score_line[0] == 0 and now[e].gym.reset_count == before[e].gym.reset_count + 1


def _torch_ppo_reference(pol_flat, cri_flat, D, H, A, obs, acts, old_logp, adv, tgt, clip, ent_coef, scale, dtype, dev):
    """PPOLearner::Learn's minibatch body (PPOLearner.cpp:139-215; DiscretePolicy::GetBackpropData, DiscretePolicy.cpp:64-75; policy forward
    DiscretePolicy.h:27-31) as a torch autograd graph in `dtype` on `dev`, with this repo's flat parameter layout loaded into nn.Sequential's
    state-dict order.  Returns (policy grads, critic grads, entropy, kl, clip fraction, value loss)."""
    def mlp(flat, sizes):
        layers, off = [], 0
        for i in range(len(sizes) - 1):
            lin = torch.nn.Linear(sizes[i], sizes[i + 1]).to(dev, dtype)
            n = sizes[i] * sizes[i + 1]
            with torch.no_grad():
                lin.weight.copy_(torch.from_numpy(flat[off:off + n].reshape(sizes[i + 1], sizes[i])).to(dev, dtype)); off += n
                lin.bias.copy_(torch.from_numpy(flat[off:off + sizes[i + 1]]).to(dev, dtype)); off += sizes[i + 1]
            layers.append(lin)
            if i < len(sizes) - 2:
                layers.append(torch.nn.ReLU())
        assert off == len(flat)
        return torch.nn.Sequential(*layers)
    pol, cri = mlp(pol_flat, [D] + list(H) + [A]), mlp(cri_flat, [D] + list(H) + [1])
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    o = t(obs).to(dtype); a = t(acts).long(); olp = t(old_logp).to(dtype); ad = t(adv).to(dtype); tg = t(tgt).to(dtype)
    vals = cri(o).view(-1)
    probs = torch.clamp(torch.softmax(pol(o) / 1.0, dim=-1), min=1e-11, max=1)
    log_probs = torch.log(probs)
    alp = log_probs.gather(-1, a[:, None]).view(-1)
    entropy = -(log_probs * probs).sum(dim=-1).mean()
    ratio = torch.exp(alp - olp)
    clipped = torch.clamp(ratio, 1 - clip, 1 + clip)
    policy_loss = -torch.min(ratio * ad, clipped * ad).mean()
    ((policy_loss - entropy * ent_coef) * scale).backward()
    value_loss = torch.nn.functional.mse_loss(vals, tg)
    (value_loss * scale).backward()
    flat = lambda seq: np.concatenate([p.grad.detach().double().cpu().numpy().reshape(-1) for p in seq.parameters()])
    with torch.no_grad():
        lr = alp - olp
        kl = ((torch.exp(lr) - 1) - lr).mean().item(); cf = (torch.abs(ratio - 1) > clip).double().mean().item()
    return flat(pol), flat(cri), entropy.item(), kl, cf, value_loss.item()


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [8192 + 37, 65536])
def test_ppo_minibatch_at_the_flagship_shape_against_torch_autograd(rows, monkeypatch):
    """SURVEY A18 at the shape the bench runs: obs 89 -> 256 x 3 -> 90 / 1, a minibatch of 8 229 rows (ragged against every tile: 128-row GEMM
    tiles, 512-row dW slabs) and one of 65 536 (the bench's: 128 slabs per dW GEMM summed with fp32 atomics, four streams joined), gathered
    through a shuffled index list out of a larger buffer.  Reference: PPOLearner.cpp:139-215 as a torch autograd graph built IN the test, in
    float64, with the same graph in torch fp32 next to it: the fp32 path's gradients are within 5e-6 of the largest entry per network (8 229
    rows) or, where an fp32 sum over 65 536 largely cancelling rows cannot be, within 3 x the error torch's own fp32 pass makes; entropy / KL /
    clip fraction / value loss 1e-5; the bf16 path (bf16 operands, fp32 sums) against the same: cosine > 0.999 per
    network and 3 % of the largest entry."""
    from rlgymppo_cpp_amd.ppo import PPOCore
    dev = torch.device("cuda", 0)
    D, A, H = 89, 90, (256, 256, 256)
    rng = np.random.RandomState(rows % 1000)
    pool = rows + 4099
    obs = (rng.randn(pool, D) * 0.7).astype(np.float32)
    acts = rng.randint(0, A, size=pool).astype(np.int32)
    adv = rng.randn(pool).astype(np.float32); tgt = rng.randn(pool).astype(np.float32)
    idx = rng.permutation(pool)[:rows].astype(np.int32)
    clip, ent_coef, scale = 0.2, 0.01, 0.25
    want = None
    monkeypatch.setenv("RLGPU_REDZONE", "65536")   # guard bytes behind every buffer of the learner: checked after the minibatch (fp32 per-layer path, bf16 fused path)
    for bf16 in (False, True):
        core = PPOCore(D, A, H, H, ent_coef=ent_coef, clip_range=clip, use_bf16=bf16, seed=99, max_rows=rows)
        pol_flat, cri_flat = core.get_params(0), core.get_params(1)
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        d_obs, d_acts, d_adv, d_tgt, d_idx = t(obs), t(acts), t(adv), t(tgt), t(idx)
        if want is None:
            # old log-probs near the policy's own (ratios around 1, some outside the clip range), from the fp32 policy's probabilities
            with torch.no_grad():
                p = core.probs(d_obs[:core.max_rows]) if pool <= core.max_rows else torch.cat([core.probs(d_obs[s:s + core.max_rows].contiguous()) for s in range(0, pool, core.max_rows)])
            lp = torch.log(p).gather(-1, d_acts.long()[:, None]).view(-1).cpu().numpy()
            old_logp = (lp + rng.randn(pool).astype(np.float32) * 0.15).astype(np.float32)
            want = _torch_ppo_reference(pol_flat, cri_flat, D, H, A, obs[idx], acts[idx], old_logp[idx], adv[idx], tgt[idx], clip, ent_coef, scale, torch.float64, dev)
            # ... and what plain PyTorch fp32 (the reference's own arithmetic: libtorch fp32 GEMMs + autograd) makes of the same minibatch: a sum
            # over 65 536 rows of terms that largely cancel is only so accurate in fp32, whoever adds them up
            t32 = _torch_ppo_reference(pol_flat, cri_flat, D, H, A, obs[idx], acts[idx], old_logp[idx], adv[idx], tgt[idx], clip, ent_coef, scale, torch.float32, dev)
            torch_err = [np.abs(t32[k] - want[k]).max() for k in (0, 1)]
        d_olp = t(old_logp)
        metrics = torch.zeros(8, device=dev)
        core.zero_grads()
        core.minibatch(d_obs, d_acts, d_olp, d_adv, d_tgt, d_idx, rows, scale, metrics)
        core.sync()
        core.check_redzones()
        gp, gc = core.get_grads(0).astype(np.float64), core.get_grads(1).astype(np.float64)
        m = metrics.cpu().numpy().astype(np.float64)
        for g, w, name, terr in ((gp, want[0], "policy", torch_err[0]), (gc, want[1], "critic", torch_err[1])):
            assert np.isfinite(g).all()
            big = np.abs(w).max()
            if not bf16:
                # 5e-6 of the largest entry, or -- where fp32 summation itself cannot do that -- within 3 x the error of torch's own fp32 pass
                assert np.abs(g - w).max() <= max(5e-6 * big, 3 * terr), f"fp32 {name} gradient: {np.abs(g - w).max()} vs largest entry {big} (torch fp32: {terr})"
            else:
                cos = float(g @ w / (np.linalg.norm(g) * np.linalg.norm(w)))
                assert cos > 0.999 and np.abs(g - w).max() <= 0.03 * big, f"bf16 {name} gradient: cosine {cos}, max diff {np.abs(g - w).max()} vs {big}"
        tol = 1e-5 if not bf16 else 5e-3
        assert abs(m[0] / rows - want[2]) < tol * max(1, abs(want[2])) and abs(m[1] / rows - want[3]) < tol and abs(m[2] / rows - want[4]) < (1e-9 if not bf16 else 2e-3)
        assert abs(m[4] / rows - want[5]) < tol * max(1, abs(want[5])) * (1 if not bf16 else 4)
        core.close()


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_no_contact_is_ever_dropped_device_fallback_against_the_reference_fixtures(tmp_path):
    """VERDICT r04 item 2.  A tick whose contacts do not fit the env's LDS-resident layout is redone with the big layout in global memory
    (rlgpu_env.hip:tick_world_big -> arena_step.h:world_step_finish_big) -- detected before any contact callback has fired.  The shipped layout
    overflows about twice in 400 M env-ticks, so the fallback is exercised by a TEST BUILD whose layout is cut down to one mesh manifold, one
    plane slot, one car-pair point and three solver contacts (make -C csrc tiny): the reference-fixture part of this suite -- the 31 free-run
    tapes for bit-equality, the two-file and sixteen-file meshes, the gym rollouts, fused collection against alternating act / step -- runs
    against it in a child process and passes as it stands; the child's counters say that env-ticks DID take the fallback and that nothing was lost."""
    import json
    import subprocess
    csrc = os.path.join(ROOT, "rlgymppo_cpp_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "tiny"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    so = os.path.join(ROOT, "rlgymppo_cpp_amd", "librlgpu_tiny.so")
    counts = str(tmp_path / "counts.json")
    env = dict(os.environ); env["RLGPU_LIB"] = so; env["RLGPU_COUNTS_OUT"] = counts
    keep = ("test_hip_free_run_is_bit_identical_to_the_reference or test_physics_ticks_match_host_port_on_golden_scenarios or test_hip_mesh_of_two_files "
            "or test_hip_tessellated_mesh_of_sixteen_files or test_hip_gym_rollouts_vs_reference_fixtures or test_gym_step_matches_host_port_team_modes "
            "or (test_fused_collection_equals_alternating_act_and_step and not 3-9) or test_hip_gameinst_episode_boundaries or test_hip_wedge_fixture")
    # (3v3 fused collection: the cut-down TickWork is too small for the inference buffers that borrow its bytes -- rlgpu_env_collect refuses, as it should)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "-k", keep, "-p", "no:cacheprovider"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:]
    c = json.load(open(counts))
    print("cut-down layout:", c, "|", r.stdout.strip().splitlines()[-1])
    assert c["big_layout_ticks"] > 100, c
    assert c["lost_contacts"] == 0, c
    # ... and the shipped layout, in this process so far: nothing lost either
    from rlgymppo_cpp_amd.env import BatchedEnv
    assert BatchedEnv(4, 1).lost_contact_count() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("team_size,n_envs,bf16", [(1, 70, True), (2, 41, True), (3, 19, True), (2, 20, False), (2, 5000, True)])
def test_step_queue_collection_equals_one_workgroup_per_group(team_size, n_envs, bf16):
    """Lockstep collection of a batch with more wavefront-groups than the device keeps resident runs through a queue of (step, group) tickets taken by
    resident wavefronts (rlgpu_env.hip:k_env_collect_q; BASELINE configs[3] / [4]).  The same flow -- reset, three launches of T gym steps with the
    sampler rewound -- through the queue (forced, also for batches that would fit) and through the one-workgroup-per-group kernel: observations,
    actions, log-probs, rewards, terminals and the downloaded states equal, bit for bit; 2v2 / 5 000 envs is a batch that takes the queue by itself."""
    from rlgymppo_cpp_amd import _lib
    from rlgymppo_cpp_amd.env import BatchedEnv
    from rlgymppo_cpp_amd.ppo import PPOCore
    dev = torch.device("cuda", 0); T = 6
    outs = []
    for mode in (1, 0):
        cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 9
        env = BatchedEnv(n_envs, team_size, cfg); env.set_collect_queue(mode)
        core = PPOCore(env.obs_size, env.n_actions, (64, 64), (64, 64), use_bf16=bf16, max_rows=max(4096, env.n_agents), seed=3)
        N, D = env.n_agents, env.obs_size
        obs = torch.zeros((T + 1, N, D), device=dev); act = torch.zeros((T, N), dtype=torch.int32, device=dev); logp = torch.zeros((T, N), device=dev)
        rew = torch.zeros((T, N), device=dev); done = torch.zeros((T, N), dtype=torch.int32, device=dev)
        env.reset(True, obs[0])
        keep = []
        for it in range(3):
            assert env.collect(core, T, obs, act, logp, rew, done); env.sync()
            keep.append([x.cpu().numpy().copy() for x in (obs, act, logp, rew, done)])
            obs[0].copy_(obs[T])
        states = env.download_states()
        outs.append((keep, [bytes(s) for s in states]))
        env.close()
    for a, b in zip(outs[0][0], outs[1][0]):
        for x, y, name in zip(a, b, ("obs", "act", "logp", "rew", "done")):
            assert np.array_equal(x, y), name
    assert outs[0][1] == outs[1][1]
    assert sum(int(k[4].sum()) for k in outs[0][0]) > 0      # episodes ended and were reset inside the launches


@pytest.mark.gpu
def test_stripe_kernels_equal_the_per_layer_path(tmp_path):
    """csrc/mlp_stripe.h (RLGPU_STRIPE=1): forward and dX chains of both networks in one launch each, activations in LDS from layer to layer.
    Same operands, same accumulation order per 32x32 tile, same rounding points as the per-layer GEMMs: gradients and metrics of a ragged
    flagship-shape minibatch (8 229 rows) are compared with the default path's for EQUALITY up to the fp32 atomics' summation order (1e-6 of the
    largest entry)."""
    import subprocess
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
from rlgymppo_cpp_amd.ppo import PPOCore
dev = torch.device('cuda', 0); rows = 8229; D, A = 89, 90
rng = np.random.RandomState(5)
core = PPOCore(D, A, (256, 256, 256), (256, 256, 256), use_bf16=True, seed=3, max_rows=rows)
t = lambda x: torch.from_numpy(x).to(dev)
obs = t((rng.randn(rows + 100, D) * 0.7).astype(np.float32)); acts = t(rng.randint(0, A, rows + 100).astype(np.int32))
olp = t((-4.5 + rng.randn(rows + 100) * 0.2).astype(np.float32)); adv = t(rng.randn(rows + 100).astype(np.float32)); tgt = t(rng.randn(rows + 100).astype(np.float32))
idx = t(rng.permutation(rows + 100)[:rows].astype(np.int32)); m = torch.zeros(8, device=dev)
core.zero_grads(); core.minibatch(obs, acts, olp, adv, tgt, idx, rows, 0.25, m); core.sync()
np.savez(sys.argv[1], gp=core.get_grads(0), gc=core.get_grads(1), m=m.cpu().numpy())
""" % ROOT
    outs = []
    for stripe in (False, True):
        out = str(tmp_path / ("s%d.npz" % stripe))
        env = dict(os.environ); env.pop("RLGPU_STRIPE", None)
        env["RLGPU_NO_FUSED"] = "1"   # (the default for this shape is csrc/ppo_fused.h, which has a test of its own below)
        if stripe: env["RLGPU_STRIPE"] = "1"
        r = subprocess.run([sys.executable, "-c", code, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        assert r.returncode == 0, r.stdout[-3000:]
        outs.append(np.load(out))
    for k in ("gp", "gc"):
        big = np.abs(outs[0][k]).max()
        assert np.abs(outs[0][k] - outs[1][k]).max() <= 1e-6 * big, k
    assert np.allclose(outs[0]["m"], outs[1]["m"], rtol=1e-5, atol=0)   # (sums of per-workgroup partials added with atomics: the order varies)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,obs_size", [(8229, 89), (1000, 127), (777, 165)])
def test_fused_minibatch_kernels_against_the_per_layer_path(rows, obs_size):
    """csrc/ppo_fused.h, the default bf16 path at the flagship shape (obs -> 256 x 3 -> 90 / 1): gather + forward + loss + dX chain of both
    networks in ONE launch per 128-row stripe, every dW / db in a second one.  Compared with the per-layer kernels (RLGPU_NO_FUSED=1: k_gemm_nt /
    k_gemm_tn / k_ppo_policy_loss / k_value_loss, themselves pinned to torch autograd by
    test_ppo_minibatch_at_the_flagship_shape_against_torch_autograd, which now runs the fused path too) on a ragged minibatch gathered through a
    shuffled index list, for the three observation widths the kernel is instantiated for (1v1: 89 -> 96 padded, 2v2 padded obs: 127 -> 128, 3v3:
    165 -> 192).  Same operands and rounding points; what differs is the summation order (the bias is the MFMA's first addend, dW slabs of 2048
    rows) and v_exp_f32 / v_log_f32 in the loss: gradients within 1e-2 of the largest entry with cosine > 0.99999, metrics within 2e-4."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fused_check.py"), str(rows), str(obs_size)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-3000:]


@pytest.mark.gpu
def test_hip_rotated_ball_basis_vs_reference_golden():
    """BallState::rotMat on the device (VERDICT r03 missing #2): the four tapes of tests/golden/ballrot_golden.npz (recorded from the reference
    with a ball basis that is not the identity; the host build equals them bit for bit: test_port_rotated_ball_basis_vs_reference_golden) as
    four envs of one batch.  The basis travels in RlgpuArenaState::hidden.ball_rot, is nine of the resident words, is what the plane contact's
    support vertex and the wheel rays' convex cast are computed in, and comes back unchanged from every download, as the reference's GetState
    does under ArenaConfig::noBallRot: every tick of every tape EQUAL to the reference."""
    from rlgymppo_cpp_amd.env import BatchedEnv
    from simlib import state_vec
    g = np.load(os.path.join(GOLD, "ballrot_golden.npz"))
    names = [str(x) for x in g["names"]]
    starts = [ArenaState.from_buffer_copy(g[f"{n}/start_raw"].tobytes()) for n in names]
    env = BatchedEnv(len(names), 1, mesh=(g["mesh_verts"], g["mesh_tris"]))
    env.upload_states(starts)
    tapes = [g[f"{n}/tape"] for n in names]
    T = max(len(t) for t in tapes)
    ctl = np.zeros((len(names), 2, 8), np.float32)
    first_diff = {}
    for t in range(T):
        for i, tp in enumerate(tapes):
            if t < len(tp): ctl[i] = tp[t]
        env.set_controls(ctl); env.physics_ticks(1)
        cur = env.download_states()
        for i, n in enumerate(names):
            if t < len(tapes[i]):
                if not np.array_equal(state_vec(cur[i]), g[f"{n}/states"][t]): first_diff.setdefault(n, t + 1)
                assert list(cur[i].hidden.ball_rot) == list(starts[i].hidden.ball_rot), f"{n} tick {t + 1}: the reported ball basis changed"
    # an explicit reset by a built-in state setter is a SetState with a default BallState: the identity again
    env.reset(True); env.sync()
    assert all(list(s.hidden.ball_rot) == [1, 0, 0, 0, 1, 0, 0, 0, 1] for s in env.download_states())
    env.close()
    assert not first_diff, f"first differing tick per tape: {first_diff}"


@pytest.mark.gpu
@pytest.mark.parametrize("rows,obs_size", [(70001, 89), (1000, 127), (513, 165)])
def test_value_stripe_kernel_against_the_inference_kernel(rows, obs_size):
    """The value pass of Learner::AddNewExperience (Learner.cpp:296-316) at the flagship shape: k_value_stripe (csrc/ppo_fused.h: the critic's
    forward chain per 128-row stripe, activations in LDS, nothing written but the values) against k_mlp_infer (RLGPU_NO_VALUE_STRIPE=1; pinned to
    the reference's ValueEstimator by tests/test_ref_learner.py in fp32 mode) on a ragged row count, for the three observation widths.  Same
    bf16 operands and rounding points; the bias is the first addend instead of the last: within 2e-3 of the largest value."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "value_stripe_check.py"), str(rows), str(obs_size)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-3000:]


@pytest.mark.gpu
def test_fp16_operand_mode_with_dynamic_loss_scale(monkeypatch):
    """BASELINE configs[4] words its precision "fp16 autocast" (VERDICT r03 row G2).  use_bf16 = 2: the PPO minibatch kernels (csrc/ppo_fused.h)
    take fp16 operands -- weights, activations, activation gradients -- with fp32 sums and master weights, and the loss gradient is multiplied by
    amp::GradScaler's dynamic scale (PRIV/Util/gradscaler.hpp:26-34: 2^16 at the start, x 2 after 2000 clean steps, x 0.5 after an overflow,
    whose optimizer step is skipped: :162,291).  The one deviation from PPOLearner.cpp:273-297, which clips the SCALED gradients: the
    gradients are unscaled BEFORE the clip, as torch documents it.
    Checked: (1) the gradients of a ragged flagship-shape minibatch, divided by the scale, against PPOLearner.cpp:139-215 as a torch autograd
    graph in float64 -- cosine > 0.999 per network and within 3 % of the largest entry, the bound the bf16 path is held to; (2) a clean
    optimizer step moves the parameters and counts towards the next doubling; (3) an overflowing minibatch (advantages of 1e30) leaves
    the parameters and Adam's step count alone, halves the scale and is counted as skipped."""
    from rlgymppo_cpp_amd.ppo import PPOCore
    monkeypatch.setenv("RLGPU_REDZONE", "65536")
    dev = torch.device("cuda", 0)
    D, A, H, rows = 89, 90, (256, 256, 256), 8229
    rng = np.random.RandomState(11)
    pool = rows + 500
    obs = (rng.randn(pool, D) * 0.7).astype(np.float32); acts = rng.randint(0, A, size=pool).astype(np.int32)
    adv = rng.randn(pool).astype(np.float32); tgt = rng.randn(pool).astype(np.float32)
    idx = rng.permutation(pool)[:rows].astype(np.int32)
    clip, ent_coef, scale = 0.2, 0.01, 0.25
    core = PPOCore(D, A, H, H, ent_coef=ent_coef, clip_range=clip, use_bf16="fp16", seed=99, max_rows=pool)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_obs, d_acts, d_adv, d_tgt, d_idx = t(obs), t(acts), t(adv), t(tgt), t(idx)
    with torch.no_grad():
        p = core.probs(d_obs)
    lp = torch.log(p).gather(-1, d_acts.long()[:, None]).view(-1).cpu().numpy()
    old_logp = (lp + rng.randn(pool).astype(np.float32) * 0.15).astype(np.float32)
    pol_flat, cri_flat = core.get_params(0), core.get_params(1)
    want = _torch_ppo_reference(pol_flat, cri_flat, D, H, A, obs[idx], acts[idx], old_logp[idx], adv[idx], tgt[idx], clip, ent_coef, scale, torch.float64, dev)
    s0, growth, skipped = core.loss_scale()
    assert s0 == 65536.0 and growth == 0 and skipped == 0
    metrics = torch.zeros(8, device=dev)
    core.zero_grads(); core.minibatch(d_obs, d_acts, t(old_logp), d_adv, d_tgt, d_idx, rows, scale, metrics); core.sync()
    for which, w, name in ((0, want[0], "policy"), (1, want[1], "critic")):
        g = core.get_grads(which).astype(np.float64) / s0
        assert np.isfinite(g).all()
        cos = float(g @ w / (np.linalg.norm(g) * np.linalg.norm(w))); big = np.abs(w).max()
        assert cos > 0.999 and np.abs(g - w).max() <= 0.03 * big, f"fp16 {name} gradient: cosine {cos}, max diff {np.abs(g - w).max()} vs {big}"
    core.clip_adam_step(0.5, 1.0); core.sync()
    p1 = core.get_params(2)
    assert np.isfinite(p1).all() and not np.array_equal(p1[:len(pol_flat)], pol_flat)
    assert core.loss_scale() == (65536.0, 1, 0)
    # an overflow: the scaled loss gradient leaves fp16's range -> non-finite gradient norm -> both optimizers skip, the scale backs off
    core.zero_grads(); core.minibatch(d_obs, d_acts, t(old_logp), t((adv * 1e30).astype(np.float32)), t((tgt * 1e30).astype(np.float32)), d_idx, rows, scale, metrics)
    core.clip_adam_step(0.5, 1.0); core.sync()
    assert np.array_equal(core.get_params(2), p1), "a step with a non-finite gradient norm must leave the parameters alone"
    assert core.loss_scale() == (32768.0, 0, 1)
    _, _, sp, sc = core.get_adam_state()
    assert sp == 1 and sc == 1
    core.check_redzones()   # no kernel of the fp16 mode wrote past a buffer of the learner (RLGPU_REDZONE, set above)
    core.close()


def test_hip_live_gym_rollouts_equal_the_reference():
    """Round 6 on the GPU: tests/golden/live_gym_golden.npz (tools/live_gym_hip.py --record; tests/test_oracle_golden.py::test_port_resident_gym_rollouts_equal_the_live_reference
    says what each rollout carries) replayed by the body of test_hip_gym_rollouts_vs_reference_fixtures with EXACT comparison: done, every reward and every
    observation row bit for bit (the other players' blocks in the reference's order; DefaultOBSPadded's shuffled lists as multisets; padded rollouts up to their first
    respawn -- simlib.live_gym_cases)."""
    from simlib import live_gym_cases
    rec = np.load(os.path.join(GOLD, "live_gym_golden.npz")); gold = np.load(os.path.join(GOLD, "sim_golden.npz"))
    n = 0
    for case, fx, horizon in live_gym_cases(rec, gold):
        test_hip_gym_rollouts_vs_reference_fixtures(fx); n += 1
    assert n >= 13
