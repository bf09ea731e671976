"""CPU tests: the oracle restatements against the committed golden vectors (generated from the real reference /
torch by tests/golden/make_*.py), and the C-ABI library's exported symbols."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import learner_ref as R
from rlgymppo_cpp_amd.state import ArenaState
from simlib import port_gym_cfg, port_gym_reset, port_gym_step

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def lg():
    return np.load(os.path.join(GOLD, "learner_golden.npz"))


@pytest.fixture(scope="module")
def sg():
    return np.load(os.path.join(GOLD, "sim_golden.npz"))


def shapes(g):
    D, A, H = int(g["D"]), int(g["A"]), int(g["H"])
    return [((H, D), (H,)), ((H, H), (H,)), ((A, H), (A,))], [((H, D), (H,)), ((H, H), (H,)), ((1, H), (1,))]


def test_policy_forward_probs_sampling(lg):
    ps, cs = shapes(lg)
    logits, _ = R.mlp_forward(lg["pol_params"], ps, lg["obs"])
    assert np.abs(logits - lg["logits"]).max() < 1e-5
    p = R.policy_probs(logits)
    assert np.abs(p - lg["probs"]).max() < 1e-6
    a, lp = R.sample_actions(lg["probs"], lg["q"])
    assert (a == lg["actions"]).all()            # bit-exact action indices on the recorded noise tape
    assert np.abs(lp - lg["logp"]).max() < 1e-5
    v, _ = R.mlp_forward(lg["cri_params"], cs, lg["obs"])
    assert np.abs(v[:, 0] - lg["values"]).max() < 1e-5


def test_gae(lg):
    adv, tg, rt = R.compute_gae(lg["gae_rews"], lg["gae_terminal"], lg["gae_truncated"], lg["gae_values"], float(lg["gae_gamma"]),
                                float(lg["gae_lambda"]), float(lg["gae_ret_std"]), float(lg["gae_clip"]))
    assert np.abs(adv - lg["gae_adv"]).max() <= 1e-4     # north_star tolerance on returns/advantages
    assert np.abs(rt - lg["gae_returns"]).max() <= 1e-4
    assert np.abs(tg - lg["gae_targets"]).max() <= 1e-4


def test_ppo_loss_grads(lg):
    ps, cs = shapes(lg)
    gp, gc, m = R.ppo_minibatch_grads(lg["pol_params"], ps, lg["cri_params"], cs, lg["obs"], lg["actions"], lg["old_logp"], lg["adv"], lg["targets"],
                                      float(lg["clip"]), float(lg["ent_coef"]), float(lg["scale"]))
    assert np.abs(gp - lg["pol_grads"]).max() < 1e-6
    assert np.abs(gc - lg["cri_grads"]).max() < 1e-6
    assert abs(m["entropy"] - float(lg["entropy"])) < 1e-5 and abs(m["kl"] - float(lg["kl"])) < 1e-6
    assert abs(m["clip_fraction"] - float(lg["clip_fraction"])) < 1e-6 and abs(m["value_loss"] - float(lg["value_loss"])) < 1e-5


def test_clip_adam(lg):
    p = lg["pol_params"]; m = np.zeros_like(p); v = np.zeros_like(p)
    for s, (gk, pk) in enumerate([("adam_g0", "adam_p1"), ("adam_g1", "adam_p2"), ("adam_g2", "adam_p3")]):
        p, m, v = R.clip_adam_step(p, lg[gk], m, v, s + 1, float(lg["adam_lr"]))
        assert np.abs(p - lg[pk]).max() < 1e-7


def test_libstdcxx_shuffle(lg):
    st = 123
    for rep in range(2):
        for i, n in enumerate([1, 2, 7, 64, 1000]):
            perm, st = R.libstdcxx_shuffle(n, st)
            assert (perm == lg[f"shuffle_{rep * 5 + i}"]).all()


def test_library_shuffler_matches_oracle_and_row_mapping(lg):
    """rlgpu_shuffler_next == libstdc++ std::shuffle with a persistent default_random_engine (the oracle restates it; the golden
    tape comes from the real thing), and rlgpu_shuffler_next_rows is that draw mapped from agent-major logical order to
    time-major rows.  Host-only entry points of librlgpu.so: no GPU needed."""
    from rlgymppo_cpp_amd.learner import Shuffler
    a, st = Shuffler(123), 123
    for rep in range(2):
        for i, n in enumerate([1, 2, 7, 64, 1000]):
            want, st = R.libstdcxx_shuffle(n, st)
            assert (a.next(n) == want).all() and (want == lg[f"shuffle_{rep * 5 + i}"]).all()
    b, c = Shuffler(77), Shuffler(77)
    T, N = 5, 12
    for _ in range(3):
        p = b.next(T * N)
        assert (c.next_rows(T, N) == ((p % T) * N + p // T)).all()


@pytest.mark.parametrize("max_rows", [60, 100, 150, 37, 400])
def test_experience_fifo_matches_reference_buffer(max_rows):
    """rlgpu_expbuf_* (slot bookkeeping, no data movement) against the oracle's literal ExperienceBuffer: the same rows stay alive in
    the same logical order, and the shuffled batches name the same rows.  Sizes cover: buffer smaller than one iteration (keeps the
    last rows), partial eviction of the oldest iteration, whole multiples, and a buffer that never fills."""
    from rlgymppo_cpp_amd.learner import ExperienceFifo, Shuffler
    T, N = 5, 12
    B = T * N
    fifo, shuf = ExperienceFifo(max_rows, T, N), Shuffler(99)
    ref = R.ExperienceBufferRef(max_rows, 99)
    slot_of = {}
    for it in range(7):
        slot = fifo.submit()
        assert 0 <= slot < fifo.num_slots
        slot_of = {k: v for k, v in slot_of.items() if v != slot}   # a reused slot's old iteration must be gone from the FIFO (checked below)
        slot_of[it] = slot
        # the reference's logical order inside one submit: trajectory after trajectory (agent-major); identity = it * B + a
        ref.submit(it * B + np.arange(B))
        assert fifo.size() == len(ref.data) == min(max_rows, (it + 1) * B)
        rows = np.empty(fifo.size(), np.int32)
        for batch_size in (16, 25):
            n = fifo.shuffled_rows(shuf, rows)
            want = ref.all_batches_shuffled(batch_size)
            for b, w in enumerate(want):
                k, a = w // B, w % B
                assert all(int(i) in slot_of for i in k)
                phys = np.array([slot_of[int(i)] for i in k]) * B + (a % T) * N + a // T   # device row: slot, time-major inside
                assert (rows[b * batch_size:(b + 1) * batch_size] == phys).all()
            assert len(want) == n // batch_size


def test_welford():
    w = R.Welford()
    xs = np.random.RandomState(0).randn(300) * 3 + 1
    w.increment(xs, 150)
    assert abs(w.std() - np.std(xs[:150], ddof=1)) < 1e-9
    assert R.Welford().std() == 1.0


def test_action_table(sg, port_lib):
    tab = np.zeros((90, 8), np.float32)
    assert port_lib.lib.port_action_table(tab.ctypes.data_as(C.c_void_p)) == 90
    assert sg["action_table"].shape == (90, 8)
    assert (tab == sg["action_table"]).all()


def _state_vec(s):
    v = list(s.ball.pos) + list(s.ball.vel) + list(s.ball.ang_vel)
    for k in range(s.num_cars):
        c = s.cars[k]
        v += list(c.pos) + list(c.vel) + list(c.ang_vel) + list(c.rot) + [float(c.flags), c.boost]
    return np.array(v, np.float32)


# per-scenario tolerances (uu / uu/s) on ball and car position over the whole golden trajectory, and the tick up to
# which they hold.  Free flight / wheels / ball contacts are tight; hitbox contacts are chaotic: those
# scenarios are compared tightly only up to the first hitbox contact and loosely after it.
PHYS_TOL = {
    "rest": (0.01, None), "throttle": (0.02, None), "steer_powerslide": (0.1, None), "jump": (0.05, None), "flip": (1.0, None),
    "double_jump": (0.05, None), "boost_turn": (0.2, None), "ball_drop": (0.1, None), "ball_roll": (0.1, None), "car_hits_ball": (0.3, None),
    "ball_side_wall": (0.1, None), "ball_back_wall_mesh": (0.1, None), "ball_corner_fillets": (0.3, None), "ball_into_goal": (0.3, None),
    "air_control": (0.1, None), "wall_ramp": (5.0, None), "car_car_head_on": (0.05, 110), "roof_landing_autoflip": (0.05, 40),
    "boost_pad_pickup": (0.05, None),
}


def test_port_physics_vs_reference_golden(sg, port_lib):
    every = int(sg["phys_every"])
    for name in sg["phys_names"]:
        name = str(name)
        st = ArenaState.from_buffer_copy(sg[f"phys/{name}/start"].tobytes())
        tape = sg[f"phys/{name}/tape"]; want = sg[f"phys/{name}/states"]
        tol, until = PHYS_TOL[name]
        for t in range(len(tape)):
            for k in range(2):
                st.cars[k].controls[:] = list(tape[t, k])
            port_lib.step(st, 1)
            if (t + 1) % every == 0 and (until is None or t < until):
                got = _state_vec(st); ref = want[(t + 1) // every - 1]
                # layout: ball 9 floats, then 20 per car: pos3 vel3 angvel3 rot9 flags boost
                perr = max(np.abs(got[0:3] - ref[0:3]).max(), np.abs(got[9:12] - ref[9:12]).max(), np.abs(got[29:32] - ref[29:32]).max())
                assert perr <= tol, f"{name}: position error {perr:.4f} uu at tick {t + 1} (tol {tol})"
                assert got[27] == ref[27] or name in ("wall_ramp", "flip"), f"{name}: car0 flags differ at tick {t + 1}: {int(got[27]):x} vs {int(ref[27]):x}"
                assert abs(got[28] - ref[28]) < 1e-3, f"{name}: boost differs at tick {t + 1}"


def test_port_gym_vs_reference_golden(sg, port_lib):
    for case in sg["gym_names"]:
        case = str(case)
        cfg = port_gym_cfg(tick_skip=int(sg[f"gym/{case}/tick_skip"]))
        st = ArenaState.from_buffer_copy(sg[f"gym/{case}/start"].tobytes())
        (st,), obs0 = port_gym_reset(port_lib, [st], cfg, run_setter=False)
        assert np.abs(obs0 - sg[f"gym/{case}/obs0"]).max() < 1e-6
        acts = sg[f"gym/{case}/actions"]; obs = sg[f"gym/{case}/obs"]; rew = sg[f"gym/{case}/rew"]; done = sg[f"gym/{case}/done"]
        # rollouts agree with the reference to ~1e-6 until the first hitbox contact (chaotic afterwards)
        horizon = {"ts8_random": 60, "ts8_chase": 40, "ts1_random": len(acts)}[case]
        for t in range(min(horizon, len(acts))):
            (st,), o, r, d = port_gym_step(port_lib, [st], cfg, acts[t])
            assert int(d[0]) == int(done[t]), f"{case}: done differs at step {t}"
            if done[t]:
                break
            assert np.abs(o - obs[t]).max() < 1e-3, f"{case}: obs differs at step {t}: {np.abs(o - obs[t]).max()}"
            assert np.abs(r - rew[t]).max() < 1e-3, f"{case}: reward differs at step {t}"


def test_cabi_exports_every_declared_symbol():
    """librlgpu.so loads on a CPU-only box and exports every function include/rlgpu.h declares (no compute calls)."""
    hdr = open(os.path.join(ROOT, "include", "rlgpu.h")).read()
    declared = set(re.findall(r"\b(rlgpu_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rlgpu_last_error"}  # mentioned in a comment only
    from rlgymppo_cpp_amd import _lib
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in rlgpu.h but not exported by librlgpu.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    cfg = _lib.default_gym_config()
    assert cfg.tick_skip == 8 and cfg.n_actions == 90 and cfg.no_touch_max_steps == 150
    assert C.sizeof(ArenaState) > 0


def test_host_helpers_without_gpu():
    from rlgymppo_cpp_amd.env import procedural_mesh, action_table
    v, t = procedural_mesh()
    assert v.shape[1] == 3 and t.shape[1] == 3 and t.max() < len(v)
    g = np.load(os.path.join(GOLD, "sim_golden.npz"))
    assert np.array_equal(v, g["mesh_verts"]) and np.array_equal(t, g["mesh_tris"])
    assert np.array_equal(action_table(), g["action_table"])


def test_padded_obs_is_a_block_shuffle_of_default_obs(port_lib):
    """DefaultOBSPadded(maxPlayers = team size) (DefaultOBSPadded.cpp:3-66) on the host port: ball / prev-action / pads / self parts
    equal DefaultOBS; the teammate blocks and the opponent blocks are the same 19-float blocks in a permuted order, and over
    many observations every order occurs."""
    from simlib import port_gym_cfg, port_gym_reset
    from rlgymppo_cpp_amd.state import default_arena
    nc, n = 6, 64
    states = [default_arena(nc) for _ in range(n)]
    _, plain = port_gym_reset(port_lib, states, port_gym_cfg(), run_setter=True)
    _, padded = port_gym_reset(port_lib, states, port_gym_cfg(obs_max_players=3), run_setter=True)
    assert plain.shape == padded.shape == (n * nc, 51 + 19 * nc)
    assert np.array_equal(plain[:, :70], padded[:, :70])
    blocks = lambda o, a, b: o[:, 70 + 19 * a: 70 + 19 * b].reshape(len(o), b - a, 19)
    orders = set()
    for lo, hi in ((0, 2), (2, 5)):   # 2 teammates, 3 opponents
        P, Q = blocks(plain, lo, hi), blocks(padded, lo, hi)
        for r in range(len(P)):
            perm = []
            for q in Q[r]:
                m = [i for i in range(hi - lo) if np.array_equal(P[r][i], q)]
                assert m, "a padded block is not one of the DefaultOBS blocks"
                perm.append(m[0])
            assert sorted(perm) == list(range(hi - lo))
            if hi - lo == 3: orders.add(tuple(perm))
    assert len(orders) == 6   # all 3! opponent orders show up
    # wider padding (maxPlayers 4 for 3v3): 3 mate slots (2 real + 1 zero block), 4 opponent slots (3 real + 1 zero), width 51 + 19 * 8
    _, wide = port_gym_reset(port_lib, states, port_gym_cfg(obs_max_players=4), run_setter=True)
    assert wide.shape == (n * nc, 51 + 38 * 4) and np.array_equal(plain[:, :70], wide[:, :70])
    zero_pos = set()
    for (lo, hi), (wlo, whi) in (((0, 2), (0, 3)), ((2, 5), (3, 7))):
        P, Q = blocks(plain, lo, hi), blocks(wide, wlo, whi)
        for r in range(len(P)):
            real = [q for q in Q[r] if q.any()]
            assert len(real) == hi - lo and len(Q[r]) - len(real) == 1
            assert sorted(map(tuple, real)) == sorted(map(tuple, P[r]))
            zero_pos.add((wlo, int(np.argmin([q.any() for q in Q[r]]))))
    assert len(zero_pos) == 3 + 4   # the zero block lands in every slot of both lists over 384 observations
    # 1v1 with maxPlayers 2: no real mate, one zero mate block; one real and one zero opponent
    s1 = [default_arena(2) for _ in range(8)]
    _, p1 = port_gym_reset(port_lib, s1, port_gym_cfg(), run_setter=True)
    _, w1 = port_gym_reset(port_lib, s1, port_gym_cfg(obs_max_players=2), run_setter=True)
    assert w1.shape == (16, 51 + 38 * 2) and np.array_equal(p1[:, :70], w1[:, :70]) and not w1[:, 70:89].any()
    for r in range(16):
        opp = w1[r, 89:127].reshape(2, 19)
        assert sum(np.array_equal(b, p1[r, 70:89]) for b in opp) == 1 and sum(not b.any() for b in opp) == 1
