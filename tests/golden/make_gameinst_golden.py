"""Generates tests/golden/gameinst_golden.npz from the REAL reference: GameInst::Start / GameInst::Step (PUB/Threading/GameInst.cpp:3-38,
compiled unedited into oracle/_ref/libref_oracle.so next to oracle/ref_driver.cpp) over SEVERAL episode ends per case -- the terminal
observation replaced by the first one of the new episode, the reward trackers rolled over -- with a user state setter written against the
reference's plugin surface whose k-th call installs the k-th state of a list (the reference's own setters draw from a wall-clock-seeded
std RNG and cannot be replayed).  Run in the build container: `python tests/golden/make_gameinst_golden.py`.

Per case <c>:  gi/<c>/cfg = team, tick_skip, obs_max_players, reward_kind, no_touch_steps;  states [n_states] ArenaState bytes (what the
setter installs, call by call);  actions [T][players] by slot;  cur_obs [T + 1][players][D] = GameInst::curObs after Start() and after every
Step;  rew [T][players];  done [T];  trackers [T][6] = curEpRew, avgEpRew.total, avgEpRew.count, avgStepRew.total, avgStepRew.count,
totalSteps;  resets [T + 1] = state-setter calls so far;  order [T + 1][players] = the gym's GameState::players order (slots).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from simlib import PortSim, RefSim, _ptr  # noqa: E402
from rlgymppo_cpp_amd.state import ArenaState, default_arena, yaw_rot  # noqa: E402


def start_states(nc, n, rs, toward_goal):
    """n start states: cars spread over the field on their wheels, the ball somewhere in the air -- or, toward_goal, a few hundred uu in front of
    a goal line flying in (blue's and orange's goal in turn), so that episodes end by GoalScoreCondition."""
    out = []
    for k in range(n):
        s = default_arena(nc)
        for c in range(nc):
            s.cars[c].pos[:] = (float(rs.uniform(-3000, 3000)), float(rs.uniform(-4000, 4000)), 17.0)
            s.cars[c].rot[:] = yaw_rot(float(rs.uniform(-np.pi, np.pi)))
            s.cars[c].vel[:] = (float(rs.uniform(-500, 500)), float(rs.uniform(-500, 500)), 0.0)
            s.cars[c].boost = float(rs.uniform(0, 100))
        if toward_goal:
            sign = 1.0 if k % 2 == 0 else -1.0
            s.ball.pos[:] = (float(rs.uniform(-300, 300)), sign * float(rs.uniform(4300, 4800)), float(rs.uniform(150, 400)))
            s.ball.vel[:] = (float(rs.uniform(-100, 100)), sign * float(rs.uniform(1500, 2500)), float(rs.uniform(-100, 200)))
        else:
            s.ball.pos[:] = (float(rs.uniform(-2500, 2500)), float(rs.uniform(-3500, 3500)), float(rs.uniform(100, 1200)))
            s.ball.vel[:] = (float(rs.uniform(-800, 800)), float(rs.uniform(-800, 800)), float(rs.uniform(-300, 300)))
        out.append(s)
    return out


CASES = {   # name: (team, obs_max_players, reward_kind, no_touch_steps, steps, toward_goal)
    "1v1_timeouts": (1, 0, 0, 7, 60, False),            # the example stack; NoTouchCondition(7) ends an episode every 7 steps
    "1v1_goals": (1, 0, 0, 150, 60, True),              # GoalScoreCondition: the EventReward's +-50 in the step that ends the episode
    "2v2_allterms_zerosum_timeouts": (2, 0, 3, 5, 36, False),
    "3v3_padded3_goals_and_timeouts": (3, 3, 1, 6, 40, True),
}


def main():
    port = PortSim(); verts, tris = port.procedural_mesh()
    ref = RefSim(verts, tris)
    out = {"mesh_verts": verts, "mesh_tris": tris, "names": np.array(list(CASES))}
    for name, (team, omp, rk, nts, T, goal) in CASES.items():
        rs = np.random.RandomState(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        nc = 2 * team
        n_states = T + 2
        states = start_states(nc, n_states, rs, goal)
        arr = (ArenaState * n_states)(*states)
        actions = rs.randint(0, 90, size=(T, nc)).astype(np.int32)
        D = 51 + 38 * omp if omp > 0 else 51 + 19 * nc
        cur = np.zeros((T + 1, nc, D), np.float32); stp = np.zeros((T, nc, D), np.float32)
        rew = np.zeros((T, nc), np.float32); done = np.zeros(T, np.int32); tr = np.zeros((T, 6), np.float32)
        resets = np.zeros(T + 1, np.int32); order = np.zeros((T + 1, nc), np.int32)
        d = ref.lib.ref_gameinst_run(team, 8, omp, rk, nts, arr, n_states, _ptr(actions), T, _ptr(cur), _ptr(stp), _ptr(rew), _ptr(done), _ptr(tr), _ptr(resets), _ptr(order))
        assert d == D, (d, D)
        assert (cur[1:] == stp).all(), "GameInst::Step returns the rows it stores in curObs"
        ends = int(done.sum())
        assert ends >= 3 and resets[-1] == 1 + ends and resets[-1] <= n_states, (name, ends, resets[-1])
        print(f"{name}: {T} steps, {ends} episode ends, avgEpRew {tr[-1, 1]:.4f} / {tr[-1, 2]:.0f}")
        # the order in which the reference's arena lists its cars (Arena::_cars is an unordered_set of pointers; it does not change while the cars live):
        # part of every state the reference reports (car_order, 4 bits per rank = slot + 1), so the states handed to the replay carry it too
        assert (order == order[0]).all()
        word = 0
        for rank, slot in enumerate(order[0]):
            word |= (int(slot) + 1) << (4 * rank)
        for k in range(n_states):
            arr[k].car_order = word
        out[f"gi/{name}/cfg"] = np.array([team, 8, omp, rk, nts], np.int32)
        out[f"gi/{name}/states"] = np.frombuffer(bytes(arr), np.uint8).reshape(n_states, -1)[:int(resets[-1])].copy()
        out[f"gi/{name}/actions"] = actions; out[f"gi/{name}/cur_obs"] = cur; out[f"gi/{name}/rew"] = rew; out[f"gi/{name}/done"] = done
        out[f"gi/{name}/trackers"] = tr; out[f"gi/{name}/resets"] = resets; out[f"gi/{name}/order"] = order
    path = os.path.join(HERE, "gameinst_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
