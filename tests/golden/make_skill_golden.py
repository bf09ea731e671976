"""Generates tests/golden/skill_golden.json from the REAL reference skill tracker (oracle/_ref/libref_skill.so = PRIV/Util/SkillTracker.cpp
and what it needs, compiled unedited from /root/reference by `make -C oracle ref_skill`).  Run here, where /root/reference exists; the GPU box
only reads the committed fixture.

  elo     SkillTracker::UpdateRatings (SkillTracker.cpp:72-86) over a scripted sequence of (winner, loser, updateWinner, updateLoser) on four
          rating sets: the ratings after every step, as float32 bit patterns.
  idle    SkillTracker::RunGames (SkillTracker.cpp:152-257) over a scripted sequence of timestep deltas with kickoff states (nobody scores
          in one env step): the version bookkeeping after every call.
  goals   the same with every episode starting with the ball behind the ORANGE goal line, and again behind the BLUE one: every evaluating
          call is one env step = one goal; which policy gets the rating points depends on the env's teamSwap (drawn from a wall-clock seeded
          engine: Math.cpp:59-64), recorded before every call.
The script also checks the rule the product follows against these recordings: ball y > 0 means the policy playing blue scored, the current policy
plays blue unless teamSwap (SkillTracker.cpp:104-146)."""
import ctypes as C
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import simlib  # noqa: E402

_f = C.POINTER(C.c_float)


def _ptr(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def main():
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libref_skill.so"))
    port = simlib.PortSim()
    verts, tris = port.procedural_mesh()
    root = tempfile.mkdtemp(prefix="skill_mesh_")
    simlib.write_cmf_parts(verts, tris, [len(tris)], root)
    assert lib.refs_init_dir(root.encode()) == 0

    gold = {}
    # ---- elo ----
    rng = np.random.default_rng(11)
    n_sets, n = 4, 400
    ratings = np.array([1000, 1000, 1180.5, 640.25], np.float32)
    winner = rng.integers(0, n_sets, n).astype(np.int32)
    loser = ((winner + rng.integers(1, n_sets, n)) % n_sets).astype(np.int32)
    flags = rng.choice([3, 3, 3, 1, 2], n).astype(np.int32)
    trace = np.zeros((n, n_sets), np.float32)
    start = ratings.copy()
    assert lib.refs_elo_script(n_sets, _ptr(ratings), n, _ptr(winner, C.c_int), _ptr(loser, C.c_int), _ptr(flags, C.c_int), C.c_float(5.0), _ptr(trace)) == 0
    gold["elo"] = {"rating_inc": 5.0, "start_bits": start.view(np.uint32).tolist(), "winner": winner.tolist(), "loser": loser.tolist(), "flags": flags.tolist(),
                   "trace_bits": trace.view(np.uint32).tolist()}

    # ---- RunGames scripts ----
    def run(goal_sign, deltas, update_interval, per_version, max_versions, start_with_version):
        d = np.asarray(deltas, np.int64)
        rows = np.zeros((len(d), 6 + max_versions), np.float32)
        lib.refs_run_script.argtypes = [C.c_float, C.c_int, C.POINTER(C.c_int64), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, _f]
        rc = lib.refs_run_script(goal_sign, len(d), _ptr(d, C.c_int64), update_interval, per_version, max_versions, int(start_with_version), 5.0, 0.1, _ptr(rows))
        assert rc == 0, rc
        return {"goal_sign": goal_sign, "deltas": list(map(int, d)), "update_interval": update_interval, "timesteps_per_version": per_version,
                "max_versions": max_versions, "start_with_version": bool(start_with_version), "rating_inc": 5.0, "sim_time": 0.1,
                "columns": ["teamSwap_before", "oldPolicyIndex_before", "runCounter", "nVersions", "timestepsSinceVersionMade", "curRating", "oldRatings..."],
                "rows_bits": rows.view(np.uint32).tolist()}

    gold["idle"] = [run(0.0, [600] * 9, 2, 1000, 2, True), run(0.0, [400, 700, 100, 1200, 50, 50, 2000], 1, 1000, 3, False)]
    gold["goals"] = [run(+1.0, [300] * 40, 1, 1000, 3, True), run(-1.0, [300] * 40, 1, 1000, 3, True), run(+1.0, [500] * 24, 2, 900, 2, False)]

    # ---- the rule, checked against the recordings ----
    def f32(bits):
        return np.array(bits, np.uint32).view(np.float32)

    def elo(w, l, inc):   # float32 arithmetic of SkillTracker.cpp:78-85 (powf through numpy's float32 power)
        w, l, inc = np.float32(w), np.float32(l), np.float32(inc)
        exp_delta = np.float32((l - w) / np.float32(400))
        expected = np.float32(np.float32(1) / np.float32(np.power(np.float32(10), exp_delta, dtype=np.float32) + np.float32(1)))
        return np.float32(w + inc * np.float32(np.float32(1) - expected)), np.float32(l + inc * np.float32(expected - np.float32(1)))

    for s in gold["goals"]:
        rows = f32(s["rows_bits"]).reshape(len(s["deltas"]), -1)
        cur = np.float32(1000); olds = []; counter = 0; since = 0
        for i, r in enumerate(rows):
            swap, idx = bool(r[0]), int(r[1])
            evaluates = counter % s["update_interval"] == 0
            counter += 1
            if evaluates:
                if not olds and s["start_with_version"]:
                    olds.append(cur)
                if olds:
                    blue_scored = s["goal_sign"] > 0
                    cur_scored = blue_scored != swap
                    if cur_scored:
                        cur, olds[idx] = elo(cur, olds[idx], s["rating_inc"])
                    else:
                        olds[idx], cur = elo(olds[idx], cur, s["rating_inc"])
                since += s["deltas"][i]
                if since >= s["timesteps_per_version"]:
                    since = 0; olds.append(cur)
                    if len(olds) > s["max_versions"]:
                        olds.pop(0)
            assert int(r[2]) == counter and int(r[3]) == len(olds) and int(r[4]) == since, (i, r, counter, len(olds), since)
            assert abs(float(r[5]) - float(cur)) < 1e-3, (i, r[5], cur)
            for k, o in enumerate(olds):
                assert abs(float(r[6 + k]) - float(o)) < 1e-3, (i, k, r[6 + k], o)
    with open(os.path.join(HERE, "skill_golden.json"), "w") as f:
        json.dump(gold, f)
    print("wrote skill_golden.json:", {k: (len(v) if isinstance(v, list) else len(v["winner"])) for k, v in gold.items()})


if __name__ == "__main__":
    main()
