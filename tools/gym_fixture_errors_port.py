"""The host build of the gym against the reference's recorded gym rollouts (tests/golden/sim_golden.npz gym/*): per case, up to which step
observations and rewards are bit-equal, and the largest errors over the whole rollout (development tool; CPU)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rlgymppo_cpp_amd.state import ArenaState
from simlib import PortSim, gym_compare_obs, gym_cfg_for_case, port_gym_reset, port_gym_step
for fn in ("sim_golden.npz", "sim_golden_one_team.npz"):
    sg = np.load(os.path.join(ROOT, "tests", "golden", fn))
    port = PortSim(); port.set_mesh(sg["mesh_verts"], sg["mesh_tris"])
    for case in [str(c) for c in sg["gym_names"]]:
        c5 = sg[f"gym/{case}/cfg"]; team, tick_skip, omp, rk, nts = [int(x) for x in c5[:5]]
        one_team = len(c5) > 5 and int(c5[5]) == 0
        cfg = gym_cfg_for_case(team, tick_skip, omp, rk, nts); cfg.one_team = 1 if one_team else 0
        nc = 2 * team
        st = ArenaState.from_buffer_copy(sg[f"gym/{case}/start_raw" if f"gym/{case}/start_raw" in sg.files else f"gym/{case}/start"].tobytes())
        (st,), obs0 = port_gym_reset(port, [st], cfg, run_setter=False)
        acts = sg[f"gym/{case}/actions"]; obs = sg[f"gym/{case}/obs"]; rew = sg[f"gym/{case}/rew"]; done = sg[f"gym/{case}/done"]
        first_bad = None; worst_r = 0.0
        for t in range(len(acts)):
            (st,), o, r, d = port_gym_step(port, [st], cfg, acts[t])
            worst_r = max(worst_r, float(np.abs(r - rew[t]).max()))
            exact = np.array_equal(r, rew[t])
            if not done[t]:
                try:
                    gym_compare_obs(o, obs[t], nc, omp, [int(x) for x in sg[f"gym/{case}/player_order"][t]], 1e-30, case, one_team)
                except AssertionError:
                    exact = False
            if not exact and first_bad is None: first_bad = t
            if done[t]: break
        print(f"{case:34s} steps {t + 1:4d}  exact {'all' if first_bad is None else 'until step %d' % first_bad}  max |reward diff| {worst_r:.3g}")
