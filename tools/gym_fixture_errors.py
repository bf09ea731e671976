"""How close the HIP gym is to the reference's recorded gym rollouts (tests/golden/sim_golden.npz gym/*): per case the largest reward error
and the smallest observation tolerance from a ladder that every step passes (development tool; GPU box)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.state import ArenaState
from simlib import gym_compare_obs, gym_cfg_for_case, GYM_HORIZON
sg = np.load(os.path.join(ROOT, "tests", "golden", sys.argv[1] if len(sys.argv) > 1 else "sim_golden.npz"))     # or sim_golden_one_team.npz
dev = torch.device("cuda", 0)
LADDER = [0.0, 1e-7, 1e-6, 1e-5, 1e-4, 2e-3, 1e-2]
for case in [str(c) for c in sg["gym_names"]]:
    team, tick_skip, omp, rk, nts = [int(x) for x in sg[f"gym/{case}/cfg"][:5]]
    nc = 2 * team
    gcfg = gym_cfg_for_case(team, tick_skip, omp, rk, nts)
    one_team = len(sg[f"gym/{case}/cfg"]) > 5 and int(sg[f"gym/{case}/cfg"][5]) == 0
    gcfg.one_team = 1 if one_team else 0
    acts = sg[f"gym/{case}/actions"]; obs = sg[f"gym/{case}/obs"]; rew = sg[f"gym/{case}/rew"]; done = sg[f"gym/{case}/done"]
    worst_tol = 0; worst_rew = 0.0; steps = 0
    for tol in LADDER:
        env = BatchedEnv(1, team, cfg=gcfg, mesh=(sg["mesh_verts"], sg["mesh_tris"]))
        env.upload_states([ArenaState.from_buffer_copy(sg[f"gym/{case}/start_raw" if f"gym/{case}/start_raw" in sg.files else f"gym/{case}/start"].tobytes())]); env.reset(False)
        nobs = torch.empty((env.n_agents, env.obs_size), device=dev); r = torch.empty(env.n_agents, device=dev); d = torch.empty(env.n_agents, dtype=torch.int32, device=dev)
        ok = True; worst_rew = 0.0; steps = 0
        try:
            for t in range(min(len(acts), GYM_HORIZON.get(case, len(acts)))):
                env.step(torch.from_numpy(acts[t].astype(np.int32)).to(dev), nobs, r, d); env.sync()
                worst_rew = max(worst_rew, float(np.abs(r.cpu().numpy() - rew[t]).max())); steps += 1
                if done[t]: break
                gym_compare_obs(nobs.cpu().numpy(), obs[t], nc, omp, [int(x) for x in sg[f"gym/{case}/player_order"][t]], tol if tol > 0 else 1e-30, f"{case} step {t}", one_team)
        except AssertionError:
            ok = False
        env.close()
        if ok:
            worst_tol = tol; break
    print(f"{case:34s} steps {steps:4d}  obs within {worst_tol:g}  max |reward diff| {worst_rew:.3g}")
