import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.state import ArenaState
sg = np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))
case = "3v3_allterms_random"
team, tick_skip, omp, rk, nts = [int(x) for x in sg[f"gym/{case}/cfg"][:5]]
gcfg = T._gym_cfg(team, tick_skip, omp, rk, nts)
env = BatchedEnv(1, team, cfg=gcfg, mesh=(sg["mesh_verts"], sg["mesh_tris"]))
st = ArenaState.from_buffer_copy(sg[f"gym/{case}/start_raw" if f"gym/{case}/start_raw" in sg.files else f"gym/{case}/start"].tobytes()); env.upload_states([st]); env.reset(False)
dev = torch.device("cuda", 0); rows = env.n_agents
nobs = torch.empty((rows, env.obs_size), device=dev); r = torch.empty(rows, device=dev); d = torch.empty(rows, dtype=torch.int32, device=dev)
acts = sg[f"gym/{case}/actions"]; obs = sg[f"gym/{case}/obs"]; rew = sg[f"gym/{case}/rew"]
order = sg[f"gym/{case}/player_order"]
for t in range(3):
    env.step(torch.from_numpy(acts[t].astype(np.int32)).to(dev), nobs, r, d); env.sync()
    got = nobs.cpu().numpy(); ref = obs[t]
    print("step", t, "player order", order[t], "reward equal", np.array_equal(r.cpu().numpy(), rew[t]), r.cpu().numpy() - rew[t])
    for row in range(rows):
        diff = np.nonzero(got[row, :70] != ref[row, :70])[0]
        if len(diff): print("  row", row, "cols", diff[:10], "got", got[row, diff[:5]], "ref", ref[row, diff[:5]])

# the host build on the same start / config / actions: which of the two does the HIP path agree with?
from simlib import PortSim, port_gym_cfg, port_gym_reset, port_gym_step, gym_cfg_for_case
port = PortSim(); port.set_mesh(sg["mesh_verts"], sg["mesh_tris"])
pcfg = gym_cfg_for_case(team, tick_skip, omp, rk, nts)
hs, hobs = port_gym_reset(port, [ArenaState.from_buffer_copy(sg[f"gym/{case}/start_raw" if f"gym/{case}/start_raw" in sg.files else f"gym/{case}/start"].tobytes())], pcfg, run_setter=False)
env2 = BatchedEnv(1, team, cfg=gcfg, mesh=(sg["mesh_verts"], sg["mesh_tris"]))
env2.upload_states([ArenaState.from_buffer_copy(sg[f"gym/{case}/start_raw" if f"gym/{case}/start_raw" in sg.files else f"gym/{case}/start"].tobytes())]); env2.reset(False)
for t in range(2):
    env2.step(torch.from_numpy(acts[t].astype(np.int32)).to(dev), nobs, r, d); env2.sync()
    hs, ho, hr, hd = port_gym_step(port, hs, pcfg, acts[t].astype(np.int32))
    got = nobs.cpu().numpy()
    print("step", t, "HIP == port obs:", np.array_equal(got, ho), " port == fixture obs (first 70 cols, row 0):", np.array_equal(ho[0, :70], obs[t][0, :70]), " max |HIP - port|", np.abs(got - ho).max())
    env2.upload_states(hs)   # (the port hands its state over in uu every step: keep the HIP env on the same footing)
