"""The GPU gym-vs-port test as a debugging loop (development tool; GPU box): 64 random envs stepped on the HIP path and on the host build from
the same states; at the first env / step whose observation rows differ, the pre-step state is replayed tick by tick on both and dumped."""
import os, sys, pickle, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rlgymppo_cpp_amd.env import BatchedEnv, action_table
from rlgymppo_cpp_amd import _lib
from rlgymppo_cpp_amd.state import ArenaState, default_arena
from simlib import PortSim, port_gym_cfg, port_gym_reset, port_gym_step, state_vec
port = PortSim(); v, t = port.procedural_mesh(); port.set_mesh(v, t)
n = 64
cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 20
env = BatchedEnv(n, 1, cfg=cfg); pcfg = port_gym_cfg(no_touch_max_steps=20)
obs = env.reset(True); env.sync()
hs, hobs = port_gym_reset(port, [default_arena(2) for _ in range(n)], pcfg, run_setter=True)
dev = torch.device("cuda", 0); rng = np.random.RandomState(5)
nobs = torch.empty_like(obs); rew = torch.empty(n * 2, device=dev); done = torch.empty(n * 2, dtype=torch.int32, device=dev)
tab = action_table()
for step in range(48):
    acts = rng.randint(0, 90, size=n * 2).astype(np.int32)
    before = [ArenaState.from_buffer_copy(bytes(s)) for s in hs]
    env.epa_counts(reset=True)
    env.step(torch.from_numpy(acts).to(dev), nobs, rew, done); env.sync()
    hs, ho, hr, hd = port_gym_step(port, hs, pcfg, acts)
    d = np.abs(nobs.cpu().numpy() - ho).max(axis=1)
    print("step", step, "max obs diff %.3g" % d.max(), "EPA", env.epa_counts())
    if d.max() > 1e-6:
        row = int(d.argmax()); e = row // 2
        print("  env", e, "row", row, "cols", np.nonzero(np.abs(nobs.cpu().numpy()[row] - ho[row]) > 1e-6)[0][:20])
        s = ArenaState.from_buffer_copy(bytes(before[e]))
        for k in range(2): s.cars[k].controls[:] = list(tab[acts[2 * e + k]])
        one = BatchedEnv(1, 1, cfg=cfg)
        h = ArenaState.from_buffer_copy(bytes(s))
        one.upload_states([s])
        for tick in range(8):
            one.epa_counts(reset=True)
            one.physics_ticks(1); cur = one.download_states()[0]
            port.step(h, 1)
            a, b = state_vec(h), state_vec(cur)
            print("   tick", tick, "max |diff|", np.abs(a - b).max(), "at", int(np.abs(a - b).argmax()), "EPA", one.epa_counts())
            if np.abs(a - b).max() > 0:
                pickle.dump({"state": bytes(s), "tick": tick}, open(os.path.join(ROOT, "gpurun_out", "dbg_gym_state.pkl"), "wb"))
                break
            one.upload_states([h])   # both continue from the host's bits
        break
    env.upload_states(hs)
print("done")
