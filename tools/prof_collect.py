"""Profiler build only (RLGPU_LIB=.../librlgpu_prof.so): per-workgroup cycles of the fused collection kernel (rlgpu_collect)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
n, T = 4096, 32
mesh = "procedural"
if len(sys.argv) > 1 and sys.argv[1] == "tess":      # the arena at ~10 k triangles in 16 .cmf files (bench.py --mesh tessellated)
    import bench
    mesh, info = bench.make_tessellated_mesh_dir(); mesh = os.path.join(mesh, "soccar"); print("mesh:", info)
env = BatchedEnv(n, 1, mesh=mesh); dev = torch.device("cuda", 0)
N, D = env.n_agents, env.obs_size
ppo = PPOCore(D, 90, (256, 256, 256), (256, 256, 256), use_bf16=True, max_rows=65536, seed=1)
obs = torch.zeros((T + 1, N, D), device=dev); acts = torch.zeros((T, N), dtype=torch.int32, device=dev); logp = torch.zeros((T, N), device=dev)
rew = torch.zeros((T, N), device=dev); done = torch.zeros((T, N), dtype=torch.int32, device=dev)
env.reset(True, obs[0])
fn = env.lib.rlgpu_env_debug_step_prof; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; fn.restype = C.c_int
nb = (n + 3) // 4
prev = None
for it in range(12):
    assert env.collect(ppo, T, obs, acts, logp, rew, done)
    obs[0].copy_(obs[T])
    buf = np.zeros(16 * nb, dtype=np.uint64)
    assert fn(env.h, buf.ctypes.data, nb) == 0
    tot = buf.reshape(-1, 16)[:, 0].astype(np.float64); inf = buf.reshape(-1, 16)[:, 1].astype(np.float64); mlp = buf.reshape(-1, 16)[:, 2].astype(np.float64)
    if prev is not None and it >= 6:   # do slow workgroups stay slow?  (they keep their envs from launch to launch)
        r = np.corrcoef(prev, tot)[0, 1]
        top = np.argsort(-prev)[: nb // 10]
        print("launch %d: correlation of workgroup cycles with the previous launch %.2f; last launch's slowest 10 %% now average %.2fM (all: %.2fM, max %.2fM)" % (it, r, tot[top].mean() / 1e6, tot.mean() / 1e6, tot.max() / 1e6))
    epa = buf.reshape(-1, 16)[:, 3].astype(np.float64)
    if it >= 6:
        slow = np.argsort(-tot)[:8]
        print("   slowest workgroups (index: Mcycles, inference Mcycles):", ", ".join("%d: %.1f, %.1f" % (int(w), tot[w] / 1e6, inf[w] / 1e6) for w in slow[:4]))
        print("   penetration-depth queries per workgroup: mean %.1f max %d; correlation with the workgroup's cycles %.2f; the 8 slowest workgroups: %s queries, %s Mcycles"
              % (epa.mean(), int(epa.max()), np.corrcoef(epa, tot)[0, 1], [int(x) for x in epa[slow]], [round(x / 1e6, 1) for x in tot[slow]]))
    if it >= 6 and tot[slow[0]] > 1.3 * tot[slow[1]]:   # an outlier workgroup: what are its envs doing (state after the launch)?
        st = env.download_states()
        w = int(slow[0])
        for e in range(4 * w, 4 * w + 4):
            s_ = st[e]; desc = []
            for k in range(2):
                c = s_.cars[k]; f = c.flags
                desc.append("car%d pos (%.0f %.0f %.0f) up.z %+.2f |v| %.0f |w| %.1f %s%s%s wheels %s boost %.0f" % (k, c.pos[0], c.pos[1], c.pos[2], c.rot[8], np.linalg.norm(c.vel[:]), np.linalg.norm(c.ang_vel[:]),
                            'G' if f & 1 else '-', 'W' if f & (1 << 12) else '-', 'D' if f & (1 << 13) else '-', format((f >> 1) & 15, '04b'), c.boost))
            print("      env %d: ball (%.0f %.0f %.0f) |v| %.0f | %s" % (e, s_.ball.pos[0], s_.ball.pos[1], s_.ball.pos[2], np.linalg.norm(s_.ball.vel[:]), " | ".join(desc)))
    prev = tot
    if it >= 9:
        tk = buf.reshape(-1, 16)[:, 4].astype(np.float64); gy = buf.reshape(-1, 16)[:, 5].astype(np.float64)
        print("launch %d: per gym step (mean workgroup): total %.0fK cycles = ticks %.0fK (%.1fK per tick) + inference %.0fK + gym bookkeeping (snapshot, events, rewards, done, obs rows, reset) %.0fK + rest %.0fK"
              % (it, tot.mean() / T / 1e3, tk.mean() / T / 1e3, tk.mean() / T / 8e3, inf.mean() / T / 1e3, gy.mean() / T / 1e3, (tot - tk - inf - gy).mean() / T / 1e3))
        b = buf.reshape(-1, 16).astype(np.float64)
        print("   inference per step: fence + observation staging %.1fK, layers %s K cycles" % (b[:, 7].mean() / T / 1e3, [round(b[:, 8 + q].mean() / T / 1e3, 1) for q in range(4)]))
        q = np.percentile(tot, [1, 50, 90, 99])
        print("launch %d: workgroup cycles min %.2fM p1 %.2fM median %.2fM mean %.2fM p90 %.2fM p99 %.2fM max %.2fM (%.1f ms); inference share of the mean %.1f%% (per step: MLP %.0fK cycles, head %.0fK); mean/max %.2f"
              % (it, tot.min() / 1e6, q[0] / 1e6, q[1] / 1e6, tot.mean() / 1e6, q[2] / 1e6, q[3] / 1e6, tot.max() / 1e6, tot.max() / 2.38e6, 100 * inf.mean() / tot.mean(), mlp.mean() / T / 1e3, (inf.mean() - mlp.mean()) / T / 1e3, tot.mean() / tot.max()))

try:
    env.lib.rlgpu_env_debug_ints.argtypes = [C.c_void_p, C.c_void_p]
    dbg = (C.c_int * 64)(); env.lib.rlgpu_env_debug_ints(env.h, dbg)
    if dbg[13]: print("candidate walks: %d env-ticks, %.1f %% of them walk, %.1f %% of the walks did not fit their fat boxes (redone with boxes grown by a quarter of that, list kept), %.2f %% not those either (exact boxes, list not kept)" % (dbg[13], 100.0 * dbg[11] / dbg[13], 100.0 * dbg[12] / max(1, dbg[11]), 100.0 * dbg[14] / max(1, dbg[11])))
except Exception as ex:
    print("no debug ints:", ex)
