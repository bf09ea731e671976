"""How often does the stepper LOSE contact points (a body touching a third mesh object with points at once; a car-car point beyond
the env's pair pool)?  Random play on the 16-file tessellated arena (one collision object per file: the case with more than two mesh objects in
reach of one body) and on the one-object procedural arena, 1v1 / 2v2 / 3v3; prints the events next to the env-ticks they happened in.
usage: lost_contacts.py [envs] [launches of 32 gym steps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0); T = 32
tess = os.path.join(bench.make_tessellated_mesh_dir()[0], "soccar")
for mesh_name, mesh in (("tessellated, 16 objects", tess), ("procedural, 1 object", "procedural")):
    for team in (1, 2, 3):
        env = BatchedEnv(n, team, mesh=mesh)
        N, D = env.n_agents, env.obs_size
        core = PPOCore(D, env.n_actions, (64, 64), (64, 64), use_bf16=True, max_rows=max(4096, N))
        obs = torch.zeros((T + 1, N, D), device=dev); acts = torch.zeros((T, N), dtype=torch.int32, device=dev); logp = torch.zeros((T, N), device=dev)
        rew = torch.zeros((T, N), device=dev); done = torch.zeros((T, N), dtype=torch.int32, device=dev); torch.cuda.synchronize()
        env.reset(True, obs[0]); env.sync()
        env.overflow_counts(reset=True); env.lost_contact_count(reset=True); env.epa_counts(reset=True)
        for k in range(launches):
            assert env.collect(core, T, obs, acts, logp, rew, done); env.sync()
            obs[0].copy_(obs[T]); torch.cuda.synchronize()
        ticks = n * launches * T * 8
        ovf, lost, epa = env.overflow_counts(), env.lost_contact_count(), env.epa_counts()
        print(f"{mesh_name:26s} {team}v{team}: {ticks / 1e6:7.1f} M env-ticks, lost-contact events {lost} ({lost / (ticks / 1e6):.3f} per M env-ticks), "
              f"queue overflows (exact fallback) {ovf}, EPA queries {epa}")
        env.close(); core.close()
