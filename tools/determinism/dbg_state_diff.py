"""Two env batches under the same flow of fused collection launches (tools/determinism/dbg_free.py, lock mode): after every launch the resident
states of both batches are downloaded and compared byte by byte; prints the first launch after which they differ, the envs and the byte offsets."""
import sys, os, numpy as np, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
from rlgymppo_cpp_amd.state import ArenaState, CarState, BallState
from rlgymppo_cpp_amd import _lib
team, n_envs, use_bf16, L = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == '1', int(sys.argv[4])
dev = torch.device("cuda", 0); CAP = 12
cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 9
ea, eb = BatchedEnv(n_envs, team, cfg), BatchedEnv(n_envs, team, cfg)
core = PPOCore(ea.obs_size, ea.n_actions, (64, 64), (64, 64), use_bf16=use_bf16, max_rows=4096)
N, D = ea.n_agents, ea.obs_size
def bufs():
    return (torch.zeros((CAP + 1, N, D), device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev), torch.zeros((CAP, N), device=dev),
            torch.full((CAP, N), -777.0, device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev))
A, B = bufs(), bufs(); torch.cuda.synchronize()
stream, ctr = core.get_sampler()
ea.reset(True, A[0][0]); eb.reset(True, B[0][0]); ea.sync(); eb.sync()
off = {name: getattr(ArenaState, name).offset for name in ("ball", "cars", "pads", "gym", "hidden")}
car_fields = {f[0]: getattr(CarState, f[0]).offset for f in CarState._fields_}
scrub = None
if os.environ.get('SCRUB'):
    scrub = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'probes', 'libscrub_probe.so')); pat = C.c_uint(int(os.environ['SCRUB'], 0))
for k in range(L):
    s, c = core.get_sampler()
    if scrub: scrub.scrub_probe(pat)
    assert ea.collect(core, CAP, *A); ea.sync()
    core.set_sampler(s, c)
    if scrub: scrub.scrub_probe(pat)
    assert eb.collect(core, CAP, *B); eb.sync()
    outs_equal = all(bool((x == y).all()) for x, y in zip(A, B))
    for nm, x, y in zip(("obs", "acts", "logp", "rew", "done"), A, B):
        d = (x != y)
        if bool(d.any()):
            idx = d.nonzero()[:5].tolist()
            print(f"   {nm}: {int(d.sum())} differing, first (step, row[, feature]) {idx}")
    sa, sb = ea.download_states(), eb.download_states()
    bad = [e for e in range(n_envs) if bytes(sa[e]) != bytes(sb[e])]
    print(f"launch {k}: outputs equal {outs_equal}; states differing in {len(bad)} envs {bad[:8]}")
    for e in bad[:3]:
        x, y = np.frombuffer(bytes(sa[e]), np.uint8), np.frombuffer(bytes(sb[e]), np.uint8)
        d = np.flatnonzero(x != y)
        where = []
        for o in d[:6]:
            sec = max((n for n in off if off[n] <= o), key=lambda n: off[n])
            extra = ""
            if sec == "cars":
                ci, co = divmod(int(o) - off["cars"], C.sizeof(CarState)); fld = max((n for n in car_fields if car_fields[n] <= co), key=lambda n: car_fields[n]); extra = f"[{ci}].{fld}+{co - car_fields[fld]}"
            where.append(f"{int(o)}:{sec}{extra}")
        print("   env", e, "bytes", len(d), where)
        for nm, st in (("A", sa[e]), ("B", sb[e])):
            print("     ", nm, "car0 last_controls", [round(v, 3) for v in st.cars[0].last_controls], "controls", [round(v, 3) for v in st.cars[0].controls], "car1 last", [round(v, 3) for v in st.cars[1].last_controls], "tick", st.tick_count, "steps", st.gym.episode_steps, "resets", st.gym.reset_count)
        print("      acts last 3 steps A", A[1][-3:, 2 * e:2 * e + 2].tolist(), "B", B[1][-3:, 2 * e:2 * e + 2].tolist(), "done A", A[4][-3:, 2 * e].tolist(), "B", B[4][-3:, 2 * e].tolist())
    if bad: break
    A[0][0].copy_(A[0][CAP]); B[0][0].copy_(B[0][CAP]); torch.cuda.synchronize()
