"""The HIP gym against the LIVE reference Gym (oracle/_ref) on rollouts no fixture holds (round 6; needs a GPU and oracle/_ref -- both are on the gpurun box).

Per seed: 1v1 / 2v2 / 3v3, every car in a ring around the ball at speed (a pile-up in the first second: touches, bumps, demolitions, several cars on the ball)
or the ball rolling at a goal, every CommonRewards term (plain or inside ZeroSumReward), DefaultOBS or DefaultOBSPadded, default mutators or
tests/golden/make_mutator_golden.py's M1 (ON_CONTACT team demolitions, short respawns, ...), a random action of the 90-row table per player and step, until the
episode ends or `steps` steps.  The reference's rollout is recorded in the layout of sim_golden.npz's gym/ entries and replayed by the body of
tests/test_gpu_parity.py::test_hip_gym_rollouts_vs_reference_fixtures with EXACT comparison: done, every reward and every observation row bit for bit.

The reference normalises with `rsqrtss` (btVector3.h:308-346), whose table is the CPU vendor's: the fixtures of tests/golden/ and the stepper's emulation
(csrc/rl_math.h) are this container's Intel core, and on the GPU box's EPYC the SAME reference binary returns other last bits from the first tick on (seen
here: every rollout "different" at step 0 by 1e-7 in the orientation entries while start state, HIP tick, host-build tick and snapshot were all equal).  So the
reference is recorded on an Intel host and replayed on the GPU:
    python tools/live_gym_hip.py --record tools/_live/gym.npz 90 100 1         (here; tools/_live/ is git-ignored and travels with gpurun)
    gpurun -- python tools/live_gym_hip.py --replay tools/_live/gym.npz        (there)
Without --record / --replay both halves run in one process (only meaningful on an Intel host: --port here, or a GPU box that has one).

tests/golden/live_gym_golden.npz is a selection of these rollouts (the ones that found something, and a few that did not), recorded with
    python tools/live_gym_hip.py --record tests/golden/live_gym_golden.npz --seeds 1,6,14,23,33,47,53,55,60,85,104,110,119 0 100

usage: python tools/live_gym_hip.py [--port] [--record FILE | --replay FILE] [--seeds a,b,...] [rollouts] [steps] [first seed]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import simlib  # noqa: E402
from simlib import RefSim, RefGym  # noqa: E402
from rlgymppo_cpp_amd.state import default_arena, HIDDEN_MUTATORS, HIDDEN_REF_ENGINE  # noqa: E402
from make_mutator_golden import mutator_sets  # noqa: E402


class Fixture(dict):
    @property
    def files(self): return list(self.keys())


def reference_rollout(ref, gold, seed, steps, m1, drag1):
    """one rollout of the live reference Gym: (case name, its fixture entries in sim_golden.npz's gym/ layout, a line describing it)"""
    L = ref.lib
    rng = np.random.RandomState(seed)
    team = 1 + seed % 3; nc = 2 * team
    rk = 2 + int(rng.randint(2)); omp = 0 if rng.rand() < 0.6 else int(rng.randint(team, 4)); use_m1 = bool(rng.rand() < 0.5); nts = int(rng.choice([30, 150]))
    s0 = default_arena(nc)
    cx, cy = rng.uniform(-2000, 2000), rng.uniform(-3000, 3000)
    if rng.rand() < 0.3:   # the ball on its way into a goal, the cars behind it
        side = rng.choice([-1.0, 1.0]); s0.ball.pos[:] = [float(rng.uniform(-600, 600)), float(side * rng.uniform(3600, 4400)), float(rng.uniform(93.15, 500))]
        s0.ball.vel[:] = [float(rng.uniform(-200, 200)), float(side * rng.uniform(900, 2200)), float(rng.uniform(-100, 400))]; cx, cy = s0.ball.pos[0], s0.ball.pos[1] - side * 900
    else:
        s0.ball.pos[:] = [float(cx), float(cy), 93.15]
    ang0 = rng.uniform(0, 2 * np.pi)
    for k in range(nc):
        a_k = ang0 + 2 * np.pi * k / nc + rng.uniform(-0.25, 0.25); rad = rng.uniform(600, 1200)
        yaw = a_k + np.pi + rng.uniform(-0.15, 0.15); v = rng.uniform(800, 2250)
        c = s0.cars[k]
        c.pos[:] = [float(cx + rad * np.cos(a_k)), float(cy + rad * np.sin(a_k)), 17.0]
        c.rot[:] = [float(np.cos(yaw)), float(np.sin(yaw)), 0.0, float(-np.sin(yaw)), float(np.cos(yaw)), 0.0, 0.0, 0.0, 1.0]
        c.vel[:] = [float(np.cos(yaw) * v), float(np.sin(yaw) * v), 0.0]; c.boost = float(rng.uniform(20, 100))
    acts = rng.randint(0, 90, size=(steps, nc)).astype(np.int32)
    g = RefGym(ref, team, 8, reward_kind=rk, no_touch_steps=nts, obs_max_players=omp)
    if use_m1: L.ref_arena_set_mutators(g.arena(), C.byref(m1), C.c_float(drag1))
    obs0 = g.reset_to(s0)
    start = ref.get_state(g.arena())
    s0.car_order = start.car_order; s0.mutators = start.mutators; s0.hidden.valid |= HIDDEN_MUTATORS
    engine0 = 1 + (seed * 2654435761) % 2147483645
    L.ref_seed_engine(C.c_uint32(engine0)); s0.hidden.valid |= HIDDEN_REF_ENGINE; s0.hidden.ref_engine = engine0
    obs, rew, done, order, last = [], [], [], [], None
    for t in range(steps):
        o, r, d, st = g.step(acts[t])
        obs.append(o); rew.append(r); done.append(d); order.append(g.player_order()); last = st
        if d: break
    case = f"live_{seed}"
    what = f"seed {seed}: {team}v{team} reward kind {rk} obs_max_players {omp} {'M1' if use_m1 else 'default mutators'} no-touch {nts}: {len(obs)} steps, done {done[-1]}"
    entries = {f"gym/{case}/start_raw": np.frombuffer(bytes(s0), np.uint8).copy(), f"gym/{case}/obs0": obs0, f"gym/{case}/actions": acts[: len(obs)],
               f"gym/{case}/obs": np.stack(obs), f"gym/{case}/rew": np.stack(rew), f"gym/{case}/done": np.array(done, np.int32),
               f"gym/{case}/player_order": np.array(order, np.int32), f"gym/{case}/final": np.frombuffer(bytes(last), np.uint8).copy(),
               f"gym/{case}/cfg": np.array([team, 8, omp, rk, nts], np.int32)}
    return case, entries, what


def main():
    port_mode = "--port" in sys.argv      # against the host build (CPU, tolerances of tests/test_oracle_golden.py): a smoke run of this script
    if port_mode: sys.argv.remove("--port")
    record = replay = None
    for flag in ("--record", "--replay"):
        if flag in sys.argv:
            i = sys.argv.index(flag); path = sys.argv[i + 1]; del sys.argv[i:i + 2]
            if flag == "--record": record = path
            else: replay = path
    seeds = None
    if "--seeds" in sys.argv:
        i = sys.argv.index("--seeds"); seeds = [int(x) for x in sys.argv[i + 1].split(",")]; del sys.argv[i:i + 2]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    gold = np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))
    cases = []          # (name, entries, description)
    if replay:
        rec = np.load(replay)
        for case, what in zip(rec["gym_names"], rec["descriptions"]):
            cases.append((str(case), {k: rec[k] for k in rec.files if k.startswith(f"gym/{case}/")}, str(what)))
    else:
        vendor = [ln.split(":")[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("vendor_id")][:1]
        if vendor != ["GenuineIntel"]: print(f"NOTE: this host's CPU is {vendor}: the reference's rsqrtss results are not the fixtures' here (see the file's head)")
        ref = RefSim(gold["mesh_verts"], gold["mesh_tris"]); L = ref.lib
        L.ref_arena_set_mutators.argtypes = [C.c_void_p, C.c_void_p, C.c_float]; L.ref_engine_state.restype = C.c_uint32
        _, m1, drag1 = mutator_sets()[0]
        for seed in (seeds if seeds is not None else range(seed0, seed0 + n)): cases.append(reference_rollout(ref, gold, seed, steps, m1, drag1))
    if record:
        out = {"gym_names": np.array([c for c, _, _ in cases]), "descriptions": np.array([w for _, _, w in cases])}
        for _, e, _ in cases: out.update(e)
        os.makedirs(os.path.dirname(os.path.abspath(record)), exist_ok=True); np.savez_compressed(record, **out)
        print(f"{len(cases)} reference rollouts ({sum(len(e[f'gym/{c}/obs']) for c, e, _ in cases)} steps) written to {record}"); return
    if port_mode:
        import test_oracle_golden as TO
        port = simlib.PortSim(); port.set_mesh(gold["mesh_verts"], gold["mesh_tris"])
    else:
        import test_gpu_parity as T
    ok = 0; bad = []
    for case, entries, what in cases:
        fx = Fixture({"gym_names": np.array([case]), "mesh_verts": gold["mesh_verts"], "mesh_tris": gold["mesh_tris"], **entries})
        simlib.GYM_EXACT.add(case)
        # DefaultOBSPadded shuffles its lists with RocketSim's own engine (DefaultOBSPadded.cpp:58-59 -> Math::GetRandEngine), the one Car::Respawn draws the spawn
        # slot from: every observation built advances it.  The stepper's replay of that engine (RlgpuArenaHidden::ref_engine, a test facility) covers the arena's
        # draws only -- its own shuffle is keyed Philox (csrc/arena_gym.h build_obs) -- so with padded observations the two pick different slots at the first
        # respawn: such a rollout is compared up to that step.
        omp = int(entries[f"gym/{case}/cfg"][2]); o = entries[f"gym/{case}/obs"]
        if omp > 0:
            resp = [t for t in range(1, len(o)) if ((o[t - 1][:, 69] == 1) & (o[t][:, 69] == 0)).any()]
            if resp:
                simlib.GYM_HORIZON[case] = simlib.GYM_HORIZON_PORT[case] = resp[0]; what += f" (padded observations: compared up to the first respawn, step {resp[0]})"
        try:
            if port_mode: TO.test_port_gym_vs_reference_golden(fx, port)
            else: T.test_hip_gym_rollouts_vs_reference_fixtures(fx)
            ok += 1; print(what + ": EXACT")
        except AssertionError as e:
            bad.append(case); print(what + ": DIFFERENT -- " + str(e)[:300])
    print(f"{ok} of {len(cases)} live Gym rollouts reproduced exactly on the {'host build' if port_mode else 'HIP path'}; different: {bad}")
    return 0 if not bad else 1


if __name__ == "__main__":
    sys.exit(main())
