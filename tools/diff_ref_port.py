"""Tick-by-tick differential run: the REAL reference (oracle/_ref) against the host build of the stepper (oracle/_build), on the
scenarios of tests/golden/make_sim_golden.py.  Development tool (build container only; needs /root/reference built into oracle/_ref).

  python tools/diff_ref_port.py [scenario|all] [--resync] [--verbose]
Prints, per scenario, the first tick at which any body state leaves the tolerance and the error growth afterwards.
--resync copies the reference state into the port after every tick (isolates per-tick errors from accumulated drift).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from simlib import PortSim, RefSim  # noqa: E402
from rlgymppo_cpp_amd.state import ArenaState  # noqa: E402


def vec(s):
    out = {"ball.pos": list(s.ball.pos), "ball.vel": list(s.ball.vel), "ball.ang": list(s.ball.ang_vel)}
    for k in range(s.num_cars):
        c = s.cars[k]
        out[f"car{k}.pos"] = list(c.pos); out[f"car{k}.vel"] = list(c.vel); out[f"car{k}.ang"] = list(c.ang_vel)
        out[f"car{k}.rot"] = list(c.rot); out[f"car{k}.flags"] = [float(c.flags)]; out[f"car{k}.boost"] = [c.boost]
    return out


def main():
    import make_sim_golden as G
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "all"
    resync = "--resync" in sys.argv; verbose = "--verbose" in sys.argv
    port = PortSim(); verts, tris = port.procedural_mesh(); port.set_mesh(verts, tris)
    ref = RefSim(verts, tris)
    sc = G.scenarios()
    if hasattr(G, "extra_scenarios"):
        sc.update(G.extra_scenarios())
    for name, (s0, fn, ticks) in sc.items():
        if which != "all" and name != which:
            continue
        nc = s0.num_cars
        a = ref.arena(nc // 2)
        ref.set_state(a, s0)
        st = ref.get_state(a)
        first = None; worst = {}
        for t in range(ticks):
            for k in range(nc):
                c = fn(t, k)
                ref.set_controls(a, k, list(c)); st.cars[k].controls[:] = list(c)
            ref.step(a, 1)
            port.step(st, 1)
            r = ref.get_state(a)
            vr, vp = vec(r), vec(st)
            bad = []
            for key in vr:
                e = max(abs(x - y) for x, y in zip(vr[key], vp[key]))
                tol = 1e-3 if key.endswith(("pos", "vel")) else (0 if key.endswith("flags") else 1e-4)
                worst[key] = max(worst.get(key, 0), e)
                if e > tol:
                    bad.append((key, e))
            if bad and (first is None or verbose or resync):
                if first is None:
                    first = t
                print(f"  {name} tick {t}: " + ", ".join(f"{k} {e:.4g}" for k, e in bad[:8]))
            if resync:
                st = ArenaState.from_buffer_copy(bytes(r))
        print(f"{name}: ticks {ticks} first_bad {first} worst " + " ".join(f"{k}={v:.3g}" for k, v in worst.items() if v > 0 and not k.endswith("flags")))


if __name__ == "__main__":
    main()
