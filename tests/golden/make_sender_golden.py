"""Generates tests/golden/sender_golden.json by running the REFERENCE's own receivers (RLGymPPO_CPP/python_scripts/render_receiver.py and
metric_receiver.py, imported from /root/reference in the build container) with their outputs captured:

  render   RenderSender::Send serialises {gamemode, state, actions} (RenderSender.cpp:22-92: PhysToJSON / PlayerToJSON / GameStateToJSON through
           nlohmann::json, whose objects are std::maps -- keys come out sorted) and hands the string to render_receiver.render_state, which
           re-shapes it (render_receiver.py:17-31) and sends json.dumps(...) to RocketSimVis over UDP.  Captured: the datagram's bytes for the
           GameState tests/cpp/host_api_check.cpp builds (the same values are written out below).
  metric   MetricSender (MetricSender.cpp:8-45) calls metric_receiver.init(py_exec, project, group, name, id) -> wandb.init(...) and
           add_metrics(dict) -> run.log(dict).  Captured: the wandb calls of a resumed run with one report.

The C++ senders (include/RLGymPPO_CPP/Util/{RenderSender,MetricSender}.h; the wandb side-car tools/metric_receiver.py) must reproduce these
byte for byte / call for call: tests/test_host_cpp.py.  Data only; run: python tests/golden/make_sender_golden.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/RLGymPPO_CPP/python_scripts"


def load(name, extra_modules=None):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    for k, v in (extra_modules or {}).items():
        sys.modules[k] = v
    spec.loader.exec_module(mod)
    return mod


def f32(x):
    return float(np.float32(x))


def vec(x, y, z):
    return [f32(x), f32(y), f32(z)]


def phys(pos, fwd, right, up, vel, ang):   # RenderSender.cpp:26-39
    return {"pos": vec(*pos), "forward": vec(*fwd), "right": vec(*right), "up": vec(*up), "vel": vec(*vel), "ang_vel": vec(*ang)}


def main():
    # ---- render: the GameState of tests/cpp/host_api_check.cpp ("RenderSender" block) --------------------------------------------------
    players = [
        {"car_id": 1, "team_num": 0, "phys": phys((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0), (0, 0, 0)),
         "boost_pickups": 0, "is_demoed": False, "on_ground": False, "ball_touched": False, "has_flip": False, "boost_amount": f32(0)},
        {"car_id": 2, "team_num": 1, "phys": phys((-100, 250.5, 17), (0.6, -0.8, 0), (0.8, 0.6, 0), (0, 0, 1), (1234.5678, -0.0001, 1e-5), (0.1, -5.5, 2.25)),
         "boost_pickups": 3, "is_demoed": True, "on_ground": True, "ball_touched": True, "has_flip": True, "boost_amount": f32(0.33)},
    ]
    pads = [False] * 34; pads[3] = True
    state = {"ball": phys((1, 2, 93.15), (1, 0, 0), (0, 1, 0), (0, 0, 1), (-2300.0, 1e16, 123456789.0), (0, 0, 6)), "players": players, "boost_pads": pads, "team_goals": [0, 0]}
    sent = []

    class FakeSock:
        def __init__(self, *a, **k): pass
        def sendto(self, data, addr): sent.append((data, addr))
    real_socket = __import__("socket")
    saved = real_socket.socket
    real_socket.socket = FakeSock
    try:
        rr = load("render_receiver")
        rr.render_state(json.dumps({"gamemode": "soccar", "state": state, "actions": [[0.0] * 8, [1.0] * 8]}, sort_keys=True, separators=(",", ":")))
    finally:
        real_socket.socket = saved
    assert len(sent) == 1 and sent[0][1] == ("127.0.0.1", 9273)
    datagram = sent[0][0].decode()

    # ---- metric: a resumed run with one report (the values of host_api_check.cpp's MetricSender block) -----------------------------------
    calls = []

    class FakeRun:
        id = "abcd1234"
        def log(self, d): calls.append(["log", {k: (repr(v) if isinstance(v, float) else v) for k, v in d.items()}])
    fake = types.ModuleType("wandb")
    fake.init = lambda **kw: (calls.append(["init", dict(kw)]), FakeRun())[1]
    mr = load("metric_receiver", {"wandb": fake})
    rid = mr.init(sys.executable, "proj", "grp", 'run "7"', "abcd1234")
    assert rid == "abcd1234"
    mr.add_metrics({"Policy Entropy": 4.25, "Cumulative Timesteps": 1234567.0, "bad": float("nan")})
    out = {"render": {"address": list(sent[0][1]), "datagram": datagram}, "metric": {"calls": calls}}
    path = os.path.join(HERE, "sender_golden.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path); print(datagram); print(calls)


if __name__ == "__main__":
    main()
