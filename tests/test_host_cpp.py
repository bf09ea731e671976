"""The C++ host layer (include/RLGymSim_CPP, include/RLGymPPO_CPP, rlgymppo_cpp_amd/host): a program written against the
reference's public API compiles against this repo's headers (CPU) and trains on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rlgymppo_cpp_amd")


def _run(cmd, **kw):
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)


def test_host_api_translation_and_utils(tmp_path):
    """CPU: plugin classes -> RlgpuGymConfig, loud failure for device-less plugins, Report / AvgTracker / Welford semantics."""
    lib = os.path.join(PKG, "librlgpu.so")
    assert os.path.exists(lib), "librlgpu.so missing: run __graft_entry__.build()"
    exe = str(tmp_path / "host_api_check")
    r = _run(["g++", "-std=c++20", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_api_check.cpp"), "-o", exe,
              "-L", PKG, "-lrlgymppo_amd", "-lrlgpu", f"-Wl,-rpath,{PKG}", "-Wl,-rpath-link,/opt/rocm/lib"])
    assert r.returncode == 0, r.stdout
    r = _run([exe], env=dict(os.environ, RLGPU_METRICS_DIR=str(tmp_path / "metrics")))
    assert r.returncode == 0 and "host api ok" in r.stdout, r.stdout
    # what the senders wrote is JSON as Python reads it, and the wandb side-car replays the metrics file
    import json
    mfile = [l for l in r.stdout.splitlines() if l.startswith("metrics file: ")][0][len("metrics file: "):]
    recs = [json.loads(l) for l in open(mfile)]
    assert recs[0]["_run"] == {"project": "proj", "group": "grp", "name": 'run "7"', "id": os.path.basename(mfile)[:-6]}
    assert recs[1]["Policy Entropy"] == 4.25 and recs[1]["Cumulative Timesteps"] == 1234567 and recs[1]["bad"] is None
    # F3: the datagram is, BYTE FOR BYTE, what the reference's own render_receiver.py sends for this GameState (tests/golden/sender_golden.json,
    # recorded by make_sender_golden.py from /root/reference/RLGymPPO_CPP/python_scripts/render_receiver.py with the socket captured)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "sender_golden.json")))
    got = [l for l in r.stdout.splitlines() if l.startswith("render datagram: ")][0][len("render datagram: "):]
    assert got == golden["render"]["datagram"], (got, golden["render"]["datagram"])
    dgram = json.loads(got)
    assert len(dgram["boost_pad_states"]) == 34 and dgram["boost_pad_states"][3] is True and dgram["cars"][1]["phys"]["pos"] == [-100, 250.5, 17]
    # ... and the wandb side-car makes, call for call, the calls the reference's metric_receiver.py makes for a resumed run with this report
    t = _run([sys.executable, os.path.join(ROOT, "tools", "metric_receiver.py"), mfile, "--once", "--record-calls"])
    assert t.returncode == 0, t.stdout
    calls = json.loads(t.stdout.strip().splitlines()[-1])
    want = golden["metric"]["calls"]
    rid = os.path.basename(mfile)[:-6]
    assert calls[0] == ["init", dict(want[0][1], id=rid)]                        # (the run id is the sender's: a new run's too, where the reference lets wandb choose)
    assert calls[1] == want[1] and calls[2] == want[1] and len(calls) == 3       # the report, NaN included, as MetricSender.cpp:31-44 hands it over


def test_portable_libm_is_the_c_librarys_bit_for_bit(tmp_path):
    """CPU: csrc/rl_libm.h (the sinf / cosf / atan2f / atanf / asinf the physics calls on the device AND on the host) against glibc over
    20 million arguments per function, and the two powf constants of the tick: identical everywhere (tests/cpp/libm_check.cpp; the same
    program over 4e8 arguments per function was clean too).  With these the HIP stepper and the host build compute the same bits
    (tests/test_gpu_parity.py::test_physics_ticks_match_host_port_on_golden_scenarios compares them for equality)."""
    exe = str(tmp_path / "libm_check")
    r = _run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "libm_check.cpp"), "-o", exe, "-lm"])
    assert r.returncode == 0, r.stdout
    r = _run([exe, "20"])
    assert r.returncode == 0 and "mismatches sinf 0 cosf 0 atan2f 0 atanf 0 asinf 0" in r.stdout, r.stdout


def test_restated_bullet_math_is_the_references_bit_for_bit(tmp_path):
    """CPU, build container (needs the reference's headers): csrc/rl_math.h against Bullet's own inline functions compiled the way the
    reference compiles them (SSE branches) -- btVector3::normalize (rsqrtss + Newton step, emulated), quaternion construction and
    products, quatRotate, setRotation, getRotation, matrix * vector, safeNormalize: identical on 300 000 random arguments each."""
    inc = "/root/reference/RLGymPPO_CPP/RLGymSim_CPP/RocketSim/libsrc"
    if not os.path.isdir(inc):
        pytest.skip("/root/reference is not mounted here")
    exe = str(tmp_path / "bullet_math_check")
    r = _run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-w", "-I", inc, os.path.join(ROOT, "tests", "cpp", "bullet_math_check.cpp"), "-o", exe, "-lm"])
    assert r.returncode == 0, r.stdout
    r = _run([exe])
    assert r.returncode == 0 and "mismatches normalize 0," in r.stdout, r.stdout


def test_example_program_is_built():
    """CPU: build() produced the host library and the example program written against the reference's API."""
    for f in ("librlgymppo_amd.so", "example_main", "infer_unit_check", "bench_main", "plugin_fallback_check"):
        assert os.path.exists(os.path.join(PKG, f)), f


@pytest.mark.gpu
def test_example_program_trains_and_checkpoints(tmp_path):
    """GPU: two iterations of the example program (step + iteration callbacks, slow GameState path), a checkpoint, and a resume."""
    exe = os.path.join(PKG, "example_main")
    ck = str(tmp_path / "ck")
    lock = dict(os.environ, RLGPU_LOCKSTEP_COLLECTION="1")     # this test counts timesteps exactly: every game makes the same number of steps per iteration
    r = _run([exe, "2", "4", "16", "4096", ck], cwd=str(tmp_path), timeout=600, env=lock)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("ITERATION COMPLETED") == 2 and "player_speed" in r.stdout and "Policy Entropy" in r.stdout
    first_run_out = r.stdout
    saved = sorted(int(d) for d in os.listdir(ck))
    assert saved and saved[-1] == 2 * 4096, saved
    for f in ("RUNNING_STATS.json", "PPO_POLICY.lt", "PPO_CRITIC.lt", "PPO_POLICY_OPTIM.lt", "PPO_CRITIC_OPTIM.lt"):
        assert os.path.getsize(os.path.join(ck, str(saved[-1]), f)) > 0
    # the payloads are the reference's own TorchScript archives (PPOLearner.cpp:408-411): torch reads them
    import torch
    pol = torch.jit.load(os.path.join(ck, str(saved[-1]), "PPO_POLICY.lt"))
    assert [tuple(p.shape) for _, p in pol.named_parameters()] == [(256, 89), (256,), (256, 256), (256,), (256, 256), (256,), (90, 256), (90,)]
    assert [k for k, _ in pol.named_parameters()][-1] == "6.bias"
    # the Python host reads the C++ host's checkpoint (same layout and payload)
    sys.path.insert(0, ROOT)
    from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
    cfg = LearnerConfig(numEnvs=64, teamSize=1, timestepsPerIteration=4096, checkpointLoadFolder=ck, checkpointSaveFolder=ck,
                        ppo=PPOLearnerConfig(batchSize=4096, miniBatchSize=4096, epochs=1, autocastLearn=True))
    L = Learner(cfg)
    assert L.load() and L.total_timesteps == 2 * 4096
    # and the C++ host resumes from it
    # ... this time with the reference's step callback (every game's GameState on the host each step) instead of the device-side metrics
    r = _run([exe, "1", "4", "16", "4096", ck], cwd=str(tmp_path), timeout=600, env=dict(lock, EXAMPLE_STEP_CALLBACK="1"))
    assert r.returncode == 0 and "loaded checkpoint" in r.stdout and str(3 * 4096) in os.listdir(ck), r.stdout[-3000:]
    def metric(out, name):
        return float([l for l in out.splitlines() if l.strip().startswith("[metric] " + name)][-1].split(":")[-1].replace(",", ""))
    assert 100 < metric(r.stdout, "player_speed") < 2300 and 0 <= metric(r.stdout, "in_air_ratio") <= 1
    assert 100 < metric(first_run_out, "player_speed") < 2300 and 0 <= metric(first_run_out, "ball_touch_ratio") <= 1   # LearnerConfig::deviceStepMetrics
    # metrics: one JSON-lines file for the run; the resumed process continued it under the run id stored in RUNNING_STATS.json
    import json
    mdir = tmp_path / "metrics" / "rlgymppo-cpp"
    files = os.listdir(mdir)
    assert len(files) == 1, files
    recs = [json.loads(l) for l in open(mdir / files[0])]
    assert "_run" in recs[0] and len(recs) == 1 + 3
    assert [int(x["Cumulative Timesteps"]) for x in recs[1:]] == [4096, 8192, 12288] and "player_speed" in recs[1] and "Policy Entropy" in recs[3]
    st2, st3 = (json.load(open(os.path.join(ck, str(k * 4096), "RUNNING_STATS.json"))) for k in (2, 3))
    assert st3["run_id"] == files[0][:-6]
    # the resumed run goes on in the action sampler's stream where the first one stopped, and moves the env batch to a fresh RNG epoch
    assert st2["sampler_calls"] > 0 and st3["sampler_calls"] > st2["sampler_calls"] and st2["env_stream_epoch"] == 0 and st3["env_stream_epoch"] == 1
    # render mode: the newest checkpoint plays one game, every step goes to RocketSimVis' UDP port as a JSON datagram
    import socket
    before_render = sorted(os.listdir(ck))
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.bind(("127.0.0.1", 9273))
    rx.settimeout(5)
    r = _run([exe, "12", "4", "16", "4096", ck, "render"], cwd=str(tmp_path), timeout=600, env=dict(lock, RLGPU_RENDER_NO_SLEEP="1"))
    assert r.returncode == 0 and "Render mode is enabled" in r.stdout and "loaded checkpoint" in r.stdout, r.stdout[-3000:]
    grams = []
    for _ in range(12):
        grams.append(json.loads(rx.recv(65536)))
    rx.close()
    assert all(len(g["cars"]) == 2 and len(g["boost_pad_states"]) == 34 and g["gamemode"] == "soccar" for g in grams)
    assert grams[0]["cars"][0]["phys"]["pos"] != grams[-1]["cars"][0]["phys"]["pos"]          # the policy is driving
    assert sorted(os.listdir(ck)) == before_render                                               # render mode saves nothing


@pytest.mark.gpu
def test_user_plugins_run_on_the_host_and_standalone_gym(tmp_path):
    """GPU: tests/cpp/plugin_fallback_check.cpp -- user RewardFunction / OBSBuilder / TerminalCondition / StateSetter / ActionParser subclasses
    compiled against include/ train through the Learner (their kinds on the host, the arenas on the device), the built-ins' host forms
    reproduce the device path's experience, and a standalone Gym resets / steps its Arena facade."""
    r = _run([os.path.join(PKG, "plugin_fallback_check")], cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0 and "plugin fallback ok" in r.stdout, r.stdout[-4000:]


def test_reference_example_source_compiles_unchanged():
    """CPU, this container only: the reference's own examplemain.cpp, byte for byte, compiles against include/ (its
    `#include "RLBotClient.h"` resolves to this repo's header because the source is fed through stdin)."""
    src = "/root/reference/examplemain.cpp"
    if not os.path.exists(src):
        pytest.skip("/root/reference is not mounted here")
    with open(src, "rb") as f:
        r = subprocess.run(["g++", "-std=c++20", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c++", "-"], stdin=f,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    # ... and LINKS, unchanged, against the host library (north star: "examplemain.cpp links unchanged"); the program is not run here (no GPU)
    import tempfile
    with tempfile.TemporaryDirectory() as d, open(src, "rb") as f:
        exe = os.path.join(d, "examplemain")
        r = subprocess.run(["g++", "-std=c++20", "-O1", "-I", os.path.join(ROOT, "include"), "-x", "c++", "-", "-o", exe, "-L", PKG, "-lrlgymppo_amd", "-lrlgpu",
                            "-Wl,-rpath," + PKG, "-Wl,-rpath-link,/opt/rocm/lib"], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0 and os.path.exists(exe), r.stdout[-3000:]


@pytest.mark.gpu
def test_infer_unit_and_host_obs_builders(tmp_path):
    """GPU: host DefaultOBS rows == device rows bit for bit (1v1, 2v2, 3v3); InferUnit on .lt archives vs the numpy oracle."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import learner_ref as R
    from rlgymppo_cpp_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(3)

    def net(dims):
        shapes = [((dims[i + 1], dims[i]), (dims[i + 1],)) for i in range(len(dims) - 1)]
        flat = np.concatenate([rs.uniform(-1, 1, ws[0] * ws[1] + bs[0]).astype(np.float32) / np.sqrt(ws[1]) for ws, bs in shapes])
        return np.ascontiguousarray(flat, np.float32), shapes

    pol, pol_shapes = net([89, 32, 24, 90]); cri, cri_shapes = net([89, 16, 1])
    pp, cp, out = str(tmp_path / "PPO_POLICY.lt"), str(tmp_path / "PPO_CRITIC.lt"), str(tmp_path / "out.bin")
    for path, flat, dims in ((pp, pol, [89, 32, 24, 90]), (cp, cri, [89, 16, 1])):
        d = np.array(dims, np.int32)
        assert lib.rlgpu_lt_write_model(path.encode(), d.ctypes.data, len(dims) - 1, flat.ctypes.data) == 0
    r = _run([os.path.join(PKG, "infer_unit_check"), pp, cp, out], timeout=600)
    assert r.returncode == 0 and "infer unit ok" in r.stdout and "skill tracker ok" in r.stdout, r.stdout[-3000:]
    raw = np.fromfile(out, np.uint8)
    n = int(raw[:4].view(np.int32)[0])
    rec = raw[4:].view(np.float32).reshape(n, 89 + 90 + 90 + 8 + 1)
    table = np.zeros((128, 8), np.float32)
    n_act = lib.rlgpu_action_table(table.ctypes.data, 128)
    assert n_act == 90 and n >= 8
    for row in rec:
        obs, probs, probs_t, act, val = row[:89], row[89:179], row[179:269], row[269:277], row[277]
        logits = R.mlp_forward(pol, pol_shapes, obs[None])[0][0]
        want = R.policy_probs(logits[None], 1.0)[0]
        assert np.abs(probs - want).max() < 1e-5                                   # fp32 MFMA-free inference path vs numpy
        assert np.abs(probs_t - R.policy_probs(logits[None], 2.5)[0]).max() < 1e-5
        assert (act == table[int(np.argmax(want))]).all()
        v = R.mlp_forward(cri, cri_shapes, obs[None])[0].reshape(-1)[0]
        assert abs(val - v) < 1e-4 * max(1.0, abs(v))


def _skill_script_text():
    """tests/golden/skill_golden.json (recorded from the real reference skill tracker by tests/golden/make_skill_golden.py) as the flat text
    infer_unit_check --skill-script reads."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "skill_golden.json")) as f:
        gold = json.load(f)
    e = gold["elo"]
    lines = ["ELO %d %d %r" % (len(e["start_bits"]), len(e["winner"]), e["rating_inc"]), " ".join(map(str, e["start_bits"]))]
    for w, l, fl, tr in zip(e["winner"], e["loser"], e["flags"], e["trace_bits"]):
        lines.append("%d %d %d %s" % (w, l, fl, " ".join(map(str, tr))))
    for s in gold["idle"] + gold["goals"]:
        lines.append("RUN %r %d %d %d %d %d %r %r" % (s["goal_sign"], len(s["deltas"]), s["update_interval"], s["timesteps_per_version"], s["max_versions"],
                                                      int(s["start_with_version"]), s["rating_inc"], s["sim_time"]))
        rows = np.array(s["rows_bits"], np.uint32)
        vals = rows.view(np.float32)
        for d, rb, rv in zip(s["deltas"], rows, vals):
            lines.append("%d %d %d %d %d %d %d %s" % (d, int(rv[0]), int(rv[1]), int(rv[2]), int(rv[3]), int(rv[4]), rb[5], " ".join(map(str, rb[6:]))))
    return "\n".join(lines) + "\n", gold


def test_skill_golden_is_what_the_reference_skill_tracker_returns():
    """CPU: the committed recordings are the real reference's (oracle/_ref/libref_skill.so, when it is there): UpdateRatings bit for bit."""
    import ctypes as C
    text, gold = _skill_script_text()
    assert text.count("RUN") == 5 and len(gold["elo"]["winner"]) == 400
    so = os.path.join(ROOT, "oracle", "_ref", "libref_skill.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libref_skill.so not built (make -C oracle ref_skill needs /root/reference)")
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import simlib
    lib = C.CDLL(so)
    verts, tris = simlib.PortSim().procedural_mesh()
    root = tempfile.mkdtemp(prefix="skill_mesh_")
    simlib.write_cmf_parts(verts, tris, [len(tris)], root)
    assert lib.refs_init_dir(root.encode()) == 0
    e = gold["elo"]
    ratings = np.array(e["start_bits"], np.uint32).view(np.float32).copy()
    w, l, fl = (np.array(e[k], np.int32) for k in ("winner", "loser", "flags"))
    trace = np.zeros((len(w), len(ratings)), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.refs_elo_script(len(ratings), p(ratings), len(w), p(w), p(l), p(fl), C.c_float(e["rating_inc"]), p(trace)) == 0
    assert (trace.view(np.uint32) == np.array(e["trace_bits"], np.uint32)).all()


@pytest.mark.gpu
def test_skill_tracker_against_reference_recordings(tmp_path):
    """GPU: the product's skill tracker replays the scripts recorded from the real reference one (tests/golden/skill_golden.json): the Elo
    arithmetic bit for bit, RunGames' version bookkeeping call by call, and -- with a user state setter that starts every episode behind a
    goal line -- the goal test on the step's GameState and who gets the points (SkillTracker.cpp:72-86, 104-146, 152-257)."""
    text, _ = _skill_script_text()
    path = str(tmp_path / "skill_script.txt")
    with open(path, "w") as f:
        f.write(text)
    r = _run([os.path.join(PKG, "infer_unit_check"), "--skill-script", path], timeout=600)
    assert r.returncode == 0 and "skill tracker vs the reference's recordings ok" in r.stdout, r.stdout[-3000:]


@pytest.mark.gpu
def test_example_program_default_collection_is_free_running(tmp_path):
    """GPU: the example program as it comes (LearnerConfig::lockstepCollection = false): every game steps at its own pace until the batch has
    timestepsPerIteration together (ThreadAgentManager.cpp:16-82), so an iteration holds between 4096 and 4096 + one step of each of the 64
    games' two players; the run trains, reports and checkpoints under the exact count."""
    import json
    exe = os.path.join(PKG, "example_main")
    ck = str(tmp_path / "ck")
    r = _run([exe, "3", "4", "16", "4096", ck], cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "free-running collection" in r.stdout and r.stdout.count("ITERATION COMPLETED") == 3 and "Policy Entropy" in r.stdout
    saved = sorted(int(d) for d in os.listdir(ck))
    assert saved and 3 * 4096 <= saved[-1] <= 3 * (4096 + 128), saved
    mdir = tmp_path / "metrics" / "rlgymppo-cpp"
    recs = [json.loads(l) for l in open(mdir / os.listdir(mdir)[0])][1:]
    ts = [int(x["Cumulative Timesteps"]) for x in recs]
    assert len(ts) == 3 and all(4096 <= b - a <= 4096 + 128 for a, b in zip([0] + ts, ts)) and ts[-1] == saved[-1]
    assert all(np.isfinite(x["Policy Entropy"]) and np.isfinite(x["Value Function Loss"]) for x in recs)


@pytest.mark.gpu
def test_example_program_with_skill_tracker(tmp_path):
    """GPU: ELO evaluation against stored versions while training; ratings land in the report, the metrics file and RUNNING_STATS.json;
    a resumed run rebuilds its old versions from the older checkpoints (Learner.cpp:311-370)."""
    import json
    exe = os.path.join(PKG, "example_main")
    ck = str(tmp_path / "ck")
    env = dict(os.environ, EXAMPLE_SKILL_TRACKER="1", RLGPU_LOCKSTEP_COLLECTION="1")     # (checkpoint folders are named by exact timestep counts below)
    r = _run([exe, "3", "4", "16", "4096", ck], cwd=str(tmp_path), timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("Running skill eval game(s)...") == 3 and r.stdout.count("New ratings:") == 3 and "Skill Rating 1v1" in r.stdout
    saved = sorted(int(d) for d in os.listdir(ck))
    assert saved == [4096, 8192, 12288]
    stats = json.load(open(os.path.join(ck, "12288", "RUNNING_STATS.json")))
    assert set(stats["skill_rating"]) == {"1v1"} and 900 < stats["skill_rating"]["1v1"] < 1100
    mdir = tmp_path / "metrics" / "rlgymppo-cpp"
    recs = [json.loads(l) for l in open(mdir / os.listdir(mdir)[0])][1:]
    assert abs(recs[-1]["Skill Rating 1v1"] - stats["skill_rating"]["1v1"]) < 1e-3
    r = _run([exe, "1", "4", "16", "4096", ck], cwd=str(tmp_path), timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "Attempting to load 2 old versions" in r.stdout and "[0]: Found at 8192" in r.stdout and "[1]: Found at 4096" in r.stdout, r.stdout[-4000:]


@pytest.mark.gpu
@pytest.mark.timeout(240)
def test_two_rank_launch_on_one_gpu_ends_instead_of_hanging(tmp_path):
    """De-risking the first real N > 1 run (only the driver has 8 GPUs): two bench_main processes as RANK 0 / 1 of WORLD_SIZE 2, both on the
    ONE visible GPU, through the same code a torchrun launch takes -- rlgpu_comm_init_env: rank 0 writes the rendezvous file (private
    directory, O_EXCL, magic + time stamp), rank 1 reads it, both enter ncclCommInitRank.  RCCL normally refuses two ranks on one device;
    the assertion is on the FAILURE MODE: the error comes back through rlgpu_comm_last_error into the Learner's exception, both processes
    end non-zero within the time limit instead of hanging, and the rendezvous file is gone.  (Should this RCCL accept the pair, the run
    must then complete as a 2-rank run: rccl_ranks 2 and two per-rank clocks in rank 0's line.)  Every line of the multi-rank host path
    except the collective on >= 2 devices has then executed before the driver's 8-GPU bench.  Fresh child processes only."""
    import json, socket, time
    exe = os.path.join(PKG, "bench_main")
    assert os.path.exists(exe)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    rdv = str(tmp_path / "rdv"); os.mkdir(rdv, 0o700)
    base = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", RLGPU_COMM_DIR=rdv, RLGPU_COMM_TAG="t%d" % os.getpid(),
                RLGPU_COMM_TIMEOUT_S="45", HSA_ENABLE_IPC_MODE_LEGACY="0", RLGPU_QUIET="1")
    cmd = [exe, "--envs", "64", "--horizon", "4", "--steps", "2", "--warmup", "1"]
    t0 = time.time()
    procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=150))
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            pytest.fail("a rank of the two-rank launch hung")
    took = time.time() - t0
    codes = [p.returncode for p in procs]
    assert not os.listdir(rdv), f"rendezvous file left behind: {os.listdir(rdv)}"
    if codes == [0, 0]:      # this RCCL build lets two ranks share a device: then it was a real 2-rank run
        line = json.loads(outs[0][0].strip().splitlines()[-1])
        assert line["rccl_ranks"] == 2 and len(line["rank_ms_per_step"]) == 2 and line["n_gpus"] == 2 and line["allreduce_calls"] > 0, line
    else:
        assert all(c != 0 for c in codes), (codes, outs)
        assert took < 140
        assert any("ncclCommInitRank" in o[1] or "rlgpu_comm_init_env" in o[1] for o in outs), outs


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_two_ranks_on_one_gpu_run_the_whole_multi_gpu_path_and_fail_fast(tmp_path):
    """VERDICT r03 item 5.  RCCL refuses two ranks on one device and this pool's boxes have one GPU, so the hosts' N > 1 path had never executed
    end to end.  RLGPU_COMM_TRANSPORT=shm puts the three collectives of include/rlgpu.h on a host-staged shared-memory transport (sums in
    rank order); everything above it is the product's: two bench_main ranks (C++ Learner) shard the envs (seed + 1000 rank), explore with
    rank-keyed samplers, take rank 0's parameters / Adam state at construction (rlgpu_learner_sync_from_rank0), all-reduce the flat gradient
    once per optimizer step, scale inside the clip, share rank 0's returns for the Welford statistic, and check a parameter checksum against
    rank 0's every iteration (RLGPU_REPLICA_CHECK_EVERY=1).
      * both ranks finish 3 iterations with EQUAL parameter checksums although their experience differs (the replicas stay replicas);
      * the result differs from a single-rank run's (the other rank's gradients did enter);
      * a rank that is killed ends the other one within the communicator's timeout, non-zero, with the reason in its output."""
    import json, re, signal, socket, time
    exe = os.path.join(PKG, "bench_main")

    def launch(steps, tag, timeout_s="60"):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        rdv = str(tmp_path / ("rdv_" + tag)); os.mkdir(rdv, 0o700)
        base = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", RLGPU_COMM_DIR=rdv, RLGPU_COMM_TAG=tag,
                    RLGPU_COMM_TRANSPORT="shm", RLGPU_COMM_TIMEOUT_S=timeout_s, RLGPU_REPLICA_CHECK_EVERY="1", RLGPU_LOCKSTEP_COLLECTION="1", RLGPU_QUIET="1")
        cmd = [exe, "--envs", "256", "--horizon", "8", "--steps", str(steps), "--warmup", "0"]
        return [subprocess.Popen(cmd, env=dict(base, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(2)]

    def finish(procs):
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=200))
            except subprocess.TimeoutExpired:
                for q in procs: q.kill()
                pytest.fail("a rank hung")
        return outs

    sums = []
    for attempt in range(1):
        procs = launch(3, "a%d" % attempt)
        outs = finish(procs)
        assert [p.returncode for p in procs] == [0, 0], outs
        cs = [re.search(r"parameter checksum ([0-9a-f]{16})", o[1]).group(1) for o in outs]
        assert cs[0] == cs[1], cs
        line = json.loads(outs[0][0].strip().splitlines()[-1])
        assert line["rccl_ranks"] == 2 and line["n_gpus"] == 2 and len(line["rank_ms_per_step"]) == 2 and line["allreduce_calls"] == 3, line
        sums.append(cs[0])
    # (since round 5 the fused minibatch kernels sum their row slabs in a fixed order -- per-slab partials + an ordered reduce, csrc/ppo_fused.h -- so
    # two launches of the same flow do end in the same bits; what this test holds is that BOTH ranks do, and that the other rank's gradients entered)
    solo = subprocess.run([exe, "--envs", "256", "--horizon", "8", "--steps", "3", "--warmup", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT,
                          env=dict(os.environ, RLGPU_LOCKSTEP_COLLECTION="1", RLGPU_QUIET="1"))
    assert solo.returncode == 0
    assert re.search(r"parameter checksum ([0-9a-f]{16})", solo.stderr).group(1) != sums[0]       # the other rank's gradients did enter
    # fail-fast: kill rank 1 in the middle of a long run
    procs = launch(1000000, "kill", timeout_s="8")
    time.sleep(6)
    assert procs[0].poll() is None and procs[1].poll() is None, "the long run ended by itself"
    procs[1].send_signal(signal.SIGKILL)
    t0 = time.time()
    try:
        out0 = procs[0].communicate(timeout=60)
    except subprocess.TimeoutExpired:
        procs[0].kill()
        pytest.fail("rank 0 went on waiting for a dead peer")
    assert procs[0].returncode != 0 and time.time() - t0 < 40
    assert "peer" in out0[1] or "peers" in out0[1], out0[1][-1500:]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_deterministic_gradient_mode_makes_a_run_a_function_of_its_seed(tmp_path):
    """VERDICT r04 item 3 (north star: "a fixed-seed run matches ... bit-exact").  The reference's gradients are sums in a fixed order (one stream,
    PPOLearner.cpp:205-215); this build's default adds the dW / db partials of a minibatch's row slabs with fp32 atomics in the order the slabs
    finish, so two launches of one seed part after the first optimizer step.  LearnerConfig::deterministicGradients (bench_main --deterministic):
    per-slab partial gradients summed in slab order + lockstep collection.
      * two single-rank runs of 50 iterations end with the SAME parameter checksum;
      * two two-rank runs (shm transport on the one GPU: sums in rank order) end with the same checksum as each other, on both ranks;
      * the deterministic single-rank result differs from the two-rank one (the other rank's gradients entered)."""
    import re, socket
    exe = os.path.join(PKG, "bench_main")
    args = ["--envs", "512", "--horizon", "16", "--steps", "50", "--warmup", "0", "--deterministic"]

    def solo():
        r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=dict(os.environ, RLGPU_QUIET="1"), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return re.search(r"parameter checksum ([0-9a-f]{16})", r.stderr).group(1)

    def pair(tag):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        rdv = str(tmp_path / ("rdv_" + tag)); os.mkdir(rdv, 0o700)
        base = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", RLGPU_COMM_DIR=rdv, RLGPU_COMM_TAG=tag,
                    RLGPU_COMM_TRANSPORT="shm", RLGPU_COMM_TIMEOUT_S="60", RLGPU_REPLICA_CHECK_EVERY="1", RLGPU_QUIET="1")
        procs = [subprocess.Popen([exe] + args, env=dict(base, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(2)]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=300))
            except subprocess.TimeoutExpired:
                for q in procs: q.kill()
                pytest.fail("a rank hung")
        assert [p.returncode for p in procs] == [0, 0], outs
        cs = [re.search(r"parameter checksum ([0-9a-f]{16})", o[1]).group(1) for o in outs]
        assert cs[0] == cs[1], cs
        return cs[0]

    a, b = solo(), solo()
    assert a == b, f"two deterministic runs of one seed ended with different parameters: {a} vs {b}"
    c, d = pair("d1"), pair("d2")
    assert c == d, f"two deterministic two-rank runs ended with different parameters: {c} vs {d}"
    assert a != c


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_the_drivers_multi_gpu_bench_command_on_one_gpu(tmp_path):
    """the command the driver will run on an 8-GPU node -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- executed end to end with N = 2 on the one GPU of the box:
    the launcher starts two bench.py ranks (fresh processes, the launcher itself never touches the GPU), each starts its bench_main, the two
    meet through the rendezvous file named after the launcher, shard the envs, all-reduce per optimizer step over the shared-memory TEST
    transport (RLGPU_COMM_TRANSPORT=shm, RLGPU_SHM_DEVICE=0: RCCL refuses two ranks on a device) and rank 0 prints ONE JSON line.  Everything of
    the driver's future command runs except the RCCL ring itself.  The line says which transport it was; without --allow-test-transport
    bench.py refuses to print a multi-rank line that did not go over RCCL."""
    import json, socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    rdv = str(tmp_path / "rdv_torchrun"); os.mkdir(rdv, 0o700)
    env = dict(os.environ, RLGPU_COMM_TRANSPORT="shm", RLGPU_SHM_DEVICE="0", RLGPU_COMM_DIR=rdv, RLGPU_COMM_TIMEOUT_S="60", RLGPU_QUIET="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--envs", "512", "--no-cpu-baseline", "--mesh", "procedural", "--trained-warmup", "0", "--learned-warmup", "0"]
    r = subprocess.run(base + ["--allow-test-transport"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and len(line["rank_ms_per_step"]) == 2 and line["transport"] == "shm", line
    assert line["allreduce_calls"] == 3 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert any(kv.startswith("RLGPU_COMM_TRANSPORT=shm") for kv in line["env_overrides"]), line["env_overrides"]
    # value = the agent-steps of BOTH ranks over the slowest rank's time: two shards of 512 envs gather at least 2 x 512 x 2 x 32 per iteration
    per_iter = line["value"] * line["ms_per_step"] * 1e-3
    assert per_iter >= 2 * 512 * 2 * 32 * 0.999, (per_iter, line["value"], line["ms_per_step"])
    assert abs(line["config"]["agent_steps_per_iter"] * 2 - per_iter) < 1e-3 * per_iter
    assert not os.listdir(rdv), os.listdir(rdv)
    # the same launch without the test flag: no line, a reason
    r2 = subprocess.run(base, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env, timeout=300)
    assert r2.returncode != 0 and not [ln for ln in r2.stdout.splitlines() if ln.startswith("{")]
    assert "not a benchmark line" in r2.stderr, r2.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_eight_ranks_on_one_gpu_the_drivers_command_equal_replicas_and_a_lost_rank(tmp_path):
    """VERDICT r05 "next" 7: WORLD = 8 before the driver's first 8-GPU run, on the one GPU of the box (shm test transport, 256 envs per rank, fresh child
    processes only).  (a) The driver's own command -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8` -- prints ONE JSON
    line with n_gpus 8 and eight per-rank times.  (b) Eight bench_main ranks started directly: equal parameter checksums on all of them after three optimizer
    steps, eight distinct sampler streams and env seeds.  (c) Rank 7 killed in the middle of a long run ends the other seven, non-zero, within the
    communicator's timeout.  That is every line of `bench.py --gpus 8` except the RCCL ring itself."""
    import json, re, signal, socket, time
    exe = os.path.join(PKG, "bench_main")
    N = 8

    def free_port():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close(); return port

    # (a) the driver's command
    rdv = str(tmp_path / "rdv_torchrun8"); os.mkdir(rdv, 0o700)
    env = dict(os.environ, RLGPU_COMM_TRANSPORT="shm", RLGPU_SHM_DEVICE="0", RLGPU_COMM_DIR=rdv, RLGPU_COMM_TIMEOUT_S="120", RLGPU_QUIET="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(N), "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(N), "--steps", "3", "--warmup", "1", "--envs", "256", "--no-cpu-baseline", "--mesh", "procedural", "--trained-warmup", "0",
           "--learned-warmup", "0", "--allow-test-transport"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env, timeout=400)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == N and line["rccl_ranks"] == N and len(line["rank_ms_per_step"]) == N and line["transport"] == "shm" and line["allreduce_calls"] == 3, line
    per_iter = line["value"] * line["ms_per_step"] * 1e-3
    assert per_iter >= N * 256 * 2 * 32 * 0.999, (per_iter, line["value"], line["ms_per_step"])

    # (b) + (c): eight ranks started directly, so that one of them can be killed by its own handle
    def launch(steps, tag, timeout_s="120"):
        rdv2 = str(tmp_path / ("rdv_" + tag)); os.mkdir(rdv2, 0o700)
        base = dict(os.environ, WORLD_SIZE=str(N), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), LOCAL_RANK="0", RLGPU_COMM_DIR=rdv2, RLGPU_COMM_TAG=tag,
                    RLGPU_COMM_TRANSPORT="shm", RLGPU_COMM_TIMEOUT_S=timeout_s, RLGPU_REPLICA_CHECK_EVERY="1", RLGPU_LOCKSTEP_COLLECTION="1", RLGPU_QUIET="1")
        c = [exe, "--envs", "256", "--horizon", "8", "--steps", str(steps), "--warmup", "0"]
        return [subprocess.Popen(c, env=dict(base, RANK=str(k)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for k in range(N)]

    procs = launch(3, "eq8")
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300))
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            pytest.fail("a rank hung")
    assert [p.returncode for p in procs] == [0] * N, [o[1][-400:] for o in outs]
    found = [re.search(r"parameter checksum ([0-9a-f]{16}), sampler stream (\d+), env seed (\d+)", o[1]) for o in outs]
    assert all(found), [o[1][-300:] for o in outs]
    assert len({m.group(1) for m in found}) == 1, [m.group(1) for m in found]             # the replicas stayed replicas
    assert sorted(int(m.group(2)) for m in found) == list(range(N))                       # eight sampler streams
    assert len({int(m.group(3)) for m in found}) == N                                     # eight env seeds
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == N and len(line["rank_ms_per_step"]) == N and line["allreduce_calls"] == 3, line

    procs = launch(1000000, "kill8", timeout_s="10")
    time.sleep(10)
    assert all(p.poll() is None for p in procs), "the long run ended by itself"
    procs[N - 1].send_signal(signal.SIGKILL)
    t0 = time.time()
    for k in range(N - 1):
        try:
            out = procs[k].communicate(timeout=90)
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            pytest.fail(f"rank {k} went on waiting for a dead peer")
        assert procs[k].returncode != 0, k
    assert time.time() - t0 < 80
    procs[N - 1].wait()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_free_running_ranks_count_timesteps_together_and_stop_together(tmp_path):
    """with free-running collection (the default) every rank gathers its own number of timesteps per iteration -- they differ
    by up to one step of every game -- so the job's count must be taken collectively: Learner::Learn's exit test (`totalTimesteps < timestepLimit`),
    the checkpoint cadence and the folder names hang on it, and a rank that leaves the loop one iteration before its peer leaves the peer waiting in
    the next all-reduce.  Two example_main ranks on the one GPU (shm transport), NO RLGPU_LOCKSTEP_COLLECTION, a timestep limit that falls
    inside an iteration: both end by themselves, after the same number of iterations, with the same cumulative count, which is not a multiple of
    the lockstep batch."""
    import re, socket
    exe = os.path.join(PKG, "example_main")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    rdv = str(tmp_path / "rdv_free"); os.mkdir(rdv, 0o700)
    B = 16384
    base = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", RLGPU_COMM_DIR=rdv, RLGPU_COMM_TAG="free",
                RLGPU_COMM_TRANSPORT="shm", RLGPU_COMM_TIMEOUT_S="30", RLGPU_REPLICA_CHECK_EVERY="1", RLGPU_QUIET="1", EXAMPLE_TIMESTEP_LIMIT=str(2 * B * 4 + 2 * B // 2))
    base.pop("RLGPU_LOCKSTEP_COLLECTION", None)
    procs = [subprocess.Popen([exe, "1000000", "1", "256", str(B)], env=dict(base, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(tmp_path)) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=200))
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            pytest.fail("a rank went on waiting for its peer: the ranks' timestep counts parted")
    assert [p.returncode for p in procs] == [0, 0], outs
    got = [re.search(r"\[example_main rank (\d)/2\] (\d+) iterations, (\d+) timesteps", o[1]) for o in outs]
    assert all(got), outs
    assert got[0].group(2) == got[1].group(2) and got[0].group(3) == got[1].group(3), [g.group(0) for g in got]
    iters, total = int(got[0].group(2)), int(got[0].group(3))
    assert iters == 5 and total >= 2 * B * 5 and total % B != 0, (iters, total)      # free-running: B .. B + one step of every game per rank and iteration


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--envs", "1024"], ["--envs", "1024", "--lockstep"], ["--envs", "700", "--overlap", "--fp16"],
                                  ["--envs", "600", "--team-size", "2", "--padded-zero-sum"], ["--envs", "300", "--team-size", "3", "--padded-zero-sum", "--epochs", "2"]])
def test_cpp_host_runs_clean_under_redzones(args):
    """RLGPU_REDZONE: guard bytes behind every device buffer of the C++ host library (experience FIFO staging, GAE / statistics scratch, index lists),
    of the env batch and of the learner; the Learner's destructor checks all three after a run of the whole training loop -- free-running and lockstep
    collection, collect-during-learn with fp16 operands, 2v2 / 3v3 with padded observations, two epochs -- and ends the process with the name of an
    overwritten buffer otherwise."""
    import json
    exe = os.path.join(PKG, "bench_main")
    r = subprocess.run([exe, "--horizon", "16", "--steps", "6", "--warmup", "1"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT,
                       env=dict(os.environ, RLGPU_REDZONE="65536", RLGPU_QUIET="1"), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RLGPU_REDZONE: clean" in r.stderr, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_python_host_two_ranks_on_one_gpu(tmp_path):
    """the Python host with WORLD_SIZE > 1 builds its communicator from the launcher's environment and must put the env batch,
    the learner and every buffer on the communicator's device.  Two ranks of tools/py_two_rank_worker.py on the one GPU (LOCAL_RANK 0 for both,
    collectives over the shared-memory transport): both finish, on device 0, with EQUAL parameters although their env shards differ (seed + 1000 rank),
    and those differ from a single-rank run's."""
    import re, socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    rdv = str(tmp_path / "rdv"); os.mkdir(rdv, 0o700)
    base = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", RLGPU_COMM_DIR=rdv, RLGPU_COMM_TAG="py2",
                RLGPU_COMM_TRANSPORT="shm", RLGPU_COMM_TIMEOUT_S="60", RLGPU_QUIET="1")
    worker = os.path.join(ROOT, "tools", "py_two_rank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, "3"], env=dict(base, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            pytest.fail("a rank hung")
    assert [p.returncode for p in procs] == [0, 0], outs
    got = [re.search(r"rank (\d) of 2: device (\d+), parameter checksum ([0-9a-f]{16})", o) for o in outs]
    assert all(got), outs
    assert sorted(g.group(1) for g in got) == ["0", "1"] and all(g.group(2) == "0" for g in got)
    assert got[0].group(3) == got[1].group(3), outs
    solo_env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    solo = subprocess.run([sys.executable, worker, "3"], env=dict(solo_env, RLGPU_QUIET="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT, timeout=240)
    assert solo.returncode == 0, solo.stdout[-2000:]
    m = re.search(r"rank 0 of 1: device 0, parameter checksum ([0-9a-f]{16})", solo.stdout)
    assert m and m.group(1) != got[0].group(3), solo.stdout[-1000:]
