"""The C++ host layer (include/RLGymSim_CPP, include/RLGymPPO_CPP, rlgymppo_cpp_amd/host): a program written against the
reference's public API compiles against this repo's headers (CPU) and trains on the GPU."""
import os
import subprocess
import sys

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
              "-L", PKG, "-lrlgpu", f"-Wl,-rpath,{PKG}", "-Wl,-rpath-link,/opt/rocm/lib"])
    assert r.returncode == 0, r.stdout
    r = _run([exe])
    assert r.returncode == 0 and "host api ok" in r.stdout, r.stdout


def test_example_program_is_built():
    """CPU: build() produced the host library and the example program written against the reference's API."""
    for f in ("librlgymppo_amd.so", "example_main"):
        assert os.path.exists(os.path.join(PKG, f)), f


@pytest.mark.gpu
def test_example_program_trains_and_checkpoints(tmp_path):
    """GPU: two iterations of the example program (step + iteration callbacks, slow GameState path), a checkpoint, and a resume."""
    exe = os.path.join(PKG, "example_main")
    ck = str(tmp_path / "ck")
    r = _run([exe, "2", "4", "16", "4096", ck], cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("ITERATION COMPLETED") == 2 and "player_speed" in r.stdout and "Policy Entropy" in r.stdout
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
    r = _run([exe, "1", "4", "16", "4096", ck], cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0 and "loaded checkpoint" in r.stdout and str(3 * 4096) in os.listdir(ck), r.stdout[-3000:]
