"""Checkpoint payload interchange with the reference: rlgpu_lt_* against archives written by the real libtorch.

tests/golden/lt_model.lt / lt_optim.lt come from torch::save(nn::Sequential) and optim::Adam::save -- the calls of
PPOLearner.cpp:408-411,466-472 -- made by tests/golden/gen_lt_fixture.cpp with the torch wheel's libtorch; lt_expected.f32
holds the tensors that program had in memory.  The opposite direction loads this repo's archives with torch.jit.load and, when the
wheel's C++ headers are there, with the reference's own loading calls (tests/cpp/lt_libtorch_check.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from rlgymppo_cpp_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
DIMS = np.array([7, 6, 5, 3], np.int32)
NP = int(sum(DIMS[i] * DIMS[i + 1] + DIMS[i + 1] for i in range(3)))


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


@pytest.fixture(scope="module")
def expected():
    e = np.fromfile(os.path.join(GOLD, "lt_expected.f32"), "<f4")
    assert e.size == 3 * NP + 1
    return e[:NP], e[NP:2 * NP], e[2 * NP:3 * NP], int(e[-1])


def _read_model(lib, path, dims=DIMS):
    n = int(sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1)))
    out = np.zeros(n, np.float32)
    rc = lib.rlgpu_lt_read_model(path.encode(), dims.ctypes.data, len(dims) - 1, out.ctypes.data)
    return rc, out


def _read_adam(lib, path):
    m = np.zeros(NP, np.float32); v = np.zeros(NP, np.float32); step = C.c_int64(-1)
    rc = lib.rlgpu_lt_read_adam(path.encode(), DIMS.ctypes.data, 3, m.ctypes.data, v.ctypes.data, C.byref(step))
    return rc, m, v, step.value


def test_reads_libtorch_archives_bit_exact(lib, expected):
    params, m, v, step = expected
    rc, got = _read_model(lib, os.path.join(GOLD, "lt_model.lt"))
    assert rc == 0, lib.rlgpu_lt_last_error()
    assert (got == params).all()
    rc, gm, gv, gs = _read_adam(lib, os.path.join(GOLD, "lt_optim.lt"))
    assert rc == 0, lib.rlgpu_lt_last_error()
    assert (gm == m).all() and (gv == v).all() and gs == step
    assert np.abs(m).max() > 0 and v.max() > 0


def test_read_errors(lib, tmp_path):
    rc, _ = _read_model(lib, os.path.join(GOLD, "lt_model.lt"), np.array([7, 6, 6, 3], np.int32))
    assert rc != 0 and b"different size" in lib.rlgpu_lt_last_error()      # PPOLearner.cpp:390-406
    rc, _ = _read_model(lib, os.path.join(GOLD, "lt_model.lt"), np.array([7, 6, 3], np.int32))
    assert rc != 0 and b"different size" in lib.rlgpu_lt_last_error()
    rc, _ = _read_model(lib, str(tmp_path / "missing.lt"))
    assert rc != 0 and b"does not exist" in lib.rlgpu_lt_last_error()           # PPOLearner.cpp:377-378
    junk = tmp_path / "junk.lt"
    junk.write_bytes(b"RLGPU_LT1\n" + bytes(100))
    rc, _ = _read_model(lib, str(junk))
    assert rc != 0 and b"not a zip" in lib.rlgpu_lt_last_error()
    cut = tmp_path / "cut.lt"
    cut.write_bytes(open(os.path.join(GOLD, "lt_model.lt"), "rb").read()[:3000])
    rc, _ = _read_model(lib, str(cut))
    assert rc != 0


def _write_pair(lib, tmp_path, params, m, v, step, lr=2e-4):
    mp, op = str(tmp_path / "PPO_POLICY.lt"), str(tmp_path / "PPO_POLICY_OPTIM.lt")
    assert lib.rlgpu_lt_write_model(mp.encode(), DIMS.ctypes.data, 3, params.ctypes.data) == 0, lib.rlgpu_lt_last_error()
    assert lib.rlgpu_lt_write_adam(op.encode(), DIMS.ctypes.data, 3, lr, m.ctypes.data, v.ctypes.data, step) == 0, lib.rlgpu_lt_last_error()
    return mp, op


def test_written_archives_round_trip_and_load_in_torch(lib, expected, tmp_path):
    torch = pytest.importorskip("torch")
    params, m, v, step = (np.ascontiguousarray(x) if isinstance(x, np.ndarray) else x for x in expected)
    mp, op = _write_pair(lib, tmp_path, params, m, v, step)
    rc, got = _read_model(lib, mp)
    assert rc == 0 and (got == params).all()
    rc, gm, gv, gs = _read_adam(lib, op)
    assert rc == 0 and (gm == m).all() and (gv == v).all() and gs == step
    mod = torch.jit.load(mp)
    names = [k for k, _ in mod.named_parameters()]
    assert names == ["0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias"]          # state-dict keys of the reference's Sequential
    assert [k for k, _ in mod.named_children()] == ["0", "1", "2", "3", "4"]                    # ReLU children are present (Module::load reads every child)
    flat = np.concatenate([p.detach().numpy().ravel() for _, p in mod.named_parameters()])
    assert (flat == params).all()
    opt = torch.jit.load(op)
    assert opt.pytorch_version == "1.5.0"
    group = getattr(opt.param_groups, "param_groups/0")
    keys = [getattr(group, f"params/{i}") for i in range(6)]
    got_m = np.concatenate([getattr(opt.state, k).exp_avg.numpy().ravel() for k in keys])
    assert (got_m == m).all() and getattr(opt.state, keys[0]).step == step and abs(group.options.lr - 2e-4) < 1e-9
    # a never-stepped optimizer has no per-parameter state (Adam creates it lazily); reading it back gives zeros and step 0
    z = np.zeros(NP, np.float32)
    _, op0 = _write_pair(lib, tmp_path, params, z, z, 0)
    rc, gm, gv, gs = _read_adam(lib, op0)
    assert rc == 0 and gs == 0 and not gm.any() and not gv.any()
    assert len(list(torch.jit.load(op0).state.named_children())) == 0


def test_written_archives_load_with_the_reference_calls(lib, expected, tmp_path):
    """torch::load(seq, ifstream) + InputArchive::load_from + Adam::load -- the reference's loader -- on this repo's archives."""
    torch = pytest.importorskip("torch")
    ti = os.path.dirname(torch.__file__)
    if not os.path.exists(os.path.join(ti, "include", "torch", "csrc", "api", "include", "torch", "torch.h")):
        pytest.skip("the torch wheel has no C++ headers here")
    exe = str(tmp_path / "lt_check")
    r = subprocess.run(["g++", "-std=c++17", "-O0", os.path.join(HERE, "cpp", "lt_libtorch_check.cpp"), f"-I{ti}/include",
                        f"-I{ti}/include/torch/csrc/api/include", f"-L{ti}/lib", "-ltorch", "-ltorch_cpu", "-lc10", f"-Wl,-rpath,{ti}/lib", "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    params, m, v, step = (np.ascontiguousarray(x) if isinstance(x, np.ndarray) else x for x in expected)
    mp, op = _write_pair(lib, tmp_path, params, m, v, step, lr=3e-4)
    out = str(tmp_path / "got.f32")
    r = subprocess.run([exe, mp, op, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    got = np.fromfile(out, "<f4")
    assert got.size == 3 * NP + 2
    assert (got[:NP] == params).all() and (got[NP:2 * NP] == m).all() and (got[2 * NP:3 * NP] == v).all()
    assert int(got[-2]) == step and abs(got[-1] - 3e-4) < 1e-9
