"""BatchedEnv — host handle of the on-device arena batch (rlgpu_env_* in include/rlgpu.h).

Stands where the reference has ThreadAgentManager + N x GameInst/Gym/Match/Arena
(RLGymPPO_CPP/src/private/RLGymPPO_CPP/Threading/ThreadAgentManager.h:10-69, RLGymSim_CPP/src/RLGymSim_CPP/Gym.cpp:40-102).
torch is used only to own device buffers and streams; every computation is a HIP kernel behind the C-ABI.
"""
import ctypes as C
import numpy as np
import torch

from . import _lib
from .state import ArenaState


def _chk(rc, handle, errfn):
    if rc != 0:
        raise _lib.RlgpuError(f"rlgpu error {rc}: {errfn(handle).decode()}")


class BatchedEnv:
    def __init__(self, n_envs: int, team_size: int = 1, cfg: "_lib.GymConfig | None" = None, device: int = 0, mesh="procedural"):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedEnv needs a GPU (the stepper is a HIP kernel; there is no CPU path)")
        self.lib = _lib.load()
        self.cfg = cfg if cfg is not None else _lib.default_gym_config()
        self.device = device
        self.n_envs = n_envs
        self.team_size = team_size
        self.h = C.c_void_p()
        rc = self.lib.rlgpu_env_create(C.byref(self.h), device, n_envs, team_size, C.byref(self.cfg))
        _chk(rc, self.h, self.lib.rlgpu_env_last_error)
        self.obs_size = self.lib.rlgpu_env_obs_size(self.h)
        self.n_agents = self.lib.rlgpu_env_num_agents(self.h)
        self.lib.rlgpu_env_enable_timing(self.h, 1)   # the Python host is the test / tools host: last_step_ms() and timing_total() are read there
        self.n_actions = self.lib.rlgpu_env_num_actions(self.h)
        self.players = 2 * team_size
        if isinstance(mesh, str) and mesh == "procedural":
            _chk(self.lib.rlgpu_env_set_procedural_mesh(self.h), self.h, self.lib.rlgpu_env_last_error)
            self.mesh_kind = "procedural"
        elif isinstance(mesh, str):
            _chk(self.lib.rlgpu_env_load_cmf_dir(self.h, mesh.encode()), self.h, self.lib.rlgpu_env_last_error)
            self.mesh_kind = "cmf:" + mesh
        elif mesh is not None:
            v, t = mesh
            v = np.ascontiguousarray(v, np.float32); t = np.ascontiguousarray(t, np.int32)
            _chk(self.lib.rlgpu_env_set_mesh(self.h, v.ctypes.data, len(v), t.ctypes.data, len(t)), self.h, self.lib.rlgpu_env_last_error)
            self.mesh_kind = "custom"
        else:
            self.mesh_kind = "none"

    def close(self):
        if self.h:
            self.lib.rlgpu_env_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev(self):
        return torch.device("cuda", self.device)

    def set_stream(self, stream: "torch.cuda.Stream | None"):
        ptr = stream.cuda_stream if stream is not None else 0
        _chk(self.lib.rlgpu_env_set_stream(self.h, C.c_void_p(ptr)), self.h, self.lib.rlgpu_env_last_error)

    def upload_states(self, states, env_ids=None):
        n = len(states)
        arr = (ArenaState * n)(*states)
        ids = None
        if env_ids is not None:
            ids = np.ascontiguousarray(env_ids, np.int32)
        _chk(self.lib.rlgpu_env_upload_states(self.h, C.addressof(arr), ids.ctypes.data if ids is not None else None, n),
             self.h, self.lib.rlgpu_env_last_error)

    def download_states(self, n=None, env_ids=None):
        if env_ids is not None:
            ids = np.ascontiguousarray(env_ids, np.int32); n = len(ids)
        else:
            ids = None; n = self.n_envs if n is None else n
        arr = (ArenaState * n)()
        _chk(self.lib.rlgpu_env_download_states(self.h, C.addressof(arr), ids.ctypes.data if ids is not None else None, n),
             self.h, self.lib.rlgpu_env_last_error)
        return list(arr)

    def reset(self, run_setter=True, obs: "torch.Tensor | None" = None):
        if obs is None:
            obs = torch.empty((self.n_agents, self.obs_size), dtype=torch.float32, device=self._dev())
        _chk(self.lib.rlgpu_env_reset(self.h, 1 if run_setter else 0, obs.data_ptr()), self.h, self.lib.rlgpu_env_last_error)
        return obs

    def step(self, actions: torch.Tensor, next_obs: torch.Tensor, reward: torch.Tensor, done: torch.Tensor):
        assert actions.dtype == torch.int32 and done.dtype == torch.int32 and next_obs.dtype == torch.float32
        assert actions.is_contiguous() and next_obs.is_contiguous() and reward.is_contiguous() and done.is_contiguous()
        _chk(self.lib.rlgpu_env_step(self.h, actions.data_ptr(), next_obs.data_ptr(), reward.data_ptr(), done.data_ptr()),
             self.h, self.lib.rlgpu_env_last_error)

    def step_controls(self, controls: torch.Tensor, next_obs: torch.Tensor, reward: torch.Tensor, done: torch.Tensor):
        """The step with host-parsed controls: one Action row (8 floats) per agent (rlgpu_env_step_controls)."""
        assert controls.dtype == torch.float32 and controls.is_contiguous() and controls.shape == (self.n_agents, 8)
        _chk(self.lib.rlgpu_env_step_controls(self.h, controls.data_ptr(), next_obs.data_ptr(), reward.data_ptr(), done.data_ptr()),
             self.h, self.lib.rlgpu_env_last_error)

    def reset_envs(self, env_ids, run_setter=True, obs: "torch.Tensor | None" = None):
        """Gym::Reset for the listed envs only; writes only their rows of `obs` (rlgpu_env_reset_envs)."""
        ids = np.ascontiguousarray(env_ids, np.int32)
        _chk(self.lib.rlgpu_env_reset_envs(self.h, ids.ctypes.data, len(ids), 1 if run_setter else 0, obs.data_ptr() if obs is not None else None),
             self.h, self.lib.rlgpu_env_last_error)

    def enable_snapshots(self, on=True):
        _chk(self.lib.rlgpu_env_enable_snapshots(self.h, 1 if on else 0), self.h, self.lib.rlgpu_env_last_error)

    def download_snapshots(self, first_env=0, n=None):
        """Every env's arena as it stood where the last step built its GameState (after the first tick + event tracker)."""
        n = self.n_envs - first_env if n is None else n
        arr = (ArenaState * n)()
        _chk(self.lib.rlgpu_env_download_snapshots(self.h, C.addressof(arr), first_env, n), self.h, self.lib.rlgpu_env_last_error)
        return list(arr)

    def overflow_counts(self, reset=False):
        """Narrowphase queue overflows since the last reset, process-wide: [BVH frontier, ball region, car region, item queue, result pool]."""
        out = (C.c_uint64 * 5)()
        _chk(self.lib.rlgpu_env_overflow_counts(self.h, C.addressof(out), 1 if reset else 0), self.h, self.lib.rlgpu_env_last_error)
        return [int(x) for x in out]

    def lost_contact_count(self, reset=False) -> int:
        """Contact points LOST since the last reset, process-wide: always 0 (include/rlgpu.h) -- an invariant for tests and soak runs to assert."""
        out = C.c_uint64(0)
        _chk(self.lib.rlgpu_env_lost_contact_count(self.h, C.byref(out), 1 if reset else 0), self.h, self.lib.rlgpu_env_last_error)
        return int(out.value)

    def set_collect_queue(self, mode: int):
        """-1: lockstep collection takes the step queue when its wavefront-groups do not all fit the device (default); 0: never; 1: always."""
        _chk(self.lib.rlgpu_env_set_collect_queue(self.h, int(mode)), self.h, self.lib.rlgpu_env_last_error)

    def big_layout_ticks(self, reset=False) -> int:
        """Env-ticks since the last reset (process-wide) whose contacts did not fit the LDS layout and were redone with the big one."""
        out = C.c_uint64(0)
        _chk(self.lib.rlgpu_env_big_layout_ticks(self.h, C.byref(out), 1 if reset else 0), self.h, self.lib.rlgpu_env_last_error)
        return int(out.value)

    def epa_counts(self, reset=False):
        """Penetration-depth queries (Bullet's second GJK + EPA, csrc/arena_epa.h) since the last reset, process-wide: [queries, of them in the full-size arena]."""
        out = (C.c_uint64 * 2)()
        _chk(self.lib.rlgpu_env_epa_counts(self.h, C.addressof(out), 1 if reset else 0), self.h, self.lib.rlgpu_env_last_error)
        return [int(x) for x in out]

    def enable_step_stats(self, on=True):
        _chk(self.lib.rlgpu_env_enable_step_stats(self.h, 1 if on else 0), self.h, self.lib.rlgpu_env_last_error)

    def step_stats(self, reset=True):
        """{player-steps, sum |car vel|, ball touches, airborne player-steps} accumulated by the step kernels since the last reset."""
        out = (C.c_float * 4)()
        _chk(self.lib.rlgpu_env_step_stats(self.h, C.addressof(out), 1 if reset else 0), self.h, self.lib.rlgpu_env_last_error)
        return np.array(out[:], np.float64)

    def collect(self, ppo, T: int, obs: torch.Tensor, actions: torch.Tensor, logp: torch.Tensor, reward: torch.Tensor, done: torch.Tensor, deterministic=False) -> bool:
        """T x (policy act + gym step) in one launch (rlgpu_collect).  obs: [T+1][n_agents][D] with obs[0] current; the others [T][n_agents].
        Returns False -- nothing launched -- when the policy does not fit the in-kernel inference (the caller then alternates act / step)."""
        assert obs.is_contiguous() and actions.is_contiguous() and logp.is_contiguous() and reward.is_contiguous() and done.is_contiguous()
        assert obs.shape[0] >= T + 1 and actions.dtype == torch.int32 and done.dtype == torch.int32
        rc = self.lib.rlgpu_collect(self.h, ppo.h, T, obs.data_ptr(), actions.data_ptr(), logp.data_ptr(), reward.data_ptr(), done.data_ptr(), 1 if deterministic else 0)
        if rc == -3:   # RLGPU_ERR_STATE
            return False
        _chk(rc, self.h, self.lib.rlgpu_env_last_error)
        return True

    def collect_free(self, ppo, T_cap: int, target_agent_steps: int, obs, actions, logp, reward, done, steps, deterministic=False) -> bool:
        """The collection phase as the reference's agent threads run it (rlgpu_collect_free): every wavefront steps its envs until the launch has
        `target_agent_steps` together, at most T_cap steps each; steps [n_envs] int32 = gym steps each env made.  Buffers as for collect() with T = T_cap.
        Returns False -- nothing launched -- when the launch would not be resident as a whole or the policy does not fit (collect in lockstep then)."""
        assert obs.is_contiguous() and actions.is_contiguous() and logp.is_contiguous() and reward.is_contiguous() and done.is_contiguous()
        assert obs.shape[0] >= T_cap + 1 and actions.dtype == torch.int32 and done.dtype == torch.int32 and steps.dtype == torch.int32 and steps.numel() >= self.n_envs
        rc = self.lib.rlgpu_collect_free(self.h, ppo.h, T_cap, C.c_int64(int(target_agent_steps)), obs.data_ptr(), actions.data_ptr(), logp.data_ptr(), reward.data_ptr(),
                                         done.data_ptr(), steps.data_ptr(), 1 if deterministic else 0)
        if rc == -3:   # RLGPU_ERR_STATE
            return False
        _chk(rc, self.h, self.lib.rlgpu_env_last_error)
        return True

    def reseed(self, seed_lo: int, seed_hi: int):
        _chk(self.lib.rlgpu_env_reseed(self.h, C.c_uint32(seed_lo & 0xffffffff), C.c_uint32(seed_hi & 0xffffffff)), self.h, self.lib.rlgpu_env_last_error)

    def set_controls(self, controls):
        """controls: float32 [n_envs, 2 * team_size, 8] (CarControls order); only the cars' controls change, the state is not rounded."""
        import numpy as np
        c = np.ascontiguousarray(controls, np.float32)
        assert c.shape == (self.n_envs, 2 * self.team_size, 8), c.shape
        _chk(self.lib.rlgpu_env_set_controls(self.h, c.ctypes.data), self.h, self.lib.rlgpu_env_last_error)

    def physics_ticks(self, ticks: int):
        _chk(self.lib.rlgpu_env_physics_ticks(self.h, ticks), self.h, self.lib.rlgpu_env_last_error)

    def sync(self):
        _chk(self.lib.rlgpu_env_sync(self.h), self.h, self.lib.rlgpu_env_last_error)

    def last_step_ms(self) -> float:
        ms = C.c_float()
        _chk(self.lib.rlgpu_env_last_step_ms(self.h, C.byref(ms)), self.h, self.lib.rlgpu_env_last_error)
        return ms.value

    def timing_total(self, reset=True):
        """(sum of step-kernel ms, launches) since the last reset, from hipEvents on the env's stream."""
        ms, n = C.c_float(), C.c_int()
        _chk(self.lib.rlgpu_env_timing_total(self.h, C.byref(ms), C.byref(n), 1 if reset else 0), self.h, self.lib.rlgpu_env_last_error)
        return ms.value, n.value

    def state_words(self) -> int:
        return self.lib.rlgpu_env_state_words(self.h)

    def check_redzones(self) -> None:
        """Debug mode (RLGPU_REDZONE=<bytes> in the environment when the batch was created): raises, naming the buffer, if a kernel wrote past the end
        of one of the batch's device buffers."""
        _chk(self.lib.rlgpu_env_check_redzones(self.h), self.h, self.lib.rlgpu_env_last_error)


def procedural_mesh():
    lib = _lib.load()
    nv, nt = C.c_int(), C.c_int()
    lib.rlgpu_procedural_mesh(None, 0, None, 0, C.byref(nv), C.byref(nt))
    v = np.zeros((nv.value, 3), np.float32); t = np.zeros((nt.value, 3), np.int32)
    rc = lib.rlgpu_procedural_mesh(v.ctypes.data, nv.value, t.ctypes.data, nt.value, C.byref(nv), C.byref(nt))
    assert rc == 0
    return v, t


def procedural_mesh_ex(fillet_segments=12, max_edge_uu=330.0):
    """The procedural arena at a chosen resolution (rlgpu_procedural_mesh_ex): the defaults give ~9 k triangles."""
    lib = _lib.load()
    nv, nt = C.c_int(), C.c_int()
    lib.rlgpu_procedural_mesh_ex(fillet_segments, C.c_float(max_edge_uu), None, 0, None, 0, C.byref(nv), C.byref(nt))
    v = np.zeros((nv.value, 3), np.float32); t = np.zeros((nt.value, 3), np.int32)
    rc = lib.rlgpu_procedural_mesh_ex(fillet_segments, C.c_float(max_edge_uu), v.ctypes.data, nv.value, t.ctypes.data, nt.value, C.byref(nv), C.byref(nt))
    assert rc == 0
    return v, t


def write_cmf_files(verts_uu, tris, out_dir, n_files=16):
    """Split a mesh into n_files .cmf files the way the game's soccar set is split (RocketSim::Init reads a directory of them:
    RS/RocketSim.cpp:70-212; format CollisionMeshFile.cpp:11-36: i32 nTris, i32 nVerts, index triplets, vertices in Bullet units = uu / 50).
    The triangles are grouped by the angle of their centroid around the field centre, so every file is one sector of the arena."""
    import os
    os.makedirs(out_dir, exist_ok=True)
    cen = verts_uu[tris].mean(axis=1)
    sector = ((np.arctan2(cen[:, 1], cen[:, 0]) + np.pi) / (2 * np.pi) * n_files).astype(np.int64).clip(0, n_files - 1)
    paths = []
    for k in range(n_files):
        tk = tris[sector == k]
        if len(tk) == 0:
            continue
        used = np.unique(tk); remap = -np.ones(len(verts_uu), np.int64); remap[used] = np.arange(len(used))
        path = os.path.join(out_dir, "mesh_%02d.cmf" % k)
        with open(path, "wb") as f:
            f.write(np.int32(len(tk)).tobytes()); f.write(np.int32(len(used)).tobytes())
            f.write(remap[tk].astype(np.int32).tobytes()); f.write((verts_uu[used] / 50.0).astype(np.float32).tobytes())
        paths.append(path)
    return paths


def action_table():
    lib = _lib.load()
    tab = np.zeros((128, 8), np.float32)
    n = lib.rlgpu_action_table(tab.ctypes.data, 128)
    return tab[:n].copy()
