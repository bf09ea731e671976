"""Free-running collection (rlgpu_collect_free; ThreadAgentManager::CollectTimesteps, PRIV/Threading/ThreadAgentManager.cpp:16-82: the agents
step at their own pace and the manager takes what every trajectory holds once the total is there) and what consumes its ragged iterations:
the experience FIFO (against the REAL ExperienceBuffer.cpp), GAE over trajectories of their own lengths (against the REAL ComputeGAE on the
concatenation the reference would build), the device-side row lookup, and the collection launch itself (a game's rows do not depend on the pace)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import learner_ref as R  # noqa: E402

SO = os.path.join(ROOT, "oracle", "_ref", "libref_learner.so")


@pytest.fixture(scope="module")
def refl():
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libref_learner.so not built (needs /root/reference + libtorch headers)")
    return C.CDLL(SO)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _ragged_fifo(max_rows, T_cap, n, min_rows):
    from rlgymppo_cpp_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.rlgpu_expbuf_create_ragged(C.byref(h), max_rows, T_cap, n, min_rows) == 0
    return lib, h


@pytest.mark.parametrize("max_rows,keep_last", [(130, 0), (60, 0), (400, 0), (150, 48)])
def test_ragged_experience_fifo_equals_reference_experience_buffer(refl, max_rows, keep_last):
    """Submissions of DIFFERENT sizes -- the concatenation of trajectories of their own lengths, some of them empty -- through the REAL
    ExperienceBuffer (SubmitExperience + GetAllBatchesShuffled) and through rlgpu_expbuf_submit_ragged + the two host-side row lookups
    (rlgpu_expbuf_shuffled_rows; rlgpu_shuffler_next_i32 + rlgpu_expbuf_map_rows, what a host that draws ahead uses): the same experience in
    the same batches.  keep_last: only a submission's last rows join (the multi-GPU rule), i.e. the reference fed the truncated submission."""
    from rlgymppo_cpp_amd.learner import Shuffler
    T_cap, N, seed, n_submits, batch = 9, 12, 321, 8, 16
    rs = np.random.RandomState(max_rows)
    lens = rs.randint(0, T_cap + 1, size=(n_submits, N)).astype(np.int32)
    lens[2, :3] = 0; lens[5] = np.maximum(lens[5], 4)
    min_rows = 40
    lens[lens.sum(axis=1) < min_rows] = T_cap // 2 + 1
    sizes_full = lens.sum(axis=1).astype(np.int32)
    sizes = np.minimum(sizes_full, keep_last).astype(np.int32) if keep_last else sizes_full
    ids = np.zeros(int(n_submits * (max_rows + sizes.max())), np.int64); counts = np.zeros(n_submits, np.int32)
    refl.refl_expbuf_run_sizes.restype = C.c_int64
    n = refl.refl_expbuf_run_sizes(C.c_int64(max_rows), seed, n_submits, _p(sizes), C.c_int64(batch), _p(ids), C.c_int64(ids.size), _p(counts))
    assert 0 < n <= ids.size
    lib, h = _ragged_fifo(max_rows, T_cap, N, min_rows)
    shuf_a, shuf_b = Shuffler(seed), Shuffler(seed)
    B = T_cap * N
    # reference identity -> device row: submission s holds ids first[s] + i for its kept rows, i.e. logical rows skip .. rows-1 of the iteration
    first = np.concatenate([[0], np.cumsum(sizes)])[:-1]
    at = 0
    slot_of = {}
    for s in range(n_submits):
        slot = C.c_int()
        assert lib.rlgpu_expbuf_submit_ragged(h, _p(lens[s]), C.c_int64(keep_last), C.byref(slot)) == 0
        slot_of = {k: v for k, v in slot_of.items() if v != slot.value}; slot_of[s] = slot.value
        cur = int(lib.rlgpu_expbuf_size(h))
        assert cur // batch == counts[s]
        rows_a = np.empty(cur, np.int32)
        assert lib.rlgpu_expbuf_shuffled_rows(h, shuf_a.h, _p(rows_a)) == 0
        perm = np.empty(cur, np.int32); rows_b = np.empty(cur, np.int32)
        assert lib.rlgpu_shuffler_next_i32(shuf_b.h, C.c_int64(cur), _p(perm)) == 0
        assert lib.rlgpu_expbuf_map_rows(h, _p(perm), C.c_int64(cur), _p(rows_b)) == 0
        assert (rows_a == rows_b).all()
        for b in range(counts[s]):
            got = ids[at:at + batch]; at += batch
            k = np.searchsorted(first, got, side="right") - 1                     # the submission each row came from
            a = got - first[k] + (sizes_full[k] - sizes[k])                       # its index in that iteration's concatenation
            off = np.concatenate([np.zeros((n_submits, 1), np.int64), np.cumsum(lens, axis=1)], axis=1)
            ag = np.array([np.searchsorted(off[kk], aa, side="right") - 1 for kk, aa in zip(k, a)])
            t = a - off[k, ag]
            assert (t >= 0).all() and (t < lens[k, ag]).all()
            phys = np.array([slot_of[int(i)] for i in k]) * B + t * N + ag
            assert (rows_a[b * batch:(b + 1) * batch] == phys).all()
    assert at == n
    lib.rlgpu_expbuf_destroy(h)


def test_shuffler_state_round_trip():
    """A permutation drawn ahead for the wrong size is taken back: the engine restored from its text form draws what it would have drawn."""
    from rlgymppo_cpp_amd import _lib
    from rlgymppo_cpp_amd.learner import Shuffler
    lib = _lib.load()
    a, b = Shuffler(5), Shuffler(5)
    a.next(100); b.next(100)
    buf = C.create_string_buffer(256)
    assert lib.rlgpu_shuffler_get_state(a.h, buf, 256) == 0
    a.next(333)                      # the wrong guess
    assert lib.rlgpu_shuffler_set_state(a.h, buf.value) == 0
    assert (a.next(257) == b.next(257)).all()
    p32 = np.empty(50, np.int32)
    assert lib.rlgpu_shuffler_next_i32(a.h, C.c_int64(50), _p(p32)) == 0
    assert (p32 == b.next(50)).all()


def _concat_ragged(x, lens):
    """time-major [T][n] -> the reference's batch: the non-empty trajectories one after the other"""
    return np.concatenate([x[:lens[j], j] for j in range(len(lens)) if lens[j] > 0])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,ret_std,clip", [(0, 1.0, 10.0), (0, 2.5, 0.7), (1, 1.0, 10.0), (0, 0.0, 0.0)])
def test_hip_ragged_gae_equals_reference_compute_gae_on_the_concatenation(refl, mode, ret_std, clip):
    """rlgpu_gae_ragged on trajectories of their own lengths (some empty) against TorchFuncs::ComputeGAE (the real code) run on the batch the
    reference would hand it: the concatenation of the non-empty trajectories with the collector's truncation marks and B + 1 value rows
    (Learner.cpp:619-640).  Mode 0 = the reference as is (the row after a trajectory's end is the next trajectory's first state, quirk Q1);
    mode 1 = every trajectory bootstraps from its own next state, i.e. the reference run per trajectory.  1e-4 on returns / advantages."""
    import torch
    from rlgymppo_cpp_amd.ppo import PPOCore
    T_cap, envs, P = 11, 37, 2
    n = envs * P
    rs = np.random.RandomState(7 + mode)
    steps = rs.randint(0, T_cap + 1, size=envs).astype(np.int32)
    steps[3] = 0; steps[envs - 1] = 0 if mode == 0 else 5    # an empty trajectory in the middle, and (mode 0) at the very end
    lens = np.repeat(steps, P)
    rews = (rs.randn(T_cap, n) * 2).astype(np.float32); vals = rs.randn(T_cap + 1, n).astype(np.float32)
    dones = (rs.rand(T_cap, n) < 0.06).astype(np.float32)
    dev = torch.device("cuda", 0)
    core = PPOCore(8, 4, (16,), (16,), max_rows=256)
    d = lambda a: torch.from_numpy(a).to(dev).contiguous()
    adv, tgt, ret = (torch.full((T_cap, n), 7.5, device=dev) for _ in range(3))
    rews_d, dones_d, vals_d, steps_d = d(rews), d(dones), d(vals), d(steps)     # (kept alive until the launch has run)
    rc = core.lib.rlgpu_gae_ragged(core.h, rews_d.data_ptr(), dones_d.data_ptr(), None, vals_d.data_ptr(), n, steps_d.data_ptr(), P,
                                   C.c_float(0.99), C.c_float(0.95), C.c_float(ret_std), C.c_float(clip), mode, adv.data_ptr(), tgt.data_ptr(), ret.data_ptr())
    assert rc == 0
    core.sync()
    got = [x.cpu().numpy() for x in (adv, tgt, ret)]
    truncs = np.zeros((T_cap, n), np.float32)
    for j in range(n):
        if lens[j] > 0:
            truncs[lens[j] - 1, j] = 1.0 - dones[lens[j] - 1, j]
    adv_w, tgt_w, ret_w = (np.full((T_cap, n), 7.5, np.float32) for _ in range(3))
    from test_ref_learner import ref_gae
    live = [j for j in range(n) if lens[j] > 0]
    if mode == 0:
        vcat = np.concatenate([_concat_ragged(vals[:T_cap], lens), [vals[lens[live[-1]], live[-1]]]]).astype(np.float32)
        a, t, r = ref_gae(refl, _concat_ragged(rews, lens), _concat_ragged(dones, lens), _concat_ragged(truncs, lens), vcat, 0.99, 0.95, ret_std, clip)
        at = 0
        for j in live:
            adv_w[:lens[j], j] = a[at:at + lens[j]]; tgt_w[:lens[j], j] = t[at:at + lens[j]]; ret_w[:lens[j], j] = r[at:at + lens[j]]; at += lens[j]
    else:
        for j in live:
            L = lens[j]
            a, t, r = ref_gae(refl, rews[:L, j].copy(), dones[:L, j].copy(), truncs[:L, j].copy(), vals[:L + 1, j].copy(), 0.99, 0.95, ret_std, clip)
            adv_w[:L, j] = a; tgt_w[:L, j] = t; ret_w[:L, j] = r
    for g, w, name in zip(got, (adv_w, tgt_w, ret_w), ("advantages", "targets", "returns")):
        assert np.abs(g - w).max() <= 1e-4 * max(1.0, np.abs(w).max()), name     # rows beyond a trajectory keep their 7.5: not written


@pytest.mark.gpu
def test_hip_row_lookup_on_the_device_equals_the_host_lookup():
    """rlgpu_traj_offsets + rlgpu_expbuf_map_rows_dev (the C++ host's path: the permutation is drawn ahead, the rows it stands for are looked up
    on the device once the iteration's trajectory lengths are known) against rlgpu_expbuf_map_rows, FIFO of three ragged iterations with a
    partly shifted-out oldest one."""
    import torch
    from rlgymppo_cpp_amd.learner import Shuffler
    T_cap, envs, P = 13, 211, 2
    n = envs * P
    lib, h = _ragged_fifo(3 * 1500, T_cap, n, 1300)
    S = lib.rlgpu_expbuf_num_slots(h)
    dev = torch.device("cuda", 0)
    off_dev = torch.zeros((S, n + 1), dtype=torch.int32, device=dev)
    rs = np.random.RandomState(3)
    shuf = Shuffler(11)
    for it in range(5):
        steps = rs.randint(0, T_cap + 1, size=envs).astype(np.int32); steps[rs.randint(envs)] = 0
        lens = np.repeat(steps, P).astype(np.int32)
        slot = C.c_int()
        assert lib.rlgpu_expbuf_submit_ragged(h, _p(lens), C.c_int64(0), C.byref(slot)) == 0
        st = torch.from_numpy(steps).to(dev)
        assert lib.rlgpu_traj_offsets(st.data_ptr(), n, P, off_dev[slot.value].data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert (off_dev[slot.value].cpu().numpy() == np.concatenate([[0], np.cumsum(lens)])).all()
        cur = int(lib.rlgpu_expbuf_size(h))
        perm = np.empty(cur, np.int32); want = np.empty(cur, np.int32)
        assert lib.rlgpu_shuffler_next_i32(shuf.h, C.c_int64(cur), _p(perm)) == 0
        assert lib.rlgpu_expbuf_map_rows(h, _p(perm), C.c_int64(cur), _p(want)) == 0
        pd = torch.from_numpy(perm).to(dev); got = torch.empty(cur, dtype=torch.int32, device=dev)
        assert lib.rlgpu_expbuf_map_rows_dev(h, pd.data_ptr(), C.c_int64(cur), off_dev.data_ptr(), got.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert (got.cpu().numpy() == want).all()
    lib.rlgpu_expbuf_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("team,n_envs,use_bf16", [(1, 1024, True), (2, 96, True), (1, 128, False), (3, 60, True)])
def test_free_running_collection_rows_are_the_lockstep_launchs(team, n_envs, use_bf16):
    """Two identical env batches and samplers: rlgpu_collect (lockstep, T_cap steps) on one, rlgpu_collect_free (target = half of that) on the
    other.  Every game's rows under the free-running launch -- observations, actions, log-probs, rewards, dones -- are the lockstep launch's
    first steps[game] rows BIT FOR BIT (the sampler's counters and the stepper do not know the pace), the launch gathered at least the target
    and less than the target plus one step of every game, and no game went beyond T_cap."""
    import torch
    from rlgymppo_cpp_amd.env import BatchedEnv
    from rlgymppo_cpp_amd.ppo import PPOCore
    dev = torch.device("cuda", 0)
    T = 6; CAP = 2 * T
    from rlgymppo_cpp_amd import _lib
    cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 9        # episodes end (and restart) inside the compared window
    ea, eb = BatchedEnv(n_envs, team, cfg), BatchedEnv(n_envs, team, cfg)
    core = PPOCore(ea.obs_size, ea.n_actions, (64, 64), (64, 64), use_bf16=use_bf16, max_rows=4096)
    N, D, P = ea.n_agents, ea.obs_size, ea.n_agents // n_envs

    def bufs():
        return (torch.zeros((CAP + 1, N, D), device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev), torch.zeros((CAP, N), device=dev),
                torch.full((CAP, N), -777.0, device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev))
    A, Bf = bufs(), bufs()
    steps = torch.full((n_envs,), -1, dtype=torch.int32, device=dev)
    stream, ctr = core.get_sampler()
    ea.reset(True, A[0][0])
    for _ in range(3):                                       # a few launches so that episodes end inside the compared window
        assert ea.collect(core, CAP, *A); A[0][0].copy_(A[0][CAP])
    assert ea.collect(core, CAP, *A); ea.sync()
    core.set_sampler(stream, ctr)
    eb.reset(True, Bf[0][0])
    for _ in range(3):
        assert eb.collect(core, CAP, *Bf); Bf[0][0].copy_(Bf[0][CAP])
    Bf[3].fill_(-777.0)
    assert eb.collect_free(core, CAP, T * N, *Bf, steps), "the launch must be resident as a whole at these sizes"
    eb.sync()
    st = steps.cpu().numpy()
    total = int(st.sum()) * P
    assert st.min() >= 0 and st.max() <= CAP and T * N <= total < T * N + N + 1, (st.min(), st.max(), total)
    mask = (np.arange(CAP)[:, None] < st[None, :])
    for name, a, b in zip(("actions", "logp", "reward", "done"), A[1:], Bf[1:]):
        a = a.cpu().numpy().reshape(CAP, n_envs, P); b = b.cpu().numpy().reshape(CAP, n_envs, P)
        assert not ((a != b) & mask[:, :, None]).any(), name
    oa = A[0].cpu().numpy().reshape(CAP + 1, n_envs, P * D); ob = Bf[0].cpu().numpy().reshape(CAP + 1, n_envs, P * D)
    assert not ((oa != ob) & (np.arange(CAP + 1)[:, None] <= st[None, :])[:, :, None]).any(), "observations"
    rb = Bf[3].cpu().numpy().reshape(CAP, n_envs, P)
    assert (rb[~mask] == -777.0).all(), "rows beyond a game's own count must stay untouched"
    assert A[4].sum().item() > 0, "no episode ended in the compared window"


@pytest.mark.gpu
def test_free_running_collection_refuses_a_batch_that_is_not_resident_at_once():
    import torch
    from rlgymppo_cpp_amd.env import BatchedEnv
    from rlgymppo_cpp_amd.ppo import PPOCore
    dev = torch.device("cuda", 0)
    n_envs = 4 * 1024 + 64 * 4 * 2      # more wavefronts of four 1v1 games than the 1024 SIMDs hold at one per SIMD (and then some)
    env = BatchedEnv(n_envs, 1)
    core = PPOCore(env.obs_size, env.n_actions, (64, 64), (64, 64), use_bf16=True, max_rows=4096)
    N, D = env.n_agents, env.obs_size
    obs = torch.zeros((3, N, D), device=dev); i32 = torch.zeros((2, N), dtype=torch.int32, device=dev); f = torch.zeros((2, N), device=dev)
    steps = torch.zeros(n_envs, dtype=torch.int32, device=dev)
    env.reset(True, obs[0])
    assert env.collect_free(core, 2, N, obs, i32, f, f.clone(), i32.clone(), steps) is False
    assert env.collect(core, 2, obs, i32, f, f.clone(), i32.clone())      # ... which the lockstep launch takes
    env.sync()
