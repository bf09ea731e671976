"""The learner oracle (oracle/learner_ref.py) and the HIP learner kernels against the REAL reference learner code:
oracle/_ref/libref_learner.so = TorchFuncs.cpp + DiscretePolicy.cpp + ValueEstimator.cpp of the reference, compiled unedited from
/root/reference against the torch wheel's libtorch (oracle/Makefile, target ref_learner; oracle/ref_learner_driver.cpp only marshals
arguments).  Skipped where the library was not built (no /root/reference and no prebuilt copy)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import learner_ref as R  # noqa: E402

SO = os.path.join(ROOT, "oracle", "_ref", "libref_learner.so")
HID, D, A = [48, 32], 89, 90


@pytest.fixture(scope="module")
def refl():
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libref_learner.so not built (needs /root/reference + libtorch headers)")
    return C.CDLL(SO)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _net(rs, out):
    dims = [D] + HID + [out]
    shapes = [((dims[i + 1], dims[i]), (dims[i + 1],)) for i in range(len(dims) - 1)]
    flat = np.concatenate([(rs.uniform(-1, 1, ws[0] * ws[1] + bs[0]) / np.sqrt(ws[1])).astype(np.float32) for ws, bs in shapes])
    return np.ascontiguousarray(flat, np.float32), shapes


def ref_gae(refl, rews, dones, truncs, values, gamma, lam, ret_std, clip):
    B = len(rews)
    adv = np.zeros(B, np.float32); tgt = np.zeros(B, np.float32); ret = np.zeros(B, np.float32)
    refl.refl_gae(_p(rews), _p(dones), _p(truncs), _p(values), B, C.c_float(gamma), C.c_float(lam), C.c_float(ret_std), C.c_float(clip), _p(adv), _p(tgt), _p(ret))
    return adv, tgt, ret


def ref_policy(refl, flat, obs, acts, temperature=1.0, seed=0):
    rows = len(obs)
    hid = np.array(HID, np.int32)
    probs = np.zeros((rows, A), np.float32); am = np.zeros(rows, np.int32); lp = np.zeros(rows, np.float32); ent = C.c_float()
    sa = np.zeros(rows, np.int32); sl = np.zeros(rows, np.float32); tape = np.zeros((rows, A), np.float32)
    refl.refl_policy(_p(hid), len(HID), D, A, _p(flat), C.c_float(temperature), _p(obs), rows, _p(acts), _p(probs), _p(am), _p(lp), C.byref(ent),
                     C.c_uint64(seed), _p(sa), _p(sl), _p(tape))
    return probs, am, lp, ent.value, sa, sl, tape


@pytest.mark.parametrize("ret_std,clip", [(1.0, 10.0), (2.5, 0.7), (0.0, 10.0), (3.0, 0.0)])
def test_gae_oracle_equals_reference_compute_gae(refl, ret_std, clip):
    """TorchFuncs::ComputeGAE (TorchFuncs.cpp:5-52) on a concatenated batch with terminals and truncations; the cases cover reward
    standardisation with clipping that bites, no standardisation (retStd 0) and no clipping (clipRange 0)."""
    rs = np.random.RandomState(int(ret_std * 10 + clip * 100))
    B = 700
    rews = (rs.randn(B) * 2).astype(np.float32); values = rs.randn(B + 1).astype(np.float32)
    dones = (rs.rand(B) < 0.03).astype(np.float32); truncs = ((rs.rand(B) < 0.02) & (dones == 0)).astype(np.float32)
    want = ref_gae(refl, rews, dones, truncs, values, 0.99, 0.95, ret_std, clip)
    got = R.compute_gae(rews, dones, truncs, values, 0.99, 0.95, ret_std, clip)
    for g, w, name in zip(got, want, ("advantages", "targets", "returns")):
        assert np.abs(g - w).max() <= 1e-5 * max(1.0, np.abs(w).max()), name   # stated tolerance of the path: 1e-4 on returns / advantages


def test_policy_oracle_equals_reference_discrete_policy(refl):
    """DiscretePolicy::GetActionProbs / GetAction(deterministic) / GetBackpropData and ValueEstimator::Forward (the real classes,
    our parameters loaded in state-dict order) against the numpy restatement."""
    rs = np.random.RandomState(4)
    pol, pol_shapes = _net(rs, A); cri, cri_shapes = _net(rs, 1)
    obs = (rs.randn(64, D) * 1.5).astype(np.float32); acts = rs.randint(0, A, 64).astype(np.int32)
    for temp in (1.0, 1.7):
        probs, am, lp, ent, *_ = ref_policy(refl, pol, obs, acts, temp)
        logits = R.mlp_forward(pol, pol_shapes, obs)[0]
        mine = R.policy_probs(logits, temp)
        assert np.abs(mine - probs).max() < 2e-6
        assert (np.argmax(mine, 1) == am).all()
        assert np.abs(np.log(mine[np.arange(64), acts]) - lp).max() < 2e-5
        assert abs(float((-(np.log(mine) * mine).sum(1)).mean()) - ent) < 2e-5
    vals = np.zeros(64, np.float32)
    hid = np.array(HID, np.int32)
    refl.refl_value(_p(hid), len(HID), D, _p(cri), _p(obs), 64, _p(vals))
    assert np.abs(R.mlp_forward(cri, cri_shapes, obs)[0].reshape(-1) - vals).max() < 1e-5


def test_sampler_equals_reference_get_action(refl):
    """GetAction(obs, deterministic = false) = torch::multinomial(probs, 1, true): with the Exp(1) tape of the same generator state the
    sampler of this repo, argmax(p / q), picks the same action indices bit for bit, and log p[a] agrees."""
    rs = np.random.RandomState(9)
    pol, _ = _net(rs, A)
    obs = (rs.randn(512, D) * 2).astype(np.float32); acts = np.zeros(512, np.int32)
    probs, _, _, _, sampled, slogp, tape = ref_policy(refl, pol, obs, acts, 1.0, seed=1234)
    a, logp = R.sample_actions(probs, tape)
    margin = R.top2_margin(probs, tape)
    differ = a != sampled
    assert not (differ & (margin > 1e-6)).any() and differ.mean() < 0.01     # only exact-tie rows may differ
    ok = ~differ
    assert np.abs(logp[ok] - slogp[ok]).max() < 1e-6
    assert len(set(sampled.tolist())) > 30                                      # it really sampled


@pytest.mark.gpu
def test_hip_learner_kernels_against_reference_code(refl):
    """HIP kernels (fp32 mode) straight against the reference classes: probabilities, deterministic actions, sampled actions on the
    reference's own Exp(1) tape, values, and GAE on a [T][n] batch laid out as the reference concatenates it."""
    torch = pytest.importorskip("torch")
    from rlgymppo_cpp_amd.ppo import PPOCore
    rs = np.random.RandomState(21)
    pol, _ = _net(rs, A); cri, _ = _net(rs, 1)
    ppo = PPOCore(D, A, tuple(HID), tuple(HID), use_bf16=False, max_rows=4096, seed=3)
    ppo.set_params(pol, 0); ppo.set_params(cri, 1)
    rows = 1024
    obs = (rs.randn(rows, D) * 1.5).astype(np.float32); acts0 = np.zeros(rows, np.int32)
    probs, am, _, _, sampled, slogp, tape = ref_policy(refl, pol, obs, acts0, 1.0, seed=77)
    dobs = torch.from_numpy(obs).cuda()
    assert np.abs(ppo.probs(dobs).cpu().numpy() - probs).max() < 2e-6
    a = torch.empty(rows, dtype=torch.int32, device="cuda"); lp = torch.empty(rows, device="cuda")
    ppo.act(dobs, a, lp, deterministic=True)
    margin_det = np.sort(probs, 1)
    assert ((a.cpu().numpy() == am) | (margin_det[:, -1] - margin_det[:, -2] < 1e-6)).all()
    ppo.act(dobs, a, lp, noise=torch.from_numpy(tape).cuda())
    differ = a.cpu().numpy() != sampled
    assert not (differ & (R.top2_margin(probs, tape) > 1e-5)).any()             # action indices bit-exact on the reference's noise
    assert np.abs(lp.cpu().numpy()[~differ] - slogp[~differ]).max() < 1e-5
    vals = np.zeros(rows, np.float32); hid = np.array(HID, np.int32)
    refl.refl_value(_p(hid), len(HID), D, _p(cri), _p(obs), rows, _p(vals))
    dv = torch.empty(rows, device="cuda"); ppo.value(dobs, dv)
    assert np.abs(dv.cpu().numpy() - vals).max() < 1e-5
    # GAE: T steps of n agents; the reference sees the agent-major concatenation plus ONE extra value (the state after the last row)
    T, n = 24, 16
    rews = (rs.randn(T, n) * 2).astype(np.float32); dones = (rs.rand(T, n) < 0.05).astype(np.float32)
    truncs = np.zeros((T, n), np.float32); truncs[T - 1] = 1 - dones[T - 1]       # the collector's mark (ThreadAgentManager.cpp:55)
    values = rs.randn(T + 1, n).astype(np.float32)
    flat = lambda x: np.ascontiguousarray(x.T.reshape(-1))                          # trajectory after trajectory
    vcat = np.concatenate([flat(values[:T]), values[T, n - 1:n]]).astype(np.float32)
    want = ref_gae(refl, flat(rews), flat(dones), flat(truncs), vcat, 0.99, 0.95, 1.7, 5.0)
    adv, tgt, ret = ppo.gae(torch.from_numpy(rews).cuda(), torch.from_numpy(dones).cuda(), torch.from_numpy(truncs).cuda(), torch.from_numpy(values).cuda(),
                            0.99, 0.95, 1.7, 5.0, 0)
    for g, w, name in ((adv, want[0], "advantages"), (tgt, want[1], "targets"), (ret, want[2], "returns")):
        assert np.abs(flat(g.cpu().numpy()) - w).max() < 1e-4, name                 # north_star: within 1e-4 on returns / advantages


@pytest.mark.parametrize("max_rows", [60, 100, 150, 37, 400])
def test_experience_fifo_equals_reference_experience_buffer(refl, max_rows):
    """The REAL ExperienceBuffer (SubmitExperience + GetAllBatchesShuffled with its own std::default_random_engine): the rows it hands
    out, batch by batch, after every submission, against (a) the numpy oracle's literal buffer and (b) the library's slot bookkeeping
    (rlgpu_expbuf_*) -- same rows alive, same shuffle, same batches, remainder dropped."""
    from rlgymppo_cpp_amd.learner import ExperienceFifo, Shuffler
    T, N, seed, n_submits, batch = 5, 12, 321, 7, 16
    B = T * N
    ids = np.zeros(n_submits * (max_rows + B), np.int64); counts = np.zeros(n_submits, np.int32)
    refl.refl_expbuf_run.restype = C.c_int64
    n = refl.refl_expbuf_run(C.c_int64(max_rows), seed, n_submits, B, C.c_int64(batch), _p(ids), C.c_int64(ids.size), _p(counts))
    assert 0 < n <= ids.size
    ids = ids[:n]
    ref = R.ExperienceBufferRef(max_rows, seed)
    fifo, shuf = ExperienceFifo(max_rows, T, N), Shuffler(seed)
    slot_of, at = {}, 0
    for s in range(n_submits):
        ref.submit(s * B + np.arange(B))
        slot = fifo.submit()
        slot_of = {k: v for k, v in slot_of.items() if v != slot}; slot_of[s] = slot
        want = ref.all_batches_shuffled(batch)
        assert len(want) == counts[s] == min(max_rows, (s + 1) * B) // batch
        rows = np.empty(fifo.size(), np.int32)
        fifo.shuffled_rows(shuf, rows)
        for b, w in enumerate(want):
            got = ids[at:at + batch]; at += batch
            assert (got == w).all()                                                  # numpy oracle == the real buffer
            k, a = got // B, got % B                                                 # reference row id -> (submission, agent-major index)
            phys = np.array([slot_of[int(i)] for i in k]) * B + (a % T) * N + a // T
            assert (rows[b * batch:(b + 1) * batch] == phys).all()                   # the library's device rows name the same experience
    assert at == n


def test_welford_equals_reference_header(refl):
    """WelfordRunningStat.h: mean / raw sum of squared deviations / count / GetSTD after chunked increments, against the Python host's
    statistic (what RUNNING_STATS.json stores) and the oracle's."""
    from rlgymppo_cpp_amd.learner import WelfordRunningStat
    rs = np.random.RandomState(2)
    xs = (rs.randn(1000) * 3 + 0.5).astype(np.float32)
    mean, m2, cnt, std = C.c_double(), C.c_double(), C.c_int64(), C.c_float()
    refl.refl_welford(_p(xs), 1000, 150, C.byref(mean), C.byref(m2), C.byref(cnt), C.byref(std))
    w = WelfordRunningStat()
    for i in range(0, 1000, 150):
        part = xs[i:i + 150].tolist()
        w.increment(part, len(part))
    assert w.count == cnt.value == 1000 and abs(w.mean - mean.value) < 1e-12 and abs(w.m2 - m2.value) < 1e-9 * m2.value
    assert abs(w.get_std() - std.value) < 1e-6
    assert w.to_json()["var"][0] == w.m2                                          # the file stores the raw accumulator, like the reference
    o = R.Welford(); o.increment(xs, 1000)
    assert abs(o.std() - std.value) < 1e-6
    one = (C.c_float * 1)(3.0)
    refl.refl_welford(one, 1, 150, C.byref(mean), C.byref(m2), C.byref(cnt), C.byref(std))
    assert std.value == 1.0 and WelfordRunningStat().get_std() == 1.0               # fewer than two samples: std 1 (WelfordRunningStat.h:70-72)


@pytest.mark.gpu
@pytest.mark.parametrize("deterministic", [False, True])
def test_add_new_experience_and_learn_composition_vs_reference_pipeline(refl, deterministic):
    """the COMPOSITION of one training iteration, not its pieces.  The Learner (fp32 mode) collects on the GPU;
    the collected rows then go through a pipeline assembled from the reference's own code (libref_learner.so: ValueEstimator, ComputeGAE,
    DiscretePolicy) and the numpy PPO oracle, in the reference's order (Learner.cpp:608-703, PPOLearner.cpp:67-349):
      B + 1 value rows (every state of the agent-major concatenation + the single final nextStates[B - 1]) | retStd read BEFORE this
      iteration's returns enter the statistic | ComputeGAE on the concatenation with the collector's truncation marks | the three report
      averages | returnStats.Increment with the FIRST min(150, B) returns of the concatenation | one epoch of two accumulated minibatches,
      clip-by-norm per network, one Adam step.
    Three iterations, so that retStd is 1 (n < 2), then the std of 150 samples, then of 300 -- ten in the deterministic-gradient mode
    (LearnerConfig.deterministicGradients: fixed-order dW / db sums).  Also: the action the policy sampled for every
    collected row is the reference DiscretePolicy's choice on the same probabilities (log-probs within 1e-5)."""
    torch = pytest.importorskip("torch")
    from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
    n_envs, T = 16, 12
    N = n_envs * 2; B = N * T
    cfg = LearnerConfig(numEnvs=n_envs, teamSize=1, timestepsPerIteration=B, expBufferSize=B, randomSeed=9, standardizeReturns=True, maxReturnsPerStatsInc=150, deterministicGradients=deterministic,
                        ppo=PPOLearnerConfig(policyLayerSizes=tuple(HID), criticLayerSizes=tuple(HID), batchSize=B, miniBatchSize=B // 2, epochs=1,
                                             policyLR=3e-4, criticLR=3e-4, entCoef=0.01, clipRange=0.2, autocastLearn=False))
    L = Learner(cfg)
    assert L.obs_size == D and L.n_agents == N
    pshapes, cshapes = L.ppo.layer_shapes(0), L.ppo.layer_shapes(1)
    hid = np.array(HID, np.int32)
    stat = R.Welford()
    mom = {0: [np.zeros(L.ppo.num_params(0), np.float32), np.zeros(L.ppo.num_params(0), np.float32)], 1: [np.zeros(L.ppo.num_params(1), np.float32), np.zeros(L.ppo.num_params(1), np.float32)]}
    am = lambda x: np.ascontiguousarray(np.moveaxis(x, 0, 1).reshape((-1,) + x.shape[2:]))      # [T][N]... -> agent-major concatenation (trajectory after trajectory)
    n_iter = 10 if deterministic else 3
    for it in range(n_iter):
        pol, cri = L.ppo.get_params(0).copy(), L.ppo.get_params(1).copy()
        L.collect(); L.ppo.sync(); torch.cuda.synchronize()
        obs = L.obs_buf.cpu().numpy().reshape(T + 1, N, D); acts = L.act_buf.cpu().numpy().reshape(T, N); logp = L.logp_buf.cpu().numpy().reshape(T, N)
        rew = L.rew_buf.cpu().numpy().reshape(T, N); done = L.done_buf.cpu().numpy().reshape(T, N).astype(np.float32)
        L.add_new_experience(); torch.cuda.synchronize()
        # -- the reference pipeline on the same rows
        states = am(obs[:T]); final_next = obs[T, N - 1][None]
        vin = np.ascontiguousarray(np.concatenate([states, final_next]), np.float32)
        vals = np.zeros(B + 1, np.float32); refl.refl_value(_p(hid), len(HID), D, _p(cri), _p(vin), B + 1, _p(vals))
        ret_std = stat.std()                                                          # before this iteration's update (Learner.cpp:651)
        trunc = np.zeros((T, N), np.float32); trunc[T - 1] = 1 - done[T - 1]          # ThreadAgentManager.cpp:55
        adv_r, tgt_r, ret_r = ref_gae(refl, am(rew), am(done), am(trunc), vals, cfg.gaeGamma, cfg.gaeLambda, ret_std, cfg.rewardClipRange)
        adv = am(L.adv.cpu().numpy().reshape(T, N)); tgt = am(L.tgt.cpu().numpy().reshape(T, N)); ret = am(L.ret.cpu().numpy().reshape(T, N))
        assert np.abs(adv - adv_r).max() < 1e-4 and np.abs(ret - ret_r).max() < 1e-4 and np.abs(tgt - tgt_r).max() < 1e-4, f"iteration {it}: GAE composition"
        rep = L.finish_report()
        assert abs(rep["Avg Return"] - float(np.abs(ret_r).mean()) / ret_std) < 1e-4 * max(1.0, float(np.abs(ret_r).mean()) / ret_std)
        assert abs(rep["Avg Advantage"] - float(np.abs(adv_r).mean())) < 1e-4 and abs(rep["Avg Val Target"] - float(np.abs(tgt_r).mean())) < 1e-4
        stat.increment(ret_r, min(cfg.maxReturnsPerStatsInc, B))                      # the first 150 returns of the concatenation (Learner.cpp:679-682)
        L._flush_returns()
        assert L.return_stats.count == stat.count == 150 * (it + 1)
        assert abs(L.return_stats.mean - stat.mean) < 1e-4 * max(1.0, abs(stat.mean)) and abs(L.return_stats.get_std() - stat.std()) < 1e-4 * max(1.0, stat.std())
        # -- the sampled actions are the reference policy's on these observations (its probabilities, the Learner's own recorded choice)
        probs, _, lp_ref, _, _, _, _ = ref_policy(refl, pol, states, am(acts).astype(np.int32), 1.0, seed=1)
        assert np.abs(am(logp) - lp_ref).max() < 1e-5, f"iteration {it}: log-probs of the collected actions"
        # -- one epoch = one optimizer step over the two accumulated minibatches (PPOLearner.cpp:127-288)
        L.learn(); L.ppo.sync(); torch.cuda.synchronize()
        gp = np.zeros_like(pol); gc = np.zeros_like(cri)
        for h in range(2):     # any split into two halves gives the same sums: the batch gradient is the mean over all B rows
            sl = slice(h * B // 2, (h + 1) * B // 2)
            g0, g1, _ = R.ppo_minibatch_grads(pol, pshapes, cri, cshapes, states[sl], am(acts)[sl], am(logp)[sl], adv_r[sl], tgt_r[sl], clip=0.2, ent_coef=0.01, ratio_scale=0.5)
            gp += g0; gc += g1
        want_p, mom[0][0], mom[0][1] = R.clip_adam_step(pol, gp, mom[0][0], mom[0][1], it + 1, 3e-4)
        want_c, mom[1][0], mom[1][1] = R.clip_adam_step(cri, gc, mom[1][0], mom[1][1], it + 1, 3e-4)
        dp, dc = np.abs(L.ppo.get_params(0) - want_p).max(), np.abs(L.ppo.get_params(1) - want_c).max()
        # (Adam's first steps are lr * g / (|g| + 1e-8) per entry: where a gradient entry is of the order of 1e-8 .. 1e-7 the quotient moves by
        # per cents when fp32 summation order moves the entry by 1e-9 -- 1e-5 is 3 % of one 3e-4 step, for a handful of such entries)
        assert dp < 1e-5 and dc < 1e-5, f"iteration {it}: parameters after the optimizer step differ by {dp:.3g} (policy) / {dc:.3g} (critic); step sizes {np.abs(want_p - pol).max():.3g} / {np.abs(want_c - cri).max():.3g}"
    assert L.cumulative_model_updates == n_iter
    if deterministic:   # the mode's own promise: the same seed gives the same bits (a second learner, the same ten iterations)
        L2 = Learner(cfg)
        for it in range(n_iter):
            L2.collect(); L2.add_new_experience(); L2.finish_report(); L2._flush_returns(); L2.learn()
        L2.ppo.sync(); torch.cuda.synchronize()
        assert np.array_equal(L2.ppo.get_params(2), L.ppo.get_params(2)), "two deterministic-mode runs of one seed differ"
