"""Learner — the host-side orchestration of one training iteration, mirroring RLGPC::Learner
(RLGymPPO_CPP/src/public/RLGymPPO_CPP/Learner.cpp:436-703) on top of the C-ABI hot path:

    collect (policy act + batched env step, all on device)     <- ThreadAgentManager::CollectTimesteps  (Learner.cpp:460)
    AddNewExperience: value preds, GAE, return stats, buffer   <- Learner.cpp:608-703
    PPOLearner::Learn: epochs x batches x minibatches           <- PPOLearner.cpp:67-349
    report / checkpoint                                         <- Learner.cpp:379-434, 245-376

Config field names and defaults are the reference's (LearnerConfig.h:14-80, PPOLearnerConfig.h:6-32); new fields
are appended only.  This module owns no math: every tensor op on the hot path is a HIP kernel behind include/rlgpu.h;
torch provides buffers and streams; the one gradient all-reduce per optimizer step is the C-ABI's own (rlgpu_comm_* on RCCL,
parallel.RcclComm) -- torch.distributed appears only in the CPU stand-in of the tests (parallel.GlooComm).
"""
import ctypes as C
import json
import os
import shutil
import threading
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib, parallel
from .env import BatchedEnv
from .ppo import PPOCore


@dataclass
class PPOLearnerConfig:  # PUB/PPO/PPOLearnerConfig.h:6-32
    policyLayerSizes: tuple = (256, 256, 256)
    criticLayerSizes: tuple = (256, 256, 256)
    batchSize: int = 50 * 1000
    epochs: int = 10
    policyLR: float = 3e-4
    criticLR: float = 3e-4
    entCoef: float = 0.005
    clipRange: float = 0.2
    miniBatchSize: int = 0
    autocastLearn: bool = False   # bf16 MFMA operands (the reference's autocast dtype, FrameworkTorch.h:12-16)
    policyTemperature: float = 1.0


@dataclass
class LearnerConfig:  # PUB/LearnerConfig.h:14-80
    numThreads: int = 8
    numGamesPerThread: int = 16
    timestepLimit: int = 0
    expBufferSize: int = 100 * 1000
    timestepsPerIteration: int = 50 * 1000
    standardizeReturns: bool = True
    maxReturnsPerStatsInc: int = 150
    deterministic: bool = False
    deterministicGradients: bool = False   # fixed-order dW / db sums (rlgpu_learner_set_deterministic): a run is reproducible from its seed
    collectionDuringLearn: bool = False   # LearnerConfig.h:46-50: the PPO epochs of iteration k run while iteration k+1 is collected
    ppo: PPOLearnerConfig = field(default_factory=PPOLearnerConfig)
    gaeLambda: float = 0.95
    gaeGamma: float = 0.99
    rewardClipRange: float = 10.0
    checkpointLoadFolder: str = "checkpoints"
    checkpointSaveFolder: str = "checkpoints"
    timestepsPerSave: int = 500 * 1000
    randomSeed: int = 123
    checkpointsToKeep: int = 5
    sendMetrics: bool = False   # JSON lines under metrics/<project>/<run id>.jsonl (the reference defaults to True with its wandb receiver)
    metricsProjectName: str = "rlgymppo-cpp"
    metricsGroupName: str = "unnamed-runs"
    metricsRunName: str = "rlgymppo-cpp-run"
    # ---- appended for the batched device path
    numEnvs: int = 0            # 0 -> numThreads * numGamesPerThread (the reference's env count)
    teamSize: int = 1
    gaeNextValueMode: int = 0   # 0 = reference-faithful cross-trajectory bootstrap, 1 = per-agent bootstrap
    device: int = 0


class WelfordRunningStat:
    """PUB/Util/WelfordRunningStat.h:5-84 (shape 1, double)."""

    def __init__(self):
        self.mean = 0.0; self.m2 = 0.0; self.count = 0

    def increment(self, samples, num):
        for x in samples[:num]:
            x = float(x)
            delta = x - self.mean
            delta_n = delta / (self.count + 1)
            self.mean += delta_n
            self.m2 += delta * delta_n * self.count
            self.count += 1

    def get_std(self):
        if self.count < 2:
            return 1.0
        var = self.m2 / (self.count - 1)
        return 1.0 if var == 0 else float(np.sqrt(var))

    def to_json(self):
        # "var" is the raw sum of squared deviations, as the reference stores its `runningVariance` member (Learner.cpp:198-201)
        return {"mean": [self.mean], "var": [self.m2], "shape": 1, "count": self.count}

    def from_json(self, j):
        self.mean = float(j["mean"][0]); self.count = int(j["count"]); self.m2 = float(j["var"][0])


class Shuffler:
    def __init__(self, seed):
        self.lib = _lib.load()
        self.h = C.c_void_p()
        assert self.lib.rlgpu_shuffler_create(C.byref(self.h), seed & 0xffffffff) == 0

    def next(self, n):
        out = np.empty(n, np.int64)
        assert self.lib.rlgpu_shuffler_next(self.h, n, out.ctypes.data) == 0
        return out

    def next_rows(self, T, n_agents, out=None):
        """The same draw as next(T * n_agents), already mapped from logical agent-major order to time-major buffer rows (int32)."""
        if out is None:
            out = np.empty(T * n_agents, np.int32)
        assert self.lib.rlgpu_shuffler_next_rows(self.h, T, n_agents, out.ctypes.data) == 0
        return out

    def __del__(self):
        try:
            self.lib.rlgpu_shuffler_destroy(self.h)
        except Exception:
            pass


class MetricSender:
    """MetricSender (PUB/Util/MetricSender.cpp:8-45) without the embedded interpreter: the same JSON-lines file the C++ host writes
    (include/RLGymPPO_CPP/Util/MetricSender.h); tools/metric_receiver.py forwards it to wandb."""

    def __init__(self, project, group, name, run_id=""):
        self.run_id = run_id or "".join(np.random.RandomState(int(time.time() * 1e6) % (2 ** 31)).choice(list("0123456789abcdefghijklmnopqrstuvwxyz"), 8))
        folder = os.path.join(os.environ.get("RLGPU_METRICS_DIR") or "metrics", project or "default")
        os.makedirs(folder, exist_ok=True)
        self.path = os.path.join(folder, self.run_id + ".jsonl")
        if not os.path.exists(self.path):
            with open(self.path, "a") as f:
                f.write(json.dumps({"_run": {"project": project, "group": group, "name": name, "id": self.run_id}}) + "\n")

    def send(self, report):
        with open(self.path, "a") as f:
            f.write(json.dumps({k: (float(v) if np.isfinite(v) else None) for k, v in report.items()}) + "\n")


class ExperienceFifo:
    """ExperienceBuffer's FIFO bookkeeping (ExperienceBuffer.cpp:17-68): which device slot holds which iteration, and which of its rows remain."""

    def __init__(self, max_rows, T, n_agents):
        self.lib = _lib.load()
        self.h = C.c_void_p()
        if self.lib.rlgpu_expbuf_create(C.byref(self.h), max_rows, T, n_agents) != 0:
            raise ValueError("RG FATAL ERROR: ExperienceBuffer: bad size")
        self.num_slots = self.lib.rlgpu_expbuf_num_slots(self.h)

    def submit(self):
        slot = C.c_int()
        assert self.lib.rlgpu_expbuf_submit(self.h, C.byref(slot)) == 0
        return slot.value

    def size(self):
        return int(self.lib.rlgpu_expbuf_size(self.h))

    def shuffled_rows(self, shuffler, out):
        """GetAllBatchesShuffled's permutation of the whole FIFO (ExperienceBuffer.cpp:104-126) as device rows; returns how many."""
        n = self.size()
        assert self.lib.rlgpu_expbuf_shuffled_rows(self.h, shuffler.h, out.ctypes.data) == 0
        return n

    def __del__(self):
        try:
            self.lib.rlgpu_expbuf_destroy(self.h)
        except Exception:
            pass


class Learner:
    def __init__(self, cfg: LearnerConfig, gym_cfg=None, mesh="procedural", rank=0, world_size=1, comm=None):
        self.cfg = cfg
        # the exchange object (parallel.py): RCCL through the C-ABI on a multi-GPU launch (the launcher's WORLD_SIZE > 1), a no-op alone.
        # A world of several ranks WITHOUT an exchange would train diverging replicas on disjoint shards: refused.
        if comm is None:
            if world_size > 1:
                raise ValueError("Learner(world_size > 1) needs the ranks' communicator: pass comm=parallel.make_comm() (RCCL from the launcher's "
                                 "environment) -- without it no gradient would ever be exchanged")
            if parallel.env_ranks()[2] > 1:
                # the communicator binds RCCL to LOCAL_RANK's device: the env batch, the learner and every buffer must live there too
                comm = parallel.make_comm()
                local = comm.device if getattr(comm, "device", None) is not None else int(os.environ.get("LOCAL_RANK", str(comm.rank)))
                if cfg.device != local:
                    cfg.device = local
            else:
                comm = parallel.SoloComm()
        elif world_size > 1 and hasattr(comm, "device") and comm.device is not None and comm.device != cfg.device:
            raise ValueError(f"Learner: the communicator is bound to device {comm.device} but cfg.device is {cfg.device}: one process per GPU, everything on its device")
        self.comm = comm
        rank, world_size = comm.rank, comm.world
        self.rank, self.world = rank, world_size
        n_envs = cfg.numEnvs or cfg.numThreads * cfg.numGamesPerThread
        self.gym_cfg = gym_cfg if gym_cfg is not None else _lib.default_gym_config()
        # every rank owns its own env shard and RNG streams: seed = randomSeed + 1000 * rank
        self.gym_cfg.seed_lo = parallel.shard_seed(cfg.randomSeed, rank)
        self.env = BatchedEnv(n_envs, cfg.teamSize, self.gym_cfg, cfg.device, mesh)
        self.env_stream_epoch = 0
        self.dev = torch.device("cuda", cfg.device)
        self.n_agents = self.env.n_agents
        self.obs_size, self.n_actions = self.env.obs_size, self.env.n_actions
        # steps per iteration: collect at least timestepsPerIteration agent steps (the reference returns >= requested, Q6)
        self.T = max(1, -(-cfg.timestepsPerIteration // self.n_agents))
        self.B = self.T * self.n_agents
        p = cfg.ppo
        self.batch_size = min(p.batchSize, self.B) if p.batchSize > 0 else self.B
        self.mini = p.miniBatchSize if p.miniBatchSize > 0 else self.batch_size
        if self.batch_size % self.mini != 0:
            raise ValueError("RG FATAL ERROR: PPOLearner: batchSize must be a multiple of miniBatchSize")  # PPOLearner.cpp:35-36
        max_rows = max(self.mini, self.n_agents)
        # identical parameters on every rank: the init seed does not depend on the rank
        self.ppo = PPOCore(self.obs_size, self.n_actions, p.policyLayerSizes, p.criticLayerSizes, p.policyLR, p.criticLR, p.entCoef, p.clipRange,
                           p.policyTemperature, p.autocastLearn, cfg.randomSeed, max_rows, cfg.device)
        # ... but every rank explores with its own noise: the sampler is keyed on the rank (identical observations on two ranks draw different actions)
        self.ppo.set_sampler(rank, 0)
        if cfg.deterministicGradients:
            self.ppo.set_deterministic(True)
        T, N, D = self.T, self.n_agents, self.obs_size
        f = dict(dtype=torch.float32, device=self.dev)
        self.obs_buf = torch.empty((T + 1, N, D), **f)     # states; row T = the state after the last step
        self.act_buf = torch.empty((T, N), dtype=torch.int32, device=self.dev)
        self.logp_buf = torch.empty((T, N), **f)
        self.rew_buf = torch.empty((T, N), **f)
        self.done_buf = torch.empty((T, N), dtype=torch.int32, device=self.dev)
        self.val_buf = torch.empty((T + 1, N), **f)
        self.trunc_buf = torch.zeros((T, N), **f)
        self.metrics = torch.zeros(8, **f)
        self.return_stats = WelfordRunningStat()
        self.shuffler = Shuffler(cfg.randomSeed)
        # the experience FIFO (ExperienceBuffer.h): `num_slots` iterations stay resident in HBM, the library tracks which rows are alive
        self.fifo = ExperienceFifo(cfg.expBufferSize, T, N)
        S = self.fifo.num_slots
        self.ex_obs = torch.empty((S * self.B, D), **f)
        self.ex_act = torch.empty(S * self.B, dtype=torch.int32, device=self.dev)
        self.ex_logp = torch.empty(S * self.B, **f)
        self.ex_adv = torch.empty(S * self.B, **f)
        self.ex_tgt = torch.empty(S * self.B, **f)
        self._pending_slot = None
        cap = min(cfg.expBufferSize, S * self.B)
        self._rows_host = [torch.empty(cap, dtype=torch.int32).pin_memory() for _ in range(2)]   # double-buffered pinned staging of the shuffle
        self._rows_dev = torch.empty(cap, dtype=torch.int32, device=self.dev)
        self._rows_flip = 0
        self._rows_ev = [None, None]
        self._next_rows = None
        self.total_timesteps = 0
        self.total_epochs = 0
        self.cumulative_model_updates = 0
        self.ts_since_save = 0
        self.report = {}
        self.iteration_callback = None
        self.run_id = ""
        self._ret_host = torch.empty(max(1, cfg.maxReturnsPerStatsInc), dtype=torch.float32).pin_memory()
        self._ret_pending = None
        self._rep_dev = None
        self._n_mb = 0
        # collectionDuringLearn: the PPO epochs go to their own HIP stream and the next collection does not wait for them.  Like the
        # reference's agent threads (ThreadAgent.cpp:72-103) the collector then reads whatever weights are there, mid-update included.
        self.s_collect = torch.cuda.current_stream(self.dev)
        # Only sound when inference keeps off the learner's activation scratch (the fused inference kernel: bf16 mode, nets that fit its
        # LDS); otherwise collection pauses during learning as with collectionDuringLearn = False.
        overlap = bool(cfg.collectionDuringLearn) and bool(self.ppo.lib.rlgpu_learner_inference_is_standalone(self.ppo.h))
        self.s_learn = torch.cuda.Stream(self.dev) if overlap else None
        self._learn_done = None
        if self.s_learn is not None and os.environ.get("RLGPU_COLLECT_SIDE_STREAM"):
            self.s_collect = torch.cuda.Stream(self.dev)      # experiment: collection off the null stream too
            self.env.set_stream(self.s_collect); self.ppo.set_stream(self.s_collect)
        self.metric_sender = None   # created by run() after a possible load(), so that a loaded run id continues (Learner.cpp:149-155)
        self.env.reset(True, self.obs_buf[0])
        self._first = True
        # one launch per collection phase: a wavefront's in-kernel inference serves its own envs (1v1: 4, 2v2: 2, 3v3: 2 per wavefront)
        self._fused_collect = not os.environ.get("RLGPU_NO_FUSED_COLLECT") and cfg.teamSize <= int(os.environ.get("RLGPU_FUSED_MAX_TEAM", "3"))

    # ---- collection ------------------------------------------------------------------------------------------------
    def collect(self):
        """T gym steps of every env with on-device policy inference (ThreadAgent::_RunFunc, ThreadAgent.cpp:58-163)."""
        if not self._first:
            self.obs_buf[0].copy_(self.obs_buf[self.T])
        self._first = False
        # the whole phase in one launch when the policy fits the in-kernel inference (rlgpu_collect), else T x (act, step)
        if self._fused_collect:
            self._fused_collect = self.env.collect(self.ppo, self.T, self.obs_buf, self.act_buf, self.logp_buf, self.rew_buf, self.done_buf,
                                                   deterministic=self.cfg.deterministic)
        if not self._fused_collect:
            for t in range(self.T):
                self.ppo.act(self.obs_buf[t], self.act_buf[t], self.logp_buf[t], deterministic=self.cfg.deterministic)
                self.env.step(self.act_buf[t], self.obs_buf[t + 1], self.rew_buf[t], self.done_buf[t])
        self.total_timesteps += self.B * self.world

    # ---- AddNewExperience (Learner.cpp:608-703) -----------------------------------------------------------------------
    def add_new_experience(self):
        T, N = self.T, self.n_agents
        rows = (T + 1) * N
        flat_obs = self.obs_buf.view(rows, self.obs_size)
        flat_val = self.val_buf.view(rows)
        step = self.ppo.max_rows
        for s in range(0, rows, step):   # minibatched value predictions (Learner.cpp:628-640)
            e = min(rows, s + step)
            self.ppo.value(flat_obs[s:e], flat_val[s:e])
        self._flush_returns()   # the previous iteration's returns enter the statistic now: they were copied out without stopping the GPU
        ret_std = self.return_stats.get_std() if self.cfg.standardizeReturns else 1.0   # read BEFORE this iteration's update (Q2)
        dones_f = self.done_buf.to(torch.float32)
        # CollectTimesteps marks the last step of every player trajectory truncated unless done (ThreadAgentManager.cpp:55)
        self.trunc_buf.zero_()
        self.trunc_buf[T - 1] = 1.0 - dones_f[T - 1]
        adv, tgt, ret = self.ppo.gae(self.rew_buf, dones_f, self.trunc_buf, self.val_buf, self.cfg.gaeGamma, self.cfg.gaeLambda, ret_std,
                                     self.cfg.rewardClipRange, self.cfg.gaeNextValueMode)
        self.adv, self.tgt, self.ret = adv, tgt, ret
        if self.cfg.standardizeReturns:
            # the first <=150 returns of the concatenated (agent-major) batch: trajectory 0's first steps (Learner.cpp:679-682)
            # (agent-major order: trajectory 0's T returns, then trajectory 1's, ... until maxReturnsPerStatsInc are taken)
            k = min(self.cfg.maxReturnsPerStatsInc, T * N)
            cols = -(-k // T)
            first = self.comm.share_from_rank0(ret[:, :cols].t().reshape(-1)[:k])   # rank 0's returns feed the shared statistic
            # no host round trip here (each one idles the GPU for ~0.1-0.2 ms): an asynchronous copy into pinned memory, consumed by
            # _flush_returns() right before the statistic is read again -- the same value enters at the same point of the sequence
            self._ret_host[:k].copy_(first, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
            self._ret_pending = (ev, k)
        # report averages stay on the device until a report is asked for
        self._rep_dev = (ret.abs().mean(), adv.abs().mean(), tgt.abs().mean(), ret_std)
        # ExperienceBuffer::SubmitExperience (Learner.cpp:694-702): this iteration's rows join the FIFO
        if self._learn_done is not None:
            self.s_collect.wait_event(self._learn_done)     # the epochs still running on the learn stream read the slots this may overwrite
        slot = self._pending_slot if self._pending_slot is not None else self.fifo.submit()
        self._pending_slot = None
        r = slice(slot * self.B, (slot + 1) * self.B)
        self.ex_obs[r].copy_(flat_obs[:self.B]); self.ex_act[r].copy_(self.act_buf.view(-1)); self.ex_logp[r].copy_(self.logp_buf.view(-1))
        self.ex_adv[r].copy_(adv.view(-1)); self.ex_tgt[r].copy_(tgt.view(-1))

    def _flush_returns(self):
        if self._ret_pending is not None:
            ev, k = self._ret_pending
            ev.synchronize()
            self.return_stats.increment(self._ret_host[:k].numpy().tolist(), k)
            self._ret_pending = None

    def _start_draw(self):
        """Draw the next permutation of the FIFO into the pinned buffer `_rows_flip` points at, on a worker thread."""
        buf = self._rows_host[self._rows_flip].numpy()
        box = {}

        def work():
            box["n"] = self.fifo.shuffled_rows(self.shuffler, buf)
        t = threading.Thread(target=work, daemon=True)
        t.start()
        self._next_rows = (t, box)

    def _take_draw(self):
        if self._next_rows is None:
            self._start_draw()
        t, box = self._next_rows
        t.join()
        self._next_rows = None
        return box["n"]

    # ---- PPOLearner::Learn (PPOLearner.cpp:67-349) ----------------------------------------------------------------
    def learn(self):
        if self.s_learn is None:
            return self._learn_epochs()
        ready = torch.cuda.Event(); ready.record(self.s_collect)    # this iteration's rows are in their FIFO slot
        self.s_learn.wait_event(ready)
        self.ppo.set_stream(self.s_learn)
        try:
            with torch.cuda.stream(self.s_learn):
                n = self._learn_epochs()
                self._learn_done = torch.cuda.Event(); self._learn_done.record(self.s_learn)
        finally:
            self.ppo.set_stream(self.s_collect if self.s_collect.cuda_stream != 0 else None)
        return n

    def _learn_epochs(self):
        p = self.cfg.ppo
        obs, acts, logp, adv, tgt = self.ex_obs, self.ex_act, self.ex_logp, self.ex_adv, self.ex_tgt
        self.metrics.zero_()
        n_mb = 0; n_updates = 0
        for ep in range(p.epochs):
            # shuffled logical (oldest first, agent-major) FIFO indices (ExperienceBuffer.cpp:106-121) as device rows.  A draw does not
            # depend on data, only on the FIFO's bookkeeping: the NEXT one is made by a worker thread (the library call releases the
            # GIL) while this thread launches the epoch and the GPU runs it -- 2-3 ms of std::shuffle per 262 144 rows otherwise sit
            # between the last kernel of one iteration and the first of the next.
            cur = self._take_draw()
            idx = self._rows_dev
            idx[:cur].copy_(self._rows_host[self._rows_flip][:cur], non_blocking=True)
            self._rows_ev[self._rows_flip] = torch.cuda.Event(); self._rows_ev[self._rows_flip].record()
            self._rows_flip ^= 1
            if self._rows_ev[self._rows_flip] is not None:
                self._rows_ev[self._rows_flip].synchronize()   # its last upload has left the pinned buffer
            if ep == p.epochs - 1:
                # the next draw is over the FIFO as it will be after the next submit; the bookkeeping is data-free, so do it now
                self._pending_slot = self.fifo.submit()
            self._start_draw()
            for b in range(cur // self.batch_size):            # remainder rows are skipped (Q5)
                self.ppo.zero_grads()
                base = b * self.batch_size
                for m in range(0, self.batch_size, self.mini):
                    self.ppo.minibatch(obs, acts, logp, adv, tgt, idx[base + m: base + m + self.mini], self.mini, self.mini / self.batch_size, self.metrics)
                    n_mb += 1
                scale = self.comm.allreduce_gradients(self.ppo)   # ONE RCCL all-reduce per optimizer step, on the learner's stream
                self.ppo.clip_adam_step(0.5, scale)
                if self.s_learn is not None:
                    self.ppo.refresh_shadows()                  # the collector's inference reads the bf16 copies: keep them live
                n_updates += 1
        self.total_epochs += p.epochs
        self.cumulative_model_updates += n_updates
        self._n_mb = n_mb
        return n_updates

    def finish_report(self):
        if self._learn_done is not None:
            self._learn_done.synchronize()                      # (a report per iteration serialises the two streams again)
        m = self.metrics.cpu().numpy(); rows = max(1, self._n_mb * self.mini)
        if self._rep_dev is not None:
            r, a, t, ret_std = self._rep_dev
            self.report["Avg Return"] = float(r.item()) / ret_std
            self.report["Avg Advantage"] = float(a.item()); self.report["Avg Val Target"] = float(t.item())
        self.report.update({"Policy Entropy": m[0] / rows, "Mean KL Divergence": m[1] / rows, "SB3 Clip Fraction": m[2] / rows,
                            "Value Function Loss": m[4] / rows, "Cumulative Timesteps": self.total_timesteps,
                            "Cumulative Model Updates": self.cumulative_model_updates, "Total Iterations": self.total_epochs})
        return self.report

    def iteration(self):
        with torch.cuda.stream(self.s_collect):
            self.collect()
            self.add_new_experience()
        self.learn()
        self.ts_since_save += self.B * self.world

    def run(self, iterations=None):
        it = 0
        while True:
            t0 = time.time()
            self.iteration()
            self.ppo.sync()
            self.finish_report()
            dt = time.time() - t0
            self.report["Overall Steps/Second"] = self.B * self.world / dt
            if self.iteration_callback:
                self.iteration_callback(self, self.report)
            if self.cfg.sendMetrics and self.rank == 0:
                if self.metric_sender is None:
                    self.metric_sender = MetricSender(self.cfg.metricsProjectName, self.cfg.metricsGroupName, self.cfg.metricsRunName, self.run_id)
                self.metric_sender.send(self.report)                                       # Learner.cpp:589-590
            if self.ts_since_save > self.cfg.timestepsPerSave and self.rank == 0:
                self.save()
            it += 1
            if iterations is not None and it >= iterations:
                break
            if self.cfg.timestepLimit and self.total_timesteps >= self.cfg.timestepLimit:
                break

    # ---- checkpoints: checkpoints/<timesteps>/{RUNNING_STATS.json, PPO_*.lt} (Learner.cpp:171-376, PPOLearner.cpp:362-502) ----
    MODEL_FILES = ("PPO_POLICY.lt", "PPO_CRITIC.lt", "PPO_POLICY_OPTIM.lt", "PPO_CRITIC_OPTIM.lt")

    def save(self):
        folder = os.path.join(self.cfg.checkpointSaveFolder, str(self.total_timesteps))
        os.makedirs(folder, exist_ok=True)
        self._flush_returns()
        stats = {"cumulative_timesteps": self.total_timesteps, "cumulative_model_updates": self.cumulative_model_updates,
                 "epoch": self.total_epochs, "reward_running_stats": self.return_stats.to_json(),
                 # not in the reference's file (its loader ignores unknown keys): where the action-noise and env-reset streams stand, so a
                 # resumed run continues them instead of replaying the first iterations'
                 "sampler_calls": self.ppo.get_sampler()[1], "env_stream_epoch": self.env_stream_epoch}
        if self.metric_sender is not None:
            stats["run_id"] = self.metric_sender.run_id                                     # Learner.cpp:204-205
        with open(os.path.join(folder, "RUNNING_STATS.json"), "w") as f:
            json.dump(stats, f, indent=4)
        m, v, sp, sc = self.ppo.get_adam_state()
        npol = self.ppo.num_params(0)
        _write_lt(os.path.join(folder, "PPO_POLICY.lt"), self.ppo.get_params(0), self.ppo.layer_shapes(0))
        _write_lt(os.path.join(folder, "PPO_CRITIC.lt"), self.ppo.get_params(1), self.ppo.layer_shapes(1))
        _write_optim(os.path.join(folder, "PPO_POLICY_OPTIM.lt"), m[:npol], v[:npol], sp, self.ppo.layer_shapes(0), self.cfg.ppo.policyLR)
        _write_optim(os.path.join(folder, "PPO_CRITIC_OPTIM.lt"), m[npol:], v[npol:], sc, self.ppo.layer_shapes(1), self.cfg.ppo.criticLR)
        self.ts_since_save = 0
        if self.cfg.checkpointsToKeep > 0:   # prune the lowest-numbered folders (Learner.cpp:256-280)
            base = self.cfg.checkpointSaveFolder
            nums = sorted(int(d) for d in os.listdir(base) if d.isdigit())
            while len(nums) > self.cfg.checkpointsToKeep:
                shutil.rmtree(os.path.join(base, str(nums.pop(0))))
        return folder

    def load(self, folder=None):
        base = self.cfg.checkpointLoadFolder
        if folder is None:
            if not os.path.isdir(base):
                return False
            nums = [int(d) for d in os.listdir(base) if d.isdigit()]
            if not nums:
                return False
            folder = os.path.join(base, str(max(nums)))   # highest-numbered sub-dir (Learner.cpp:291-309)
        with open(os.path.join(folder, "RUNNING_STATS.json")) as f:
            stats = json.load(f)
        self.total_timesteps = int(stats["cumulative_timesteps"]); self.cumulative_model_updates = int(stats["cumulative_model_updates"])
        self.total_epochs = int(stats["epoch"]); self.return_stats.from_json(stats["reward_running_stats"])
        self.run_id = stats.get("run_id", "")                                               # Learner.cpp:238-239
        self.ppo.set_sampler(self.rank, int(stats.get("sampler_calls", self.total_timesteps // max(1, self.n_agents))))
        # the env batch restarts from fresh resets: key them on a new epoch of the RandomState / respawn streams
        self.env_stream_epoch = int(stats.get("env_stream_epoch", 0)) + 1
        self.env.reseed(parallel.shard_seed(self.cfg.randomSeed, self.rank), self.env_stream_epoch)
        self.env.reset(True, self.obs_buf[0])   # the constructor's reset drew epoch 0's first states: draw this epoch's instead
        pol = _read_lt(os.path.join(folder, "PPO_POLICY.lt"), self.ppo.layer_shapes(0))     # size check of every param (PPOLearner.cpp:380-408)
        self.ppo.set_params(pol, 0)
        if os.path.exists(os.path.join(folder, "PPO_CRITIC.lt")):                            # the critic file is optional (PPOLearner.cpp:421-422)
            self.ppo.set_params(_read_lt(os.path.join(folder, "PPO_CRITIC.lt"), self.ppo.layer_shapes(1)), 1)
        n = self.ppo.num_params(2); npol = self.ppo.num_params(0)
        m = np.zeros(n, np.float32); v = np.zeros(n, np.float32); sp = sc = 0
        po = _read_optim(os.path.join(folder, "PPO_POLICY_OPTIM.lt"), self.ppo.layer_shapes(0))
        co = _read_optim(os.path.join(folder, "PPO_CRITIC_OPTIM.lt"), self.ppo.layer_shapes(1))
        if po is not None: m[:npol], v[:npol], sp = po          # missing/empty optimizer file -> reset (PPOLearner.cpp:442-451)
        if co is not None: m[npol:], v[npol:], sc = co
        self.ppo.set_adam_state(m, v, sp, sc)
        self.ppo.set_lr(self.cfg.ppo.policyLR, self.cfg.ppo.criticLR)                        # UpdateLearningRates (Learner.cpp:501)
        return True


# ---- .lt payloads: the reference's own TorchScript zip archives (PPOLearner.cpp:362-477), through rlgpu_lt_* (csrc/lt_archive.cpp) ----
def _dims(shapes):
    return np.array([shapes[0][0][1]] + [ws[0] for ws, _ in shapes], np.int32)


def _lt_fail(lib, what, path):
    raise RuntimeError(f"RG FATAL ERROR: {what} {path}: {lib.rlgpu_lt_last_error().decode()}")


def _write_lt(path, flat, shapes):
    lib, d, a = _lib.load(), _dims(shapes), np.ascontiguousarray(flat, np.float32)
    if lib.rlgpu_lt_write_model(path.encode(), d.ctypes.data, len(shapes), a.ctypes.data) != 0:
        _lt_fail(lib, "failed to save model to", path)


def _read_lt(path, shapes):
    """torch::load + the size check of every parameter (PPOLearner.cpp:372-408)."""
    lib, d = _lib.load(), _dims(shapes)
    out = np.empty(sum(ws[0] * ws[1] + bs[0] for ws, bs in shapes), np.float32)
    if lib.rlgpu_lt_read_model(path.encode(), d.ctypes.data, len(shapes), out.ctypes.data) != 0:
        _lt_fail(lib, "failed to load model from", path)
    return out


def _write_optim(path, m, v, step, shapes, lr):
    lib, d = _lib.load(), _dims(shapes)
    m = np.ascontiguousarray(m, np.float32); v = np.ascontiguousarray(v, np.float32)
    if lib.rlgpu_lt_write_adam(path.encode(), d.ctypes.data, len(shapes), lr, m.ctypes.data, v.ctypes.data, int(step)) != 0:
        _lt_fail(lib, "failed to save optimizer to", path)


def _read_optim(path, shapes):
    """None when the file is missing or empty: the optimizer is reset (PPOLearner.cpp:436-451)."""
    if not os.path.exists(path) or os.path.getsize(path) == 0:
        return None
    lib, d = _lib.load(), _dims(shapes)
    n = sum(ws[0] * ws[1] + bs[0] for ws, bs in shapes)
    m = np.empty(n, np.float32); v = np.empty(n, np.float32); step = C.c_int64()
    if lib.rlgpu_lt_read_adam(path.encode(), d.ctypes.data, len(shapes), m.ctypes.data, v.ctypes.data, C.byref(step)) != 0:
        _lt_fail(lib, "failed to load optimizer from", path)                                # :460-465
    return m, v, step.value
