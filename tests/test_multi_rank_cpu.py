"""The N > 1 path on CPU: two processes, gloo, 127.0.0.1 (envs sharded, ONE gradient all-reduce per optimizer step).

The HIP kernels cannot run here, so the ranks' "learner" is the numpy oracle (oracle/learner_ref.py); what is under test
is the host logic every rank runs around the kernels: rlgymppo_cpp_amd/parallel.py (the Comm interface the Learner calls: shard seeds, gradient all-reduce +
pre-clip scale, shared return statistic, max-over-ranks timing) and the claim DESIGN.md makes about it: sum-all-reduce,
scale by 1/world, THEN clip-by-norm + Adam gives on every rank exactly what one learner gets on the union of the shards.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _problem():
    """A tiny PPO problem: shapes, parameters and a 64-row batch split into two 32-row shards."""
    from oracle import learner_ref as R
    rng = np.random.RandomState(5)
    D, A, H, n = 11, 7, 16, 64
    shapes_p = [((H, D), (H,)), ((A, H), (A,))]; shapes_c = [((H, D), (H,)), ((1, H), (1,))]
    npar = lambda sh: sum(int(np.prod(w)) + int(np.prod(b)) for w, b in sh)
    pol = (rng.randn(npar(shapes_p)) * 0.3).astype(np.float32); cri = (rng.randn(npar(shapes_c)) * 0.3).astype(np.float32)
    obs = rng.randn(n, D).astype(np.float32)
    logits, _ = R.mlp_forward(pol, shapes_p, obs)
    probs = R.policy_probs(logits)
    acts = rng.randint(0, A, n)
    old_logp = (np.log(probs[np.arange(n), acts]) + 0.1 * rng.randn(n)).astype(np.float32)
    adv = rng.randn(n).astype(np.float32); tgt = rng.randn(n).astype(np.float32)
    return dict(D=D, A=A, shapes_p=shapes_p, shapes_c=shapes_c, pol=pol, cri=cri, obs=obs, acts=acts, old_logp=old_logp, adv=adv, tgt=tgt, n=n)


def _grads(P, rows, ratio):
    from oracle import learner_ref as R
    gp, gc, _ = R.ppo_minibatch_grads(P["pol"], P["shapes_p"], P["cri"], P["shapes_c"], P["obs"][rows], P["acts"][rows], P["old_logp"][rows], P["adv"][rows],
                                      P["tgt"][rows], clip=0.2, ent_coef=0.01, ratio_scale=ratio)
    return np.concatenate([gp, gc]).astype(np.float32)


def _step(P, flat_grad, scale):
    """what rlgpu_clip_adam_step(max_norm=0.5, grad_scale=scale) does, per network (oracle restatement)"""
    from oracle import learner_ref as R
    npol = P["pol"].size
    outs = []
    for par, g in ((P["pol"], flat_grad[:npol]), (P["cri"], flat_grad[npol:])):
        p, m, v = R.clip_adam_step(par.copy(), g * scale, np.zeros_like(par), np.zeros_like(par), step=1, lr=1e-2, max_norm=0.5)
        outs.append(p)
    return np.concatenate(outs)


class _OracleCore:
    """Stands where the Learner has its PPOCore: the same two members the Learner's exchange call site touches
    (learner.py: `scale = self.comm.allreduce_gradients(self.ppo)`): a flat [policy | critic] gradient tensor behind grad_tensor()."""
    def __init__(self, flat_grad):
        self.g = torch.from_numpy(flat_grad)

    def grad_tensor(self):
        return self.g


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from rlgymppo_cpp_amd import parallel
    from rlgymppo_cpp_amd.learner import WelfordRunningStat
    comm = parallel.GlooComm("gloo")          # the CPU stand-in of RcclComm: same interface, same call sites
    assert (comm.rank, comm.world) == (rank, world)
    P = _problem()
    half = P["n"] // world
    rows = np.arange(rank * half, (rank + 1) * half)
    # every rank: gradient of ITS shard as one full local batch (ratio = 1), then the one collective -- through the Learner's call
    core = _OracleCore(_grads(P, rows, 1.0))
    scale = comm.allreduce_gradients(core)
    new_params = _step(P, core.grad_tensor().numpy(), scale)
    # shared return statistic: rank-specific returns in, rank 0's out (learner.py add_new_experience)
    ret = torch.arange(10, dtype=torch.float32) + 100.0 * rank
    shared = comm.share_from_rank0(ret)
    ws = WelfordRunningStat(); ws.increment(shared.numpy().tolist(), 10)
    tmax = comm.max_over_ranks(1.0 + rank)
    tsum = comm.sum_over_ranks(1.0 + rank)
    comm.barrier()
    q.put((rank, new_params, scale, shared.numpy(), ws.get_std(), tmax, tsum, parallel.shard_seed(123, rank)))
    comm.close()


@pytest.mark.timeout(300)
def test_two_ranks_allreduce_equals_single_learner():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0
    P = _problem()
    # the single learner on the union of the shards: one batch of n rows (two minibatches of n/2, ratio 1/2 each)
    half = P["n"] // 2
    g_single = _grads(P, np.arange(0, half), 0.5) + _grads(P, np.arange(half, P["n"]), 0.5)
    want = _step(P, g_single, 1.0)
    for rank, new_params, scale, shared, std, tmax, tsum, seed in res:
        assert scale == 0.5
        assert np.abs(new_params - want).max() < 2e-6, np.abs(new_params - want).max()
        assert np.array_equal(shared, np.arange(10, dtype=np.float32))       # rank 0's returns everywhere
        assert tmax == 2.0 and tsum == 3.0
    assert np.array_equal(res[0][1], res[1][1])                               # replicas stay bitwise identical
    assert res[0][4] == res[1][4]
    assert res[0][7] == 123 and res[1][7] == 1123                             # disjoint env RNG streams, rank 0 = single-GPU run


_RDV_CHILD = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from rlgymppo_cpp_amd import _lib
lib = _lib.load()
buf = C.create_string_buffer(512)
assert lib.rlgpu_comm_rendezvous_path(buf, 512) == 0
print("PATH", buf.value.decode())
"""


def _rdv_rank(rank, port, tmp):
    """What one rank of `python -m torch.distributed.run ... bench.py --gpus 2` does up to the rendezvous: bench.py's own child environment,
    then a process (standing in for bench_main) that asks the library where it would meet the others.  Runs in a process of its own, like a rank."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    env = bench.rank_child_env(dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RLGPU_COMM_DIR=tmp))
    r = subprocess.run([sys.executable, "-c", _RDV_CHILD % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=200)
    open(os.path.join(tmp, f"rank{rank}.out"), "w").write(r.stdout)


@pytest.mark.timeout(300)
def test_bench_ranks_meet_at_the_same_rendezvous_file(tmp_path):
    """bench.py starts one bench_main per rank, each from its OWN Python process: the ncclUniqueId file has to be named after something the
    ranks share (the launcher's pid, handed down as RLGPU_COMM_TAG), not after each child's parent -- or rank 1 waits for a file that
    rank 0 never writes.  (The communicator itself needs GPUs; the file name does not.)"""
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rdv_rank, args=(r, port, str(tmp_path))) for r in range(2)]
    for p in procs: p.start()
    for p in procs:
        p.join(250); assert p.exitcode == 0
    paths = []
    for r in range(2):
        out = open(tmp_path / f"rank{r}.out").read()
        line = [ln for ln in out.splitlines() if ln.startswith("PATH ")]
        assert line, out
        paths.append(line[0][5:])
    assert paths[0] == paths[1] and str(port) in paths[0] and paths[0].startswith(str(tmp_path)), paths
