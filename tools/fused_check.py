"""ppo_fused.h against the per-layer bf16 path (RLGPU_NO_FUSED=1): gradients and metrics of one minibatch, each path in a process of its own.
   usage: fused_check.py [rows] [obs_size]"""
import os, sys, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8229
D = int(sys.argv[2]) if len(sys.argv) > 2 else 89
code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
from rlgymppo_cpp_amd.ppo import PPOCore
dev = torch.device('cuda', 0); rows = %d; D, A = %d, 90
rng = np.random.RandomState(5)
core = PPOCore(D, A, (256, 256, 256), (256, 256, 256), use_bf16=True, seed=3, max_rows=rows)
t = lambda x: torch.from_numpy(x).to(dev)
obs = t((rng.randn(rows + 100, D) * 0.7).astype(np.float32)); acts = t(rng.randint(0, A, rows + 100).astype(np.int32))
olp = t((-4.5 + rng.randn(rows + 100) * 0.2).astype(np.float32)); adv = t(rng.randn(rows + 100).astype(np.float32)); tgt = t(rng.randn(rows + 100).astype(np.float32))
idx = t(rng.permutation(rows + 100)[:rows].astype(np.int32)); m = torch.zeros(8, device=dev)
core.zero_grads(); core.minibatch(obs, acts, olp, adv, tgt, idx, rows, 0.25, m); core.sync()
np.savez(sys.argv[1], gp=core.get_grads(0), gc=core.get_grads(1), m=m.cpu().numpy())
core.check_redzones()   # (RLGPU_REDZONE, set below: no kernel of the minibatch wrote past a buffer of the learner)
""" % (ROOT, rows, D)
outs = []
with tempfile.TemporaryDirectory() as tmp:
    for fused in (False, True):
        out = os.path.join(tmp, "f%d.npz" % fused)
        env = dict(os.environ); env.pop("RLGPU_NO_FUSED", None); env["RLGPU_REDZONE"] = "65536"
        if not fused: env["RLGPU_NO_FUSED"] = "1"
        r = subprocess.run([sys.executable, "-c", code, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        assert r.returncode == 0, r.stdout[-3000:]
        outs.append(dict(np.load(out)))
ok = True
for k in ("gp", "gc"):
    a, b = outs[0][k].astype(np.float64), outs[1][k].astype(np.float64)
    big = np.abs(a).max(); err = np.abs(a - b).max()
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    print(f"{k}: largest entry {big:.4e}, max |fused - per-layer| {err:.3e} ({err / big:.2e} of it), cosine {cos:.8f}")
    ok = ok and err <= 1e-2 * big and cos > 0.99999
print("metrics per-layer:", outs[0]["m"][:5]); print("metrics fused    :", outs[1]["m"][:5])
ok = ok and np.allclose(outs[0]["m"], outs[1]["m"], rtol=2e-4, atol=1e-3)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
