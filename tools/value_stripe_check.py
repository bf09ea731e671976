"""k_value_stripe (ppo_fused.h) against k_mlp_infer (RLGPU_NO_VALUE_STRIPE=1): values of the flagship critic over a ragged row count, each path in a
   process of its own.  usage: value_stripe_check.py [rows] [obs_size]"""
import os, sys, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 70001
D = int(sys.argv[2]) if len(sys.argv) > 2 else 89
code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
from rlgymppo_cpp_amd.ppo import PPOCore
dev = torch.device('cuda', 0); rows = %d; D = %d
rng = np.random.RandomState(7)
core = PPOCore(D, 90, (256, 256, 256), (256, 256, 256), use_bf16=True, seed=3, max_rows=rows)
obs = torch.from_numpy((rng.randn(rows, D) * 0.7).astype(np.float32)).to(dev)
v = core.value(obs); core.sync()
np.save(sys.argv[1], v.cpu().numpy())
core.check_redzones()   # (RLGPU_REDZONE, set below)
""" % (ROOT, rows, D)
outs = []
with tempfile.TemporaryDirectory() as tmp:
    for stripe in (False, True):
        out = os.path.join(tmp, "v%d.npy" % stripe)
        env = dict(os.environ); env.pop("RLGPU_NO_VALUE_STRIPE", None); env["RLGPU_REDZONE"] = "65536"
        if not stripe: env["RLGPU_NO_VALUE_STRIPE"] = "1"
        r = subprocess.run([sys.executable, "-c", code, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        assert r.returncode == 0, r.stdout[-3000:]
        outs.append(np.load(out))
a, b = outs
err = np.abs(a - b).max(); big = np.abs(a).max()
print(f"{rows} rows, obs {D}: largest |value| {big:.4f}, max |stripe - k_mlp_infer| {err:.3e}")
ok = np.isfinite(b).all() and err <= 2e-3 * max(big, 1.0)
print("OK" if ok else "MISMATCH"); sys.exit(0 if ok else 1)
