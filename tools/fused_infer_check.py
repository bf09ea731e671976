# fused (k_mlp_infer) vs unfused (staging + per-layer GEMMs + head) policy / critic inference: same results? time per call?
import os, sys, subprocess, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from rlgymppo_cpp_amd.ppo import PPOCore
def run(fused):
    if fused: os.environ.pop("RLGPU_NO_FUSED_INFER", None)
    else: os.environ["RLGPU_NO_FUSED_INFER"] = "1"
    ppo = PPOCore(89, 90, (256, 256, 256), (256, 256, 256), use_bf16=True, max_rows=65536, seed=7)
    g = torch.Generator(device="cpu").manual_seed(1)
    obs = (torch.randn(8192 + 17, 89, generator=g) * 1.5).cuda()
    a = torch.empty(obs.shape[0], dtype=torch.int32, device="cuda"); lp = torch.empty(obs.shape[0], device="cuda")
    ppo.act(obs, a, lp)
    p = ppo.probs(obs[:100])
    v = torch.empty(obs.shape[0], device="cuda"); ppo.value(obs, v)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    res = (a.cpu().numpy(), lp.cpu().numpy(), p.cpu().numpy(), v.cpu().numpy())
    o8, a8, l8 = obs[:8192], a[:8192].clone(), lp[:8192].clone()
    big = torch.empty(1 << 26, device="cuda")
    for _ in range(3): ppo.act(o8, a8, l8)
    torch.cuda.synchronize()
    big.zero_()            # ~100 us of GPU work queued first, so the launches below are all enqueued before the first one starts
    big.zero_()
    t0.record()
    for _ in range(20): ppo.act(o8, a8, l8)
    t1.record(); torch.cuda.synchronize()
    if fused and os.environ.get("RLGPU_FUSED_STAMPS"):
        st = ppo.grad_tensor()[:32].view(torch.int64).cpu().numpy()
        d = np.diff(st[:8])
        print("phase cycles (stage, layers..., head):", d.tolist(), "total", int(st[6 if False else len(d)] - st[0]))
    return res + (t0.elapsed_time(t1) / 20,)
B = run(False)
for R in ("32",):
    os.environ["RLGPU_FUSED_STAMPS"] = "1"
    A = run(True)
    print("act us (host-paced loop) fused %.1f unfused %.1f" % (A[4] * 1e3, B[4] * 1e3))
    print("  actions equal:", (A[0] == B[0]).mean(), "logp maxdiff", np.abs(A[1] - B[1]).max(), "probs maxdiff", np.abs(A[2] - B[2]).max(), "values maxdiff", np.abs(A[3] - B[3]).max())
