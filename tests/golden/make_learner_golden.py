"""Generates tests/golden/learner_golden.npz — golden vectors for the learner-side math.

Run in the BUILD container (CPU torch 2.10 is importable there; it is the same ATen the reference's libtorch path
executes): `python tests/golden/make_learner_golden.py`.  The graph below follows the reference call by call:
  DiscretePolicy::GetActionProbs / GetAction / GetBackpropData  (RLGymPPO_CPP/src/private/RLGymPPO_CPP/PPO/DiscretePolicy.cpp:44-75)
  PPOLearner::Learn loss terms                                   (PPO/PPOLearner.cpp:139-215)
  clip_grad_norm_ + torch::optim::Adam                            (PPO/PPOLearner.cpp:273-288)
  std::shuffle(std::default_random_engine(seed))                  (PPO/ExperienceBuffer.cpp:106-121; via a g++ probe)
The file holds data only (inputs and expected outputs).
"""
import os
import subprocess
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def make_mlp(sizes, gen):
    layers = []
    for i in range(len(sizes) - 1):
        lin = torch.nn.Linear(sizes[i], sizes[i + 1])
        with torch.no_grad():
            bound = 1.0 / np.sqrt(sizes[i])
            lin.weight.uniform_(-bound, bound, generator=gen)
            lin.bias.uniform_(-bound, bound, generator=gen)
        layers.append(lin)
        if i < len(sizes) - 2:
            layers.append(torch.nn.ReLU())
    return torch.nn.Sequential(*layers)


def flat_params(seq):
    return np.concatenate([p.detach().numpy().reshape(-1) for p in seq.parameters()]).astype(np.float32)


def flat_grads(seq):
    return np.concatenate([p.grad.detach().numpy().reshape(-1) for p in seq.parameters()]).astype(np.float32)


def main():
    torch.set_num_threads(1)
    gen = torch.Generator().manual_seed(1234)
    D, A, H, N = 89, 90, 64, 96
    pol = make_mlp([D, H, H, A], gen)
    cri = make_mlp([D, H, H, 1], gen)
    obs = torch.randn(N, D, generator=gen) * 0.7
    out = {"D": D, "A": A, "H": H, "pol_params": flat_params(pol), "cri_params": flat_params(cri), "obs": obs.numpy()}

    # ---- GetActionProbs / GetAction
    T = 1.0
    with torch.no_grad():
        logits = pol(obs)
        probs = torch.clamp(torch.softmax(logits / T, dim=-1), min=1e-11, max=1)
        g2 = torch.Generator().manual_seed(777)
        q = torch.empty_like(probs).exponential_(1.0, generator=g2)
        g3 = torch.Generator().manual_seed(777)
        act_mn = torch.multinomial(probs, 1, True, generator=g3).flatten()
        act_q = torch.argmax(probs / q, dim=-1)
        assert torch.equal(act_mn, act_q), "multinomial != argmax(p/q) on this torch build"
        logp = torch.log(probs).gather(-1, act_mn[:, None]).flatten()
        values = cri(obs).flatten()
    out.update(logits=logits.numpy(), probs=probs.numpy(), q=q.numpy(), actions=act_mn.numpy().astype(np.int32), logp=logp.numpy(),
               values=values.numpy(), det_actions=torch.argmax(probs, dim=-1).numpy().astype(np.int32))

    # ---- PPO minibatch loss + grads (PPOLearner.cpp:139-215)
    clip, ent_coef, scale = 0.2, 0.01, 0.25
    old_logp = (logp + torch.randn(N, generator=gen) * 0.3).detach()
    adv = torch.randn(N, generator=gen)
    targets = torch.randn(N, generator=gen)
    vals = cri(obs).view(-1)
    probs_t = torch.clamp(torch.softmax(pol(obs) / T, dim=-1), min=1e-11, max=1)
    log_probs = torch.log(probs_t)
    action_log_probs = log_probs.gather(-1, act_mn[:, None]).view(-1)
    entropy = -(log_probs * probs_t).sum(dim=-1).mean()
    ratio = torch.exp(action_log_probs - old_logp)
    clipped = torch.clamp(ratio, 1 - clip, 1 + clip)
    policy_loss = -torch.min(ratio * adv, clipped * adv).mean()
    ppo_loss = (policy_loss - entropy * ent_coef) * scale
    value_loss = torch.nn.functional.mse_loss(vals, targets) * scale
    ppo_loss.backward()
    value_loss.backward()
    with torch.no_grad():
        log_ratio = action_log_probs - old_logp
        kl = ((torch.exp(log_ratio) - 1) - log_ratio).mean()
        clip_fraction = (torch.abs(ratio - 1) > clip).float().mean()
    out.update(old_logp=old_logp.numpy(), adv=adv.numpy(), targets=targets.numpy(), clip=clip, ent_coef=ent_coef, scale=scale,
               pol_grads=flat_grads(pol), cri_grads=flat_grads(cri), entropy=float(entropy), kl=float(kl), clip_fraction=float(clip_fraction),
               ratio_mean=float(ratio.mean()), value_loss=float(value_loss / scale), policy_loss=float(policy_loss))

    # ---- clip_grad_norm_(0.5) + Adam, 3 steps on the policy with fixed fake gradients
    opt = torch.optim.Adam(pol.parameters(), lr=2e-4)
    steps = []
    for s in range(3):
        for p in pol.parameters():
            p.grad = torch.randn(p.shape, generator=gen) * (0.05 * (s + 1))
        gflat = flat_grads(pol)
        torch.nn.utils.clip_grad_norm_(pol.parameters(), 0.5)
        opt.step()
        steps.append((gflat, flat_params(pol)))
    out.update(adam_lr=2e-4, adam_g0=steps[0][0], adam_p1=steps[0][1], adam_g1=steps[1][0], adam_p2=steps[1][1], adam_g2=steps[2][0], adam_p3=steps[2][1])

    # ---- GAE on a concatenated batch (formulas of Util/TorchFuncs.cpp:23-48 evaluated with torch fp32 scalars)
    B = 257
    rews = (torch.randn(B, generator=gen) * 3).numpy()
    terminal = (torch.rand(B, generator=gen) < 0.05).float().numpy()
    truncated = (torch.rand(B, generator=gen) < 0.04).float().numpy()
    truncated[-1] = 1.0
    gvals = torch.randn(B + 1, generator=gen).numpy()
    gamma, lam, ret_std, cr = 0.99, 0.95, 1.7, 10.0
    adv_o = np.zeros(B, np.float32); ret_o = np.zeros(B, np.float32)
    last_gae = torch.tensor(0.0); last_ret = torch.tensor(0.0)
    for step in range(B - 1, -1, -1):
        done = 1 - torch.tensor(terminal[step]); trunc = 1 - torch.tensor(truncated[step])
        nr = torch.clamp(torch.tensor(rews[step]) / torch.tensor(np.float32(ret_std)), -cr, cr)
        pred_ret = nr + np.float32(gamma) * torch.tensor(gvals[step + 1]) * done
        delta = pred_ret - torch.tensor(gvals[step])
        ret = torch.tensor(rews[step]) + last_ret * np.float32(gamma) * done * trunc
        ret_o[step] = ret.item(); last_ret = ret
        last_gae = delta + np.float32(gamma) * np.float32(lam) * done * trunc * last_gae
        adv_o[step] = last_gae.item()
    out.update(gae_rews=rews, gae_terminal=terminal, gae_truncated=truncated, gae_values=gvals, gae_gamma=gamma, gae_lambda=lam, gae_ret_std=ret_std,
               gae_clip=cr, gae_adv=adv_o, gae_returns=ret_o, gae_targets=(gvals[:B] + adv_o).astype(np.float32))

    # ---- libstdc++ std::shuffle with a persistent default_random_engine(123): two consecutive shuffles of iota(n)
    src = r"""
#include <algorithm>
#include <numeric>
#include <random>
#include <vector>
#include <cstdio>
int main() {
    std::default_random_engine rng(123);
    for (int rep = 0; rep < 2; rep++) {
        for (int n : {1, 2, 7, 64, 1000}) {
            std::vector<long> v(n); std::iota(v.begin(), v.end(), 0);
            std::shuffle(v.begin(), v.end(), rng);
            for (long x : v) printf("%ld ", x);
            printf("\n");
        }
    }
}
"""
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "s.cpp"), "w").write(src)
        subprocess.check_call(["g++", "-O1", "-o", os.path.join(td, "s"), os.path.join(td, "s.cpp")])
        lines = subprocess.check_output([os.path.join(td, "s")]).decode().strip().split("\n")
    for i, ln in enumerate(lines):
        out[f"shuffle_{i}"] = np.array([int(x) for x in ln.split()], np.int64)

    np.savez_compressed(os.path.join(HERE, "learner_golden.npz"), **out)
    print("wrote learner_golden.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
