// oracle/ref_learner_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into, imported by or executed from the product).
// A C-ABI door into the REAL reference learner code, compiled from where it lies under /root/reference against the libtorch that
// ships inside the torch wheel (CPU):  RLGPC::TorchFuncs::ComputeGAE (PRIV/Util/TorchFuncs.cpp:5-52), RLGPC::DiscretePolicy
// (PRIV/PPO/DiscretePolicy.cpp: GetActionProbs :44-49, GetAction :51-62, GetBackpropData :64-75) and RLGPC::ValueEstimator
// (PRIV/PPO/ValueEstimator.cpp), RLGPC::ExperienceBuffer (PRIV/PPO/ExperienceBuffer.cpp).  Those translation units build unedited; PPOLearner.cpp (loss + optimizer loop) needs the
// reference's MSVC-only gradscaler.hpp and is NOT buildable here, so the PPO loss / gradients / Adam stay pinned by the
// torch-Python goldens (tests/golden/make_learner_golden.py).  What this driver adds is only argument marshalling.
#include <private/RLGymPPO_CPP/Util/TorchFuncs.h>
#include <private/RLGymPPO_CPP/PPO/DiscretePolicy.h>
#include <private/RLGymPPO_CPP/PPO/ValueEstimator.h>
#include <private/RLGymPPO_CPP/PPO/ExperienceBuffer.h>
#include <public/RLGymPPO_CPP/Util/WelfordRunningStat.h>
#include <torch/torch.h>
#include <cstring>

using namespace RLGPC;

namespace {
// flat fp32 parameters in state-dict order (0.weight, 0.bias, 2.weight, ...) into a reference network
void load_params(torch::nn::Sequential& seq, const float* flat) {
    torch::NoGradGuard ng;
    size_t off = 0;
    for (auto& p : seq->parameters()) {
        const int64_t n = p.numel();
        p.copy_(torch::from_blob((void*)(flat + off), {n}, torch::kFloat32).view(p.sizes()));
        off += (size_t)n;
    }
}
IList hidden_list(const int* hidden, int n) { return IList(hidden, hidden + n); }
}  // namespace

extern "C" {

// values: B + 1 entries.  Outputs: B entries each.
void refl_gae(const float* rews, const float* dones, const float* truncs, const float* values, int B, float gamma, float lambda, float ret_std,
              float clip_range, float* adv_out, float* targets_out, float* returns_out) {
    FList r(rews, rews + B), d(dones, dones + B), t(truncs, truncs + B), v(values, values + B + 1);
    torch::Tensor adv, tgt; FList ret;
    TorchFuncs::ComputeGAE(r, d, t, v, adv, tgt, ret, gamma, lambda, ret_std, clip_range);
    adv = adv.contiguous().to(torch::kFloat32).cpu(); tgt = tgt.contiguous().to(torch::kFloat32).cpu();
    std::memcpy(adv_out, adv.data_ptr<float>(), (size_t)B * 4);
    std::memcpy(targets_out, tgt.data_ptr<float>(), (size_t)B * 4);
    std::memcpy(returns_out, ret.data(), (size_t)B * 4);
}

// DiscretePolicy on `rows` observations: clamped probabilities, the deterministic action, log-prob of the given actions, mean entropy,
// and -- with a non-zero seed -- GetAction's sampled actions together with the Exp(1) tape torch::multinomial consumed for them
// (multinomial(p, 1, replacement) == argmax(p / q) with q ~ Exp(1) drawn from the same generator state).
void refl_policy(const int* hidden, int n_hidden, int D, int A, const float* params, float temperature, const float* obs, int rows, const int* acts,
                 float* probs_out, int* argmax_out, float* logp_out, float* entropy_mean_out, uint64_t seed, int* sampled_out, float* sampled_logp_out,
                 float* exp_tape_out) {
    DiscretePolicy policy(D, A, hidden_list(hidden, n_hidden), torch::kCPU, temperature);
    load_params(policy.seq, params);
    torch::NoGradGuard ng;
    torch::Tensor x = torch::from_blob((void*)obs, {rows, D}, torch::kFloat32).clone();
    torch::Tensor probs = policy.GetActionProbs(x).contiguous();
    std::memcpy(probs_out, probs.data_ptr<float>(), (size_t)rows * A * 4);
    auto det = policy.GetAction(x, true);
    torch::Tensor da = det.action.to(torch::kInt32).contiguous();
    std::memcpy(argmax_out, da.data_ptr<int32_t>(), (size_t)rows * 4);
    torch::Tensor a = torch::from_blob((void*)acts, {rows, 1}, torch::kInt32).clone();
    auto bp = policy.GetBackpropData(x, a);
    torch::Tensor lp = bp.actionLogProbs.contiguous().view({rows});
    std::memcpy(logp_out, lp.data_ptr<float>(), (size_t)rows * 4);
    *entropy_mean_out = bp.entropy.item<float>();
    if (seed != 0) {
        torch::manual_seed(seed);
        auto s = policy.GetAction(x, false);
        torch::Tensor sa = s.action.to(torch::kInt32).contiguous(), sl = s.logProb.contiguous();
        std::memcpy(sampled_out, sa.data_ptr<int32_t>(), (size_t)rows * 4);
        std::memcpy(sampled_logp_out, sl.data_ptr<float>(), (size_t)rows * 4);
        torch::manual_seed(seed);
        torch::Tensor q = torch::empty_like(probs).exponential_(1);
        std::memcpy(exp_tape_out, q.contiguous().data_ptr<float>(), (size_t)rows * A * 4);
    }
}

// The real ExperienceBuffer: `n_submits` submissions of `rows_per_submit` rows whose `actions` column carries the row's identity
// (submit * rows_per_submit + i); after every submission GetAllBatchesShuffled(batch_size) is called once and the identities of all
// batches are appended to ids_out (rows of batch 0, batch 1, ...).  counts_out[s] = number of batches after submission s.
// Returns the number of ids written.
int64_t refl_expbuf_run(int64_t max_size, int seed, int n_submits, int rows_per_submit, int64_t batch_size, int64_t* ids_out, int64_t ids_cap, int32_t* counts_out) {
    ExperienceBuffer buf(max_size, seed, torch::kCPU);
    int64_t n = 0;
    for (int s = 0; s < n_submits; s++) {
        ExperienceTensors t;
        torch::Tensor ids = torch::arange((int64_t)s * rows_per_submit, (int64_t)(s + 1) * rows_per_submit, torch::kFloat64);
        for (torch::Tensor& x : t) x = torch::zeros({rows_per_submit, 1}, torch::kFloat64);
        t.actions = ids.view({rows_per_submit, 1}).clone();
        buf.SubmitExperience(t);
        auto batches = buf.GetAllBatchesShuffled(batch_size);
        counts_out[s] = (int32_t)batches.size();
        for (auto& b : batches) {
            torch::Tensor a = b.actions.contiguous().view({-1});
            for (int64_t i = 0; i < a.numel(); i++) { if (n < ids_cap) ids_out[n] = (int64_t)a[i].item<double>(); n++; }
        }
    }
    return n;
}

// The same with submissions of different sizes (what a free-running collection hands over: ThreadAgentManager.cpp:47-60 concatenates whatever each
// trajectory holds): submission s has sizes[s] rows with identities first_id[s] + i.
int64_t refl_expbuf_run_sizes(int64_t max_size, int seed, int n_submits, const int32_t* sizes, int64_t batch_size, int64_t* ids_out, int64_t ids_cap, int32_t* counts_out) {
    ExperienceBuffer buf(max_size, seed, torch::kCPU);
    int64_t n = 0, next_id = 0;
    for (int s = 0; s < n_submits; s++) {
        ExperienceTensors t;
        const int64_t rows = sizes[s];
        torch::Tensor ids = torch::arange(next_id, next_id + rows, torch::kFloat64);
        next_id += rows;
        for (torch::Tensor& x : t) x = torch::zeros({rows, 1}, torch::kFloat64);
        t.actions = ids.view({rows, 1}).clone();
        buf.SubmitExperience(t);
        auto batches = buf.GetAllBatchesShuffled(batch_size);
        counts_out[s] = (int32_t)batches.size();
        for (auto& b : batches) {
            torch::Tensor a = b.actions.contiguous().view({-1});
            for (int64_t i = 0; i < a.numel(); i++) { if (n < ids_cap) ids_out[n] = (int64_t)a[i].item<double>(); n++; }
        }
    }
    return n;
}

// WelfordRunningStat (PUB/Util/WelfordRunningStat.h:5-84), shape 1, fed in chunks of `chunk` samples like Learner.cpp:679-682 does
void refl_welford(const float* samples, int n, int chunk, double* mean_out, double* m2_out, int64_t* count_out, float* std_out) {
    WelfordRunningStat w(1);
    for (int i = 0; i < n; i += chunk) {
        const int k = std::min(chunk, n - i);
        FList part(samples + i, samples + i + k);
        w.Increment(part, k);
    }
    *mean_out = w.runningMean[0]; *m2_out = w.runningVariance[0]; *count_out = w.count; *std_out = w.GetSTD()[0];
}

void refl_value(const int* hidden, int n_hidden, int D, const float* params, const float* obs, int rows, float* values_out) {
    ValueEstimator critic(D, hidden_list(hidden, n_hidden), torch::kCPU);
    load_params(critic.seq, params);
    torch::NoGradGuard ng;
    torch::Tensor x = torch::from_blob((void*)obs, {rows, D}, torch::kFloat32).clone();
    torch::Tensor v = critic.Forward(x).contiguous().view({rows});
    std::memcpy(values_out, v.data_ptr<float>(), (size_t)rows * 4);
}

}
