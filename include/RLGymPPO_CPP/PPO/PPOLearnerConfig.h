// PPOLearnerConfig -- the knobs of the PPO update.  Field names and defaults are the reference's (PUB/PPO/PPOLearnerConfig.h:6-32),
// because user code assigns them by name (cfg.ppo.batchSize = ...); what each one means in this build:
//
//   policyLayerSizes / criticLayerSizes   hidden widths of the two MLPs (Linear + ReLU per entry, then the output layer)
//   batchSize                             rows per optimizer step; the experience buffer is cut into floor(size / batchSize) batches
//   miniBatchSize                         rows per forward/backward launch sequence (gradients accumulate over a batch); 0 = batchSize
//   epochs                                passes over the whole experience buffer per iteration, a fresh shuffle each
//   policyLR / criticLR                   Adam step sizes (beta 0.9 / 0.999, eps 1e-8, gradient norm clipped to 0.5 per network)
//   entCoef, clipRange                    entropy bonus and PPO ratio clip
//   policyTemperature                     softmax temperature of the policy head
//   autocastLearn                         the reference's fp16 autocast switch; here: bf16 MFMA operands and activations with fp32
//                                         accumulation and fp32 master weights (RlgpuLearnerConfig::use_bf16)
//   halfPrecModels                        accepted for source compatibility; the bf16 path already keeps bf16 weight copies
//   measureGradientNoise (+ interval, decay)   accepted, not built
#pragma once
#include "../Lists.h"

namespace RLGPC {

struct PPOLearnerConfig {
    // networks
    IList policyLayerSizes = {256, 256, 256};
    IList criticLayerSizes = {256, 256, 256};

    // batching
    int64_t batchSize = 50 * 1000;
    int epochs = 10;

    // optimisation
    float policyLR = 3e-4f;
    float criticLR = 3e-4f;
    float entCoef = 0.005f;
    float clipRange = 0.2f;

    int64_t miniBatchSize = 0;

    // precision
    bool autocastLearn = false;
    bool halfPrecModels = false;

    float policyTemperature = 1;

    // gradient-noise diagnostics of the reference (fields only)
    bool measureGradientNoise = false;
    int gradientNoiseUpdateInterval = 10;
    float gradientNoiseAvgDecay = 0.9925f;
};

}  // namespace RLGPC
