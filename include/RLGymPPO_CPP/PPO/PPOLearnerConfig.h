// PPOLearnerConfig: every field name and default of PUB/PPO/PPOLearnerConfig.h:6-32
#pragma once
#include "../Lists.h"
namespace RLGPC {
struct PPOLearnerConfig {
    IList policyLayerSizes = {256, 256, 256};
    IList criticLayerSizes = {256, 256, 256};
    int64_t batchSize = 50 * 1000;
    int epochs = 10;
    float policyLR = 3e-4f;
    float criticLR = 3e-4f;
    float entCoef = 0.005f;
    float clipRange = 0.2f;
    int64_t miniBatchSize = 0;          // 0 = batchSize
    bool autocastLearn = false;         // bf16 MFMA operands / bf16 activations, fp32 accumulate and master weights (rlgpu use_bf16)
    bool halfPrecModels = false;        // accepted, unused (the bf16 path already keeps bf16 weight shadows)
    float policyTemperature = 1;
    bool measureGradientNoise = false;  // not built
    int gradientNoiseUpdateInterval = 10;
    float gradientNoiseAvgDecay = 0.9925f;
};
}
