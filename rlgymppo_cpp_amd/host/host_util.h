// Internals shared by the host layer's translation units (Learner.hip, Gym.hip, SkillTracker.hip): not part of the public headers.
#pragma once
#include <hip/hip_runtime.h>
#include <RLGymSim_CPP/Gym.h>

#define HOST_HIP(call)                                                                                   \
    do {                                                                                                 \
        hipError_t _e = (call);                                                                          \
        if (_e != hipSuccess) RG_ERR_CLOSE(#call << " failed: " << hipGetErrorString(_e));               \
    } while (0)

namespace RLGSC {
// the arena mesh RocketSim::Init pointed at ("<folder>/soccar/*.cmf"), or the procedural soccar mesh when there is none
void LoadArenaMesh(rlgpu_env* env, bool quiet);
// a host Arena facade with the cars of one env in the device's slot order (blue, orange, blue, ...; blue only without opponents)
Arena* MakeScratchArena(int teamSize, bool spawnOpponents);
}
