// RLBotClient.h (TOP/RLBotClient.h:1-52) -- the reference's example program includes this header next to its sources.  The RLBot
// socket client it declares lives in an un-vendored third-party library (TOP/RLBotCPP), which is outside this build; what is kept is
// the parameter block and the inference object a bot is built around (RLGPC::InferUnit), so programs written against the header
// compile, and RLBotClient::Run says what is missing instead of failing to link.
#pragma once
#include <RLGymSim_CPP/Utils/OBSBuilders/OBSBuilder.h>
#include <RLGymSim_CPP/Utils/ActionParsers/ActionParser.h>
#include <RLGymPPO_CPP/Util/InferUnit.h>

struct RLBotParams {
    int port = 0;                                  // the port of rlbot/port.cfg
    RLGSC::OBSBuilder* obsBuilder = NULL;
    RLGSC::ActionParser* actionParser = NULL;
    std::filesystem::path policyPath;              // a trained PPO_POLICY.lt
    int obsSize = 0;
    std::vector<int> policyLayerSizes = {};
    int tickSkip = 8;
};

namespace RLBotClient {
inline void Run(const RLBotParams&) { RG_ERR_CLOSE("RLBotClient::Run(): the RLBot socket client (RLBotCPP) is not part of this build; drive RLGPC::InferUnit from your own bot loop"); }
}
