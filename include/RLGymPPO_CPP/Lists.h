// The list vocabulary of the learner layer.  The reference declares FList / FList2 / IList a second time here
// (PUB/Lists.h:4-31); this build has one definition, in the sim layer's Framework.h, and re-exports it so that both
// `RLGSC::FList` and `RLGPC::FList` name the same std::vector<float>.
#pragma once
#include <RLGymSim_CPP/Framework.h>

namespace RLGPC {
using RLGSC::FList;    // std::vector<float>
using RLGSC::FList2;   // std::vector<FList>
using RLGSC::IList;    // std::vector<int>
}
