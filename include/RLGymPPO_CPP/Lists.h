#pragma once
#include <RLGymSim_CPP/Framework.h>
namespace RLGPC { using RLGSC::FList; using RLGSC::FList2; using RLGSC::IList; }
