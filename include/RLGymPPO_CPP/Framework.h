#pragma once
#include "Lists.h"
#include <chrono>
#include <iomanip>
#include <iostream>
#define RG_LOG(s) do { std::cout << s << std::endl; } while (0)
