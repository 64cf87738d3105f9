#pragma once
#include "../../include/mmx_hip.h"
typedef MmxGemmParams GemmParams;
