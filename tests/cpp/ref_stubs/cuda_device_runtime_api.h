// TEST-ONLY stand-in (see cuda_runtime_api.h in this directory)
#pragma once
#include "cuda_runtime_api.h"
