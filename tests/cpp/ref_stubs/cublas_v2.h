// TEST-ONLY stand-in: base/cuda_config.h includes <cublas_v2.h> and uses nothing from it
#pragma once
