// Stand-in for <tiny-cuda-nn/gpu_matrix.h> (test infrastructure): included by the reference's envmap.cuh, nothing of it is used on the path.
#pragma once
#include <tiny-cuda-nn/common.h>
