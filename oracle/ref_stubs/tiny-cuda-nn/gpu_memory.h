// Stand-in for <tiny-cuda-nn/gpu_memory.h> (test infrastructure): the render-path sources only DECLARE functions returning it.
#pragma once
#include <tiny-cuda-nn/common.h>
namespace tcnn {
template <typename T> class GPUMemory {
public:
	T* data() const { return nullptr; }
};
} // namespace tcnn
