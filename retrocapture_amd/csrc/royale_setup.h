// Host half of the crt-royale kernels: what the GLSL vertex shaders compute from uniforms
// (tile sizes, sigma, blur weights, texture-coordinate scales), evaluated once per launch in
// float with the vertex shaders' operation order, and the plane equations of their varyings.
#pragma once
#include "kernel_registry.h"

namespace rc {
void registerRoyaleKernels(std::vector<KernelEntry>& r);
}
