// Exact linear -> sRGB8 conversion of the GL's sRGB render targets (see srgb_encode.cpp).
#pragma once
#include <cstdint>
#include <vector>

#include "kernels/rc_device.h"

namespace rc {
// the byte an sRGB8 render target stores for x (host evaluation of the full algorithm)
uint8_t srgb8Encode(float x);
// per-run table the kernels use (rcd::kSrgbRuns entries: byte at the run's start << 16 | offset of the
// crossing inside the run, 8192 = none; bit 14: the next run starts with the byte this one ends with); false if
// the measured structure does not hold
bool buildSrgbRunTable(std::vector<uint32_t>* table);
uint8_t srgb8EncodeByRunTable(float x, const uint32_t* table);
// second form (rcd::kSrgb2Runs entries, (byte << 13) + (8192 - crossing), bit 30 = the first form's bit 14): covers the
// linear segment too, so the device encode is clamp, one LDS read, one add, one bit-field extract
bool buildSrgbRunTable2(std::vector<uint32_t>* table);
uint8_t srgb8EncodeByRunTable2(float x, const uint32_t* table);
// both tables (first form, then second form at + rcd::kSrgbRuns) in device memory of `device` (current device must be `device`); created on first use
const uint32_t* deviceSrgbRunTable(int device);
}  // namespace rc
