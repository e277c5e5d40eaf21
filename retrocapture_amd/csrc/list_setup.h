// Per-launch host setup of the instruction-list passes (kernels/pass_lists.hip): uniform blocks by name, image-adjustment's vertex stage.
#pragma once
#include "kernel_registry.h"

namespace rc {
void setupTvoutTweaks(const PassGeometry& g, rcd::PassLaunch& L);
void setupImageAdjustment(const PassGeometry& g, rcd::PassLaunch& L);
void setupJinc2Sharper(const PassGeometry& g, rcd::PassLaunch& L);
void setupCrtLottes(const PassGeometry& g, rcd::PassLaunch& L);
void setupFakeLottes(const PassGeometry& g, rcd::PassLaunch& L);
void setupSideBySide(const PassGeometry& g, rcd::PassLaunch& L);
void setupSameboyLcd(const PassGeometry& g, rcd::PassLaunch& L);
void setupCrtConsumer(const PassGeometry& g, rcd::PassLaunch& L);
void setupReverseAa(const PassGeometry& g, rcd::PassLaunch& L);
void setupAdvancedAa(const PassGeometry& g, rcd::PassLaunch& L);
}  // namespace rc
