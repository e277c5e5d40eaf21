// Indices into PassLaunch::params for the crt-royale kernels: per-launch constants that the
// GLSL vertex shaders derive from uniforms, computed on the host by royale_setup.cpp.
#pragma once

enum {  // pass 0
  RP0_INTERLACED = 0,
};
enum {  // pass 1
  RP1_Y_STEP = 0, RP1_UV_STEP_Y, RP1_PH, RP1_TSY,
};
enum {  // blur9 (passes 3, 4)
  RPB_W12 = 0, RPB_W34, RPB_K12, RPB_K34, RPB_SUM_INV, RPB_DX, RPB_DY,
};
enum { RP5_MAG_Y = 0 };
enum { RP6_MAG_X = 0, RP6_SRC_DX, RP6_TILE_SIZE_UV_X };
enum {  // pass 7
  RP7_TPS_X = 0, RP7_TPS_Y, RP7_START_X, RP7_START_Y, RP7_UVS_X, RP7_UVS_Y, RP7_SCAN_TW, RP7_SCAN_TH, RP7_SCAN_TIX, RP7_SCAN_TIY,
};
enum { RP8_CENTER_WEIGHT = 0, RP8_MASK_AMPLIFY };
enum {  // blur17 (passes 9, 10)
  RPG_W12 = 0, RPG_W34, RPG_W56, RPG_W78, RPG_K12, RPG_K34, RPG_K56, RPG_K78, RPG_SUM_INV, RPG_DXY, RPG_MASK_AMPLIFY,
};
enum {  // pass 11: params[0..43] are the shader's 44 #pragma parameters; derived values follow
  RP11_ASPECT_X = 44, RP11_ASPECT_Y = 45,
  // general form (tex2Daa / curved geometry, pass_royale_last_general.hip): the vertex stage's outputs by varying slot
  // (4 * VARn + component, gen/royale_last_fs.inc royale_last_fs_inputs); slots 0 and 1 (tex_uv) are planes instead
  RP11_VARYING0 = 48,
};
constexpr int kLastVaryings = 26;
// geometry-aa-last-pass.glsl FS 5480: geom_mode_runtime > 0.5 or geom_overscan != 1 select the tex2Daa / ray-cast form
inline bool lastIsGeneral(const float* P) { return P[30] > 0.5f || P[37] != 1.0f || P[38] != 1.0f; }

// PassLaunch::flags
enum { RC_FLAG_UNDEF_VARYING_ZERO = 1 };
