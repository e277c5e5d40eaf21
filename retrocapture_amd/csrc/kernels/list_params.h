// PassLaunch::params layout of the instruction-list passes of pass_lists.hip: the shader's #pragma parameters first (as for every
// pass), the list's uniform block - in the dword layout gen/<name>_fs.inc addresses - from kListU0 on (filled by list_setup.cpp).
#pragma once
constexpr int kListU0 = 34;   // past the longest parameter list (crt-consumer: 33)
constexpr int kTvoutU = 10;             // tvout_tweaks_fs_uniforms: 6 parameters, TextureSize, InputSize
constexpr int kImageAdjU = 21;          // image_adjustment_fs_uniforms: 16 parameters, FrameCount, TextureSize, InputSize
constexpr int kImageAdjFrameCount = 16;
constexpr int kJinc2U = 2;              // jinc2_sharper_fs_uniforms: TextureSize
constexpr int kLottesU = 24;            // crt_lottes_fs_uniforms: sizes, 13 parameters, gl_FbWposYTransform at 20
constexpr int kFakeLottesU = 20;        // fakelottes_fs_uniforms: sizes, 10 parameters, gl_FbWposYTransform at 16
constexpr int kSbsU = 9;                // side_by_side_fs_uniforms: TextureSize, InputSize, five parameters
constexpr int kSameboyLcdU = 5;         // sameboy_lcd_fs_uniforms: TextureSize, three parameters
constexpr int kConsumerU = 44;          // crt_consumer_fs_uniforms: FrameCount at 0, sizes, 33 parameters, gl_FbWposYTransform at 40
constexpr int kReverseAaU = 3;          // reverse_aa_fs_uniforms: TextureSize, REVERSEAA_SHARPNESS
