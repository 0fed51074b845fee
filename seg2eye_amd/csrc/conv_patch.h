// Internal interface of the patch-resident 3x3 stride-1 convolution (conv_patch.hip).
#pragma once
#include "common.h"

// 0 = the generic implicit-GEMM kernel runs this shape; otherwise the pixel-tile width (32 or 64) to launch with
int s2e_conv_patch_plan(int dtype, const s2e_conv_desc* d);
int s2e_conv_patch_launch(int dtype, int tile_w, const void* x, const void* w, const float* bias, const void* res,
                          const void* aux, void* y, const s2e_conv_desc* d, int kpad, hipStream_t st);
