// Internal interface of the patch-resident 3x3 stride-1 convolution (conv_patch.hip).
#pragma once
#include "common.h"

// 0 = the generic implicit-GEMM kernel runs this shape; otherwise the pixel-tile width (16, 32 or 64) to launch with.
// *splits (may be NULL) = number of channel-chunk splits (> 1: fp32 partial slabs + conv_finish_kernel).
int s2e_conv_patch_plan(int dtype, const s2e_conv_desc* d, int* splits);
size_t s2e_conv_patch_workspace_bytes(int dtype, const s2e_conv_desc* d);
int s2e_conv_patch_launch(int dtype, int tile_w, int splits, const void* x, const void* w, const float* bias, const void* res,
                          const void* aux, void* y, const s2e_conv_desc* d, int kpad, float* partial, hipStream_t st);
