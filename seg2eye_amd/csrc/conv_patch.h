// Internal interface of the patch-resident stride-1 convolution (3x3 and 4x4; conv_patch.hip).
#pragma once
#include "common.h"

struct s2e_patch_plan {
    int tw, th;       // rectangle of output pixels per tile: width <= 64, height; tw * th <= 256
    int splits;       // channel-chunk splits (> 1: fp32 partial slabs + conv_finish_kernel)
};
// 0 = the generic implicit-GEMM kernel runs this shape; 1 = this kernel does, with *plan (may be NULL) filled in.
int s2e_conv_patch_plan(int dtype, const s2e_conv_desc* d, s2e_patch_plan* plan);
size_t s2e_conv_patch_workspace_bytes(int dtype, const s2e_conv_desc* d);
int s2e_conv_patch_launch(int dtype, const s2e_patch_plan* plan, const void* x, const void* w, const float* bias, const void* res,
                          const void* aux, void* y, const s2e_conv_desc* d, int kpad, float* partial, hipStream_t st);
