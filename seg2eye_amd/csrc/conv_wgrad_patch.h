// Internal interface of the patch-resident 3x3 stride-1 weight-gradient kernel (conv_wgrad_patch.hip).
#pragma once
#include "common.h"

// 0 = the generic kernel (conv_wgrad.hip) runs this shape; otherwise the pixel-slab width (16, 32 or 64)
int s2e_wgrad_patch_plan(int dtype, const s2e_conv_desc* d);
size_t s2e_wgrad_patch_workspace_bytes(int slab_w, const s2e_conv_desc* d);
int s2e_wgrad_patch_launch(int slab_w, const void* x, const void* gy, float* dw, float* dbias, const s2e_conv_desc* d,
                           void* workspace, size_t workspace_bytes, const int* rect_list, const int* rect_count, hipStream_t st);

// Cin = 8 (label-map convs): 0 = generic kernel; otherwise the slab width.  Needs its workspace (no atomics fallback).
int s2e_wgrad_c8_plan(int dtype, const s2e_conv_desc* d);
size_t s2e_wgrad_c8_workspace_bytes(int slab_w, const s2e_conv_desc* d);
int s2e_wgrad_c8_launch(int slab_w, const void* x, const void* gy, float* dw, float* dbias, const s2e_conv_desc* d,
                        void* workspace, hipStream_t st);
