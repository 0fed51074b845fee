// Internal interface of the degenerate-channel conv kernels (conv_small.hip).
#pragma once
#include "common.h"

enum { SMALL_NONE = 0, SMALL_FWD_COUT1 = 1, SMALL_FWD_CIN1 = 2, SMALL_DGRAD_COUT1 = 3, SMALL_WGRAD_COUT1 = 4, SMALL_WGRAD_CIN1 = 5 };

struct SmallConvParams {
    const void* x; const void* w; const float* bias; const void* res; const void* aux; void* y;
    const void* gy; float* dw;
    int N, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad, in_act, out_act, aux_mode, Kpad;
};

int s2e_small_conv_kind(int dtype, const s2e_conv_desc* d);
int s2e_small_conv_launch(int dtype, int kind, const SmallConvParams& p, hipStream_t st);
int s2e_small_wgrad_kind(int dtype, const s2e_conv_desc* d);
size_t s2e_small_wgrad_workspace_bytes(int dtype, int kind, const s2e_conv_desc* d);
int s2e_small_wgrad_launch(int dtype, int kind, const s2e_conv_desc* d, const SmallConvParams& p, void* workspace,
                           size_t workspace_bytes, hipStream_t st);
