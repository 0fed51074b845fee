// Internal interface of the degenerate-channel conv kernels (conv_small.hip).
#pragma once
#include "common.h"

enum { SMALL_NONE = 0, SMALL_FWD_COUT1 = 1, SMALL_FWD_CIN1 = 2, SMALL_DGRAD_COUT1 = 3, SMALL_WGRAD_COUT1 = 4, SMALL_WGRAD_CIN1 = 5,
       SMALL_FWD_C8S2 = 6, SMALL_DGRAD_C8S2 = 7 };       // the PatchGAN's 8-channel 4x4 stride-2 first layer (conv_c8.hip)

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

// conv_c8.hip
bool s2e_c8s2_fwd_ok(int dtype, const s2e_conv_desc* d);
int s2e_c8s2_fwd_launch(const SmallConvParams& p, hipStream_t st);
bool s2e_c8s2_dgrad_ok(int dtype, const s2e_conv_desc* d);
int s2e_c8s2_dgrad_launch(const SmallConvParams& p, hipStream_t st);
bool s2e_c8s2_wgrad_ok(int dtype, const s2e_conv_desc* d);
// dW / dbias of jobs[idx[0 .. n)] are ADDED to (chunks of 4 jobs a launch; partial tiles in the workspace + a reduction launch, or -- without
// s2e_c8s2_wgrad_workspace_bytes(n) of workspace -- fp32 atomics)
size_t s2e_c8s2_wgrad_workspace_bytes(int n_jobs);
int s2e_c8s2_wgrad_launch(const s2e_wgrad_multi_job* jobs, const int* idx, int n, void* workspace, size_t workspace_bytes, hipStream_t st);
