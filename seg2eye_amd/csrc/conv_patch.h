// Internal interface of the patch-resident stride-1 convolution (3x3 and 4x4; conv_patch.hip).
#pragma once
#include "common.h"

struct s2e_patch_plan {
    int tw, th;       // rectangle of output pixels per tile: width <= 64, height; tw * th <= 256
    int splits;       // channel-chunk splits (> 1: fp32 partial slabs + conv_finish_kernel)
    int s2d;          // 0; 1 / 2: a 4x4 stride-2 pad-2 layer (forward / data gradient) as a 2x2 conv over the space-to-depth view
    int bn;           // 0 = by Cout (128 above 64 channels); 64: split layers that fill the chip better with narrower tiles and fewer splits
};
// 0 = the generic implicit-GEMM kernel runs this shape; 1 = this kernel does, with *plan (may be NULL) filled in.
int s2e_conv_patch_plan(int dtype, const s2e_conv_desc* d, s2e_patch_plan* plan);
size_t s2e_conv_patch_workspace_bytes(int dtype, const s2e_conv_desc* d);
int s2e_conv_patch_launch(int dtype, const s2e_patch_plan* plan, const void* x, const void* w, const float* bias, const void* res,
                          const void* aux, void* y, const s2e_conv_desc* d, int kpad, float* partial, hipStream_t st);

// the rectangle of output pixels (any width <= 64, as many rows as fit 256 pixels and the 400-pixel patch) that fills best;
// returns the fraction of the 256 accumulator rows that land on image pixels
double s2e_patch_rectangle(const s2e_conv_desc* d, int ks, int* tw_out, int* th_out);

// Duo form (conv_duo.hip): two independent 4-wave workgroups per CU, half a tile out of phase; 256 pixels x 128 channels per
// workgroup, 32-channel K-steps, wave-private epilogue; bf16 3x3.
int s2e_conv_duo_plan(int dtype, const s2e_conv_desc* d, s2e_patch_plan* plan);
int s2e_conv_duo_launch(const s2e_patch_plan* plan, const void* x, const void* w, const float* bias, const void* res,
                        const void* aux, void* y, const s2e_conv_desc* d, int kpad, float* stats_part, const int* rect_list,
                        const int* rect_count, hipStream_t st);
int s2e_conv_duo_stats_slots(const s2e_conv_desc* d, const s2e_patch_plan* plan);
int s2e_spade_conv_modulate_duo(int dtype, const void* actv, const void* w_packed, const float* bias, const void* x,
                                const float* stats, const float* style, int style_ld, void* out, void* gamma_out,
                                int N, int H, int W, int C, int nh, int lrelu, int flags, int tw, int th,
                                const int* rect_list, const int* rect_count, hipStream_t st);
