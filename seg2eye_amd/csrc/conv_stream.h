// Internal interface of the persistent stream-K implicit-GEMM convolution (conv_stream.hip).
#pragma once
#include "common.h"

// 0 = conv_igemm.hip runs this shape; 1 = this kernel does (bf16, vector channel counts, no fused input activation,
// Cout > 32: every generic forward / data-gradient launch of the Seg2Eye step).
int s2e_conv_stream_plan(int dtype, const s2e_conv_desc* d);
size_t s2e_conv_stream_workspace_bytes(int dtype, const s2e_conv_desc* d);
int s2e_conv_stream_launch(const void* x, const void* w, const float* bias, const void* res, const void* aux, void* y,
                           const s2e_conv_desc* d, int kpad, void* workspace, size_t workspace_bytes, hipStream_t st);
