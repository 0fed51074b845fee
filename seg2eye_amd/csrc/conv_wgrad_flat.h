// Internal interface of the flat-slab patch-resident weight-gradient kernel (conv_wgrad_flat.hip): the stride-2, 4x4 and 1x1 layers.
#pragma once
#include "common.h"

enum { WF_NONE = 0, WF_K1 = 1, WF_K3S1 = 2, WF_K3S2 = 3, WF_K4S1 = 4, WF_K4S2 = 5 };

// WF_NONE = not a shape of this kernel (or switched off: S2E_WGRAD_FLAT, a bit mask over the kinds, bit k = kind k); otherwise the kind
int s2e_wgrad_flat_kind(int dtype, const s2e_conv_desc* d);
// dW / dbias of jobs[idx[0 .. n)] are ADDED to with fp32 atomics, all in one launch (chunks of 24 jobs)
int s2e_wgrad_flat_launch(const s2e_wgrad_multi_job* jobs, const int* idx, int n, hipStream_t st);
