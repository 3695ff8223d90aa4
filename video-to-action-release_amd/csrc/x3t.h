// Internal (not part of the C ABI): launcher of conv_frames_x3 (csrc/igemm_x3t.hip), called from conv_dma_launch (csrc/igemm_h.hip).
#pragma once
#include "common.h"
int conv_frames_x3_eligible(int B, int F, int HW, int C, int Cout, int rows_per_batch, bool has_rowvec);
int conv_frames_x3_launch(const float* x, const float* w_packed, const float* bias, const float* rowvec, const float* residual, float* y,
                          const void* zeros, int B, int F, int HW, int C, int Cout, int rows_per_batch, float* stats, hipStream_t stream);
int conv_patch_x3_eligible(int N, int H, int W, int C, int Cout, int ncu);
int conv_patch_x3_launch(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, const void* zeros, int N,
                         int H, int W, int C, int Cout, int ups, hipStream_t stream);
extern "C" int v2a_conv2d_x3p_eligible(int N, int H, int W, int C, int Cout);
int conv_patch_x3_ups4_eligible(int N, int H, int W, int C, int Cout, int ncu);
// conv_maps_x3 (csrc/igemm_x3m.hip): 3 x 3 / stride 1 / pad 1 over N square maps of S = 32 / 16 / 8 / 4, split over 32-channel chunks
int conv_maps_x3_eligible(int N, int S, int C, int Cout);
int conv_maps_x3_split(int M, int Cout, int C);
int conv_maps_x3_launch(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, float* partial,
                        const void* zeros, int N, int S, int C, int Cout, int splitk, hipStream_t stream);
