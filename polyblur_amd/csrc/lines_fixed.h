// Fixed-plan line transforms of the estimation (lines_fixed.hip): the line lengths whose plan is known at compile time.
// PB_ERR_UNSUPPORTED = not one of them: the caller runs the run-time-plan kernel of estimate.hip.
#pragma once
#include "common.h"

// column transform: the directional maxima (gy_out == nullptr: grad_cols_kernel's MODE 1), or the y derivative itself as
// float / __half planes (gy_out, gy_dtype = PB_F32 | PB_F16: MODE 0; gx, mags, n_angles, discard_sat unused)
int pb_launch_cols_fixed(pb_ctx *ctx, const float *gray, const float *gx, int P, int H, int W, int lognb, unsigned *mags,
                         int n_angles, int discard_sat, const FftPlan *pl, void *gy_out = nullptr, int gy_dtype = PB_F32);
// row transform; C == 0: float planes as they are (gx_dtype = PB_F32 | PB_F16 planes out), else gray + range + transform
int pb_launch_rows_fixed(pb_ctx *ctx, const float *in, int C, float *gray, void *gx, float2 *part, long images, int H, int W, int nth,
                         const FftPlan *pl, int gx_dtype = PB_F32);
// whether both transforms of an H x W plane are compiled in (the typed outputs exist in these kernels only)
bool pb_lines_fixed_shape(int H, int W);
