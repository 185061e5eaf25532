// Fixed-plan line transforms of the estimation (lines_fixed.hip): the column transform + directional maxima for the line
// lengths whose plan is known at compile time.  PB_ERR_UNSUPPORTED = not one of them: the caller runs the run-time-plan kernel.
#pragma once
#include "common.h"

int pb_launch_cols_fixed(pb_ctx *ctx, const float *gray, const float *gx, int P, int H, int W, int lognb, unsigned *mags,
                         int n_angles, int discard_sat, const FftPlan *pl);

int pb_launch_rows_fixed(pb_ctx *ctx, const float *in, int C, float *gray, float *gx, float2 *part, long images, int H, int W, int nth,
                         const FftPlan *pl);
