// Internal launch interface of isp_elementwise.hip (used by isp_api.hip).
#pragma once
#include "isp_common.h"

namespace ew {

struct PtrList { const void* p[64]; };

// finalize modes: combine per-block partials into FrameParams / the ISP state
enum FinMode {
  FIN_BOUNDS = 0,      // partial rows {min,max}            -> FP_LO, FP_HI, FP_INV
  FIN_STATS = 1,       // rows {gmin,gmax,slog2,sgray,s0,s1,s2}, n px -> stateless metering + Reinhard scalars
  FIN_BOUNDS2 = 2,     // {min,max}                          -> FP_LO2, FP_HI2, FP_INV2
  FIN_MAXOUT = 3,      // {-,max}                            -> FP_MAXOUT = max(1e-6, max)
  FIN_ISP_BOUNDS = 4,  // {min,max} + state9, alpha          -> FP_LO/FP_HI = blended bounds (+raw to out)
  FIN_ISP_STATS = 5,   // 7 rows + blended bounds + state9   -> state9 updated (camera_isp.py:164-166)
  FIN_ISP_SUMS = 6,    // 7 rows                             -> out8 = [lmin,lmax,sum_log,sum_gray,sr,sg,sb,n]
  FIN_RAW_BOUNDS = 7   // {min,max}                          -> out2 raw
};

struct FinArgs {
  const float* partials; int stride; int nblocks;
  float* fp;            // FrameParams
  float* state9;        // ISP state (in/out) or NULL
  const float* bounds_in;  // FIN_ISP_SUMS/FIN_ISP_STATS: blended bounds (device) or NULL -> fp
  float* out;           // raw outputs (FIN_ISP_SUMS, FIN_RAW_BOUNDS)
  float n_px;           // pixel count for the means
  float alpha;          // ISP lerp weight
  float intensity, la, ca;
  int bounds_post;      // FIN_BOUNDS: 0 = bounds are final; 1 = clamp to [0,1]; 2 = clamp, then round to f16
};
int finalize(int mode, const FinArgs& a, hipStream_t s);

// ISP reinhard scalars from state9 -> FrameParams (camera_isp.py:186-195)
int isp_reinhard_prep(const float* state9, float* fp, float intensity, float ca, hipStream_t s);

}  // namespace ew
