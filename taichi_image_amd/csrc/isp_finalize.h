// Finalize step shared by finalize_kernel (isp_elementwise.hip) and the last-arriver block of the
// persistent tile kernel (isp_tile.h): turns the reduced partials into the scalars of the next pass.
#pragma once
#include "isp_common.h"

namespace ew {

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
  const float* state9_in;  // the previous state when it is not to be overwritten (NULL: state9 itself, in place)
  const float* bounds_in;  // FIN_ISP_SUMS/FIN_ISP_STATS: blended bounds (device) or NULL -> fp
  float* out;           // raw outputs (FIN_ISP_SUMS, FIN_RAW_BOUNDS)
  float n_px;           // pixel count for the means
  float alpha;          // ISP lerp weight
  float intensity, la, ca;
  int bounds_post;      // FIN_BOUNDS: 0 = bounds are final; 1 = clamp to [0,1]; 2 = clamp, then round to f16
};

// FrameParams entries a stateless finalize mode writes (what a pulled finalize publishes)
MI_DEV constexpr bool finalize_writes(int mode, int i) {
  return mode == FIN_BOUNDS ? (i == FP_LO || i == FP_HI || i == FP_INV)
         : mode == FIN_BOUNDS2 ? (i == FP_LO2 || i == FP_HI2 || i == FP_INV2)
         : mode == FIN_STATS ? (i == FP_BMIN || i == FP_BMAX || i == FP_LMEAN || i == FP_GMEAN ||
                                (i >= FP_RMEAN && i < FP_RMEAN + 3) || i == FP_MAPKEY || i == FP_EI ||
                                (i >= FP_MEAN3 && i < FP_MEAN3 + 3))
                             : false;
}

// tot: row 0 = min, row 1 = max, rows 2..6 = sums (fp64) over all blocks.  One thread runs this.
// HW: log / pow / exp on the hardware units (v_log_f32 / v_exp_f32, 1 ulp) instead of the libm-grade
// routines: used by the pulled finalize, where every block of the consuming pass waits for this thread
// (a few dozen cycles instead of ~2000).  The scalars differ by ~1e-6 relative, inside the tonemap
// tolerance of the parity contract.
template <bool HW = false>
MI_DEV void finalize_scalars(int mode, const FinArgs& a, const double* tot) {
  float lo = (float)tot[0], hi = (float)tot[1];
  if (mode == FIN_BOUNDS && a.bounds_post > 0) {
    // tile bounds pass: reduced before the clamp / work-dtype rounding of bayer.py:155,134
    lo = fminf(fmaxf(lo, 0.f), 1.f); hi = fminf(fmaxf(hi, 0.f), 1.f);
    if (a.bounds_post > 1) { lo = (float)(half_t)lo; hi = (float)(half_t)hi; }
  }
  float* fp = a.fp;
  switch (mode) {
    case FIN_BOUNDS:
      fp[FP_LO] = lo; fp[FP_HI] = hi; fp[FP_INV] = 1.0f / (hi - lo);   // tonemap.py:13
      break;
    case FIN_BOUNDS2:
      fp[FP_LO2] = lo; fp[FP_HI2] = hi; fp[FP_INV2] = 1.0f / (hi - lo);
      break;
    case FIN_MAXOUT:
      fp[FP_MAXOUT] = fmaxf(1e-6f, hi);                               // camera_isp.py:190,213
      break;
    case FIN_RAW_BOUNDS:
      a.out[0] = lo; a.out[1] = hi;
      break;
    case FIN_ISP_BOUNDS: {
      // camera_isp.py:156-157: b = lerp(alpha, new, prev) = new + alpha * (prev - new)
      const float* prev = a.state9_in ? a.state9_in : a.state9;
      const float pmin = prev[0], pmax = prev[1];
      fp[FP_LO] = lo + a.alpha * (pmin - lo);
      fp[FP_HI] = hi + a.alpha * (pmax - hi);
      break;
    }
    default: {
      const float LN2 = 0.6931471805599453f;
      // lo/hi here are min/max of max(gray,1e-4); producers that post the raw gray bounds (the whole-frame kernel) rely
      // on this clamp (monotone, so max of the reduced value == reduced max), for the others it is the identity
      lo = fmaxf(lo, 1e-4f); hi = fmaxf(hi, 1e-4f);
      const float lmin = HW ? __builtin_amdgcn_logf(lo) * LN2 : logf(lo);
      const float lmax = HW ? __builtin_amdgcn_logf(hi) * LN2 : logf(hi);
      const float slog = (float)(tot[2] * 0.6931471805599453);
      const float sgray = (float)tot[3];
      const float s0 = (float)tot[4], s1 = (float)tot[5], s2 = (float)tot[6];
      if (mode == FIN_ISP_SUMS) {
        a.out[0] = lmin; a.out[1] = lmax; a.out[2] = slog; a.out[3] = sgray;
        a.out[4] = s0; a.out[5] = s1; a.out[6] = s2; a.out[7] = a.n_px;
      } else if (mode == FIN_ISP_STATS) {
        // camera_isp.py:131-134,164-166
        const float n = a.n_px;
        const float* b = a.bounds_in ? a.bounds_in : fp + FP_LO;
        const float v[9] = {b[0], b[1], lmin, lmax, slog / n, sgray / n, s0 / n, s1 / n, s2 / n};
        const float* prev = a.state9_in ? a.state9_in : a.state9;
        for (int i = 0; i < 9; ++i) a.state9[i] = v[i] + a.alpha * (prev[i] - v[i]);
      } else {
        // tonemap.py:99-103 (log_bounds = (lmin, -lmax): reference sign quirk), :115-119
        const float n = a.n_px;
        const float Bmin = lmin, Bmax = -lmax;
        const float lmean = slog / n, gmean = sgray / n;
        const float rm[3] = {s0 / n, s1 / n, s2 / n};
        const float key = (Bmax - lmean) / (Bmax - Bmin);
        fp[FP_BMIN] = Bmin; fp[FP_BMAX] = Bmax; fp[FP_LMEAN] = lmean; fp[FP_GMEAN] = gmean;
        fp[FP_RMEAN] = rm[0]; fp[FP_RMEAN + 1] = rm[1]; fp[FP_RMEAN + 2] = rm[2];
        fp[FP_MAPKEY] = 0.3f + 0.7f * (HW ? __builtin_amdgcn_exp2f(1.4f * __builtin_amdgcn_logf(key)) : powf(key, 1.4f));
        fp[FP_EI] = HW ? __builtin_amdgcn_exp2f(-a.intensity * 1.4426950408889634f) : expf(-a.intensity);
        for (int c = 0; c < 3; ++c) fp[FP_MEAN3 + c] = gmean + a.ca * (rm[c] - gmean);
      }
      break;
    }
  }
}

// The stateless finalize modes (FIN_BOUNDS, FIN_STATS, FIN_BOUNDS2) for the whole-frame kernel's grid barriers, where
// every wave of the chip waits for the ONE lane that runs this: reciprocals on the hardware unit instead of IEEE
// division sequences (seven of them in a row cost 0.7 us of the 0.9 us this step took), one reciprocal of n for the five
// means, no fp64.  v_rcp_f32 is exact for powers of two, so bounds of exactly (0, 1) still give inv == 1 - the test the
// kernel's shortcut rests on; elsewhere the scalars differ from finalize_scalars by ~1e-7 relative (contract: 1e-4).
// tot: as finalize_scalars, already narrowed to fp32.
MI_DEV void finalize_scalars_fast(int mode, const FinArgs& a, const float* tot) {
  float lo = tot[0], hi = tot[1];
  float* fp = a.fp;
  if (mode == FIN_BOUNDS) {
    if (a.bounds_post > 0) {
      lo = fminf(fmaxf(lo, 0.f), 1.f); hi = fminf(fmaxf(hi, 0.f), 1.f);
      if (a.bounds_post > 1) { lo = (float)(half_t)lo; hi = (float)(half_t)hi; }
    }
    fp[FP_LO] = lo; fp[FP_HI] = hi; fp[FP_INV] = __builtin_amdgcn_rcpf(hi - lo);   // tonemap.py:13
  } else if (mode == FIN_BOUNDS2) {
    fp[FP_LO2] = lo; fp[FP_HI2] = hi; fp[FP_INV2] = __builtin_amdgcn_rcpf(hi - lo);
  } else if (mode == FIN_MAXOUT) {
    fp[FP_MAXOUT] = fmaxf(1e-6f, hi);                                               // camera_isp.py:190,213
  } else {                                                                          // FIN_STATS: tonemap.py:99-103, :115-119
    const float LN2 = 0.6931471805599453f;
    lo = fmaxf(lo, 1e-4f); hi = fmaxf(hi, 1e-4f);
    const float Bmin = __builtin_amdgcn_logf(lo) * LN2, Bmax = -(__builtin_amdgcn_logf(hi) * LN2);
    const float rn = __builtin_amdgcn_rcpf(a.n_px);
    const float lmean = tot[2] * LN2 * rn, gmean = tot[3] * rn;
    const float key = (Bmax - lmean) * __builtin_amdgcn_rcpf(Bmax - Bmin);
    fp[FP_BMIN] = Bmin; fp[FP_BMAX] = Bmax; fp[FP_LMEAN] = lmean; fp[FP_GMEAN] = gmean;
    fp[FP_MAPKEY] = 0.3f + 0.7f * __builtin_amdgcn_exp2f(1.4f * __builtin_amdgcn_logf(key));
    fp[FP_EI] = __builtin_amdgcn_exp2f(-a.intensity * 1.4426950408889634f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float rm = tot[4 + c] * rn;
      fp[FP_RMEAN + c] = rm;
      fp[FP_MEAN3 + c] = gmean + a.ca * (rm - gmean);
    }
  }
}

}  // namespace ew
