// Per-pixel tonemap arithmetic shared by the fused tile kernels and the elementwise kernels.
// Restates tonemap.py:12-17,78-131 and camera_isp.py:117-128,186-218 of the reference.
// Transcendentals use the gfx950 hardware units (v_log_f32 / v_exp_f32 / v_rcp_f32, <= 1 ulp):
// these stages carry the 1e-4 relative tolerance of the parity contract, not bit-exactness.
#pragma once
#include "isp_common.h"

#pragma clang fp contract(fast)

MI_DEV float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
MI_DEV float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
MI_DEV float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// pow for the bases that occur here: b == 0 -> 0 (e > 0), b < 0 -> NaN, like powf.
MI_DEV float hw_pow(float b, float e) { return hw_exp2(e * hw_log2(b)); }
MI_DEV float hw_log(float x) { return hw_log2(x) * 0.6931471805599453f; }

// color/__init__.py:7-10
// Written with explicit fmas (round 4): under `contract(fast)` WHICH of the two products of (r .299 + g .587) is fused is
// the compiler's choice per instantiation, and two kernels that must derive the same bits from the same pixel - pass 1 of
// the ISP Reinhard and the recomputing pass 2 of tonemap_reinhard(write_back=False) - differed in the last bit of the gray.
// This is the form the compiler had picked in the whole-frame kernel: g .587 first, then r, then b.
MI_DEV float rgb_gray(float r, float g, float b) { return __builtin_fmaf(b, 0.114f, __builtin_fmaf(r, 0.299f, g * 0.587f)); }

// clamp to [0, 1] in one instruction (v_med3_f32; a NaN input yields min3 of the others = 0)
MI_DEV float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 1.f); }

// tonemap.py:12-17 with gamma == 1 (linear_func into the f32 temp): clamp((x-lo)*inv, 0, 1)
MI_DEV float norm01(float x, float lo, float inv) { return clamp01((x - lo) * inv); }

// Running statistics of tonemap.py:78-103 / camera_isp.py:117-128.  min/max are taken on
// max(gray, 1e-4) and the (monotone) log is applied once when the partials are combined.
struct StatsAcc {
  float gmin, gmax, slog, sgray, s0, s1, s2;
  MI_DEV void init() {
    gmin = __builtin_inff(); gmax = -__builtin_inff();
    slog = sgray = s0 = s1 = s2 = 0.f;
  }
  MI_DEV void add(float t0, float t1, float t2) {
    float g = rgb_gray(t0, t1, t2);
    float gc = fmaxf(g, 1e-4f);
    gmin = fminf(gmin, gc); gmax = fmaxf(gmax, gc);
    slog += hw_log2(gc);           // sum of log2; scaled by ln2 when combined
    sgray += g; s0 += t0; s1 += t1; s2 += t2;
  }
};

struct ReinhardK {
  float map_key, ei, mean3[3], la, ca;
};

// reinhard_func (tonemap.py:120-131) / camera_isp.py:200-210 on an already normalised pixel.
// CA0 (color_adapt == 0): adapt_color == gray and mean3 is the same for the three channels, so
// the three channels share one pow.  A template parameter, not a per-pixel test: the callers branch
// once per row of pixels (a scalar branch), keeping the pixel code straight-line.
// the adaptation term the three channels of a pixel share when color_adapt == 0 (tonemap.py:120-129)
MI_DEV float reinhard_adapt_ca0(const float (&t)[3], const ReinhardK& k) {
  const float g = rgb_gray(t[0], t[1], t[2]);
  const float am = __builtin_fmaf(k.la, g - k.mean3[0], k.mean3[0]);
  return hw_pow(k.ei * am, k.map_key);
}
// one channel: t / (ad + t)
MI_DEV float reinhard_map(float t, float ad) { return t * hw_rcp(ad + t); }

template <bool CA0>
MI_DEV void reinhard_px(const float (&t)[3], const ReinhardK& k, float (&out)[3]) {
  if constexpr (CA0) {
    const float ad = reinhard_adapt_ca0(t, k);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = reinhard_map(t[c], ad);
  } else {
    const float g = rgb_gray(t[0], t[1], t[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float ac = __builtin_fmaf(k.ca, t[c] - g, g);
      const float am = __builtin_fmaf(k.la, ac - k.mean3[c], k.mean3[c]);
      const float ad = hw_pow(k.ei * am, k.map_key);
      out[c] = t[c] * hw_rcp(ad + t[c]);
    }
  }
}

// camera_isp.py:200: the ISP's normalisation, no clamp.  The value is ROUNDED where it is made (f32_rounded: an empty asm the
// contraction pass cannot look through): under `contract(fast)` the product would otherwise be free to fuse with whatever
// adds to it downstream - `ad + t` of reinhard_map, `t - g` of the colour adaptation - in one kernel and not in another,
// and the kernels that map the same pixel (pass 1, the recomputing pass 2, the one-launch tonemap, the camera-group
// kernel of isp_mega_cam.h) must agree to the bit.  (Round 4: pass 1 had fused, the camera-group kernel had not - p
// differed in the last bit for ~1 pixel in 20 000 whenever the bounds were not exactly (0, 1).)
MI_DEV float isp_norm(float x, float lo, float inv) { return f32_rounded((x - lo) * inv); }

// tonemap.py:12-17: clamp(((x-lo)*inv)^(1/gamma), 0, 1) * scale.  gamma_inv == 1 skips the pow
// (powf(x, 1) == x exactly).  NaN -> 0 through fmaxf.
MI_DEV float linear_px(float x, float lo, float inv, float gamma_inv, float scale) {
  float v = (x - lo) * inv;
  if (gamma_inv != 1.f) v = hw_pow(v, gamma_inv);
  return fminf(fmaxf(v, 0.f), 1.f) * scale;
}

// The same for N values with ONE scalar branch on gamma.  Written per value, hipcc if-converts the
// branch into select(pow(v), v): a v_log + v_exp (quarter-rate) per value even when gamma == 1.
// The empty volatile asm makes the pow block non-speculatable, so it stays a real branch.
#ifndef MI_CENSUS_GAMMA            /* reading aid of scripts/isa_census.py: gamma == 1 as a constant */
#define MI_CENSUS_GAMMA(expr) (expr)
#endif
template <int N>
MI_DEV void linear_n(float (&v)[N], float lo, float inv, float gamma_inv, float scale) {
  if (MI_CENSUS_GAMMA(gamma_inv != 1.f)) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = clamp01(hw_pow((v[i] - lo) * inv, gamma_inv)) * scale;
  } else {
    // gamma == 1: clamp(x, 0, 1) * scale == med3(x * scale, 0, scale) up to one rounding
    const float inv_s = inv * scale;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_fmed3f((v[i] - lo) * inv_s, 0.f, scale);
  }
}

#pragma clang fp contract(off)

// camera_isp.py:186-195: the Reinhard scalars of the ISP path from the metering 9-vector (no contraction: the two
// kernels that derive them - rgb_pass_kernel<PM_ISP_RH_P1>, mega::camera_kernel - must get the same bits)
MI_DEV void isp_reinhard_scalars(const float* state9, float* fp, float intensity, float ca) {
  const float bmin = state9[0], bmax = state9[1], lmin = state9[2], lmax = state9[3];
  const float lmean = state9[4], mean = state9[5];
  const float key = (lmax - lmean) / (lmax - lmin);
  fp[FP_LO] = bmin; fp[FP_HI] = bmax; fp[FP_INV] = 1.0f / (bmax - bmin);
  fp[FP_MAPKEY] = 0.3f + 0.7f * powf(key, 1.4f);
  fp[FP_EI] = expf(-intensity);
  for (int c = 0; c < 3; ++c) fp[FP_MEAN3 + c] = mean + ca * (state9[6 + c] - mean);
}
