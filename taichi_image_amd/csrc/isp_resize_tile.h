// Fused unpack -> demosaic (+colour matrix) -> bilinear resize: ISP.load_packed12/16 with
// resize_width / scale (camera_isp.py:333-347,371-373,302-315) in ONE pass over the packed frame.
// The reference materialises the f16 CFA (25 MB) and the full-resolution f16 RGB (75 MB) of a 4K
// frame; here neither exists.
//
// One 256-thread block produces a 64x16 tile of DESTINATION pixels:
//   1. the source CFA region its bilinear taps need (+2 px demosaic halo) is unpacked into LDS (fp32
//      holding the work-dtype values), 8 pixels = 12/16 packed bytes per lane and unit;
//   2. every lane owns 4 destination pixels.  The 4 taps of a pixel are a 2x2 source quad, which
//      always contains one site of each kernel K0..K3: each lane evaluates the four 13-tap kernels
//      with COMPILE-TIME weights at run-time LDS addresses (no divergence on the site parity), rounds
//      to the work dtype exactly where the reference stores its RGB image, and blends.
// Arithmetic: bit-exact against oracle isp_load_packed12 (same tap order, correctly rounded
// border renormalisation, p = I/scale by true division, mix(x, y, a) = x(1-a) + ya uncontracted).
#pragma once
#include "isp_tile.h"

#pragma clang fp contract(off)

namespace rtile {

using tile::KW;
using tile::Params;
using tile::TAP_DC;
using tile::TAP_DR;
using tile::static_for;

constexpr int DW = 64, DH = 16;                  // destination tile
constexpr int THREADS = 256;
constexpr int R_CAP = 48, C_CAP = 176;           // source rows / columns (8-aligned) the LDS tile can hold
constexpr int PITCH = C_CAP + 4;                 // floats; rows stay 16-byte aligned

struct RParams {
  Params t;                 // source description, colour matrix, weights (tile::Params)
  int Hd, Wd;
  float s0, s1;             // interpolate.py:60-66: p = (r / s0, c / s1)
};

// largest source extent a destination tile can need at these scales (host-side admission test)
static inline bool scales_fit(float s0, float s1) {
  const int rows = (int)((DH - 1) / s0) + 1 + 6 + 1;
  const int cols = (int)((DW - 1) / s1) + 1 + 6 + 1 + 7;
  return s0 > 0.f && s1 > 0.f && rows <= R_CAP && cols <= C_CAP;
}

// source quad origin of destination index i along one axis: min(trunc(i / s), n - 2)
MI_DEV int quad_origin(int i, float s, int n) {
  const int q = (int)((float)i / s);
  return q < n - 2 ? q : n - 2;
}

// One demosaiced source pixel (bayer.py:138-155) at run-time position (rr, cc): kernel KIDX,
// taps from the LDS region whose element (0, 0) is image (rb, cb).  The weights are literals:
// an SGPR operand would halve the issue rate of every FMA (scratch/issue_bench.hip).
// BORDER = false: the caller knows that no lane of the wave is near the image frame (and in_scale == 1).
template <int KIDX, bool EXACT, bool BORDER = true>
MI_DEV void demosaic_at(const Params& p, const float* lds, int rb, int cb, int rr, int cc, float (&rgb)[3]) {
  const float* ctr = lds + (rr - rb) * PITCH + (cc - cb);
  float acc[3];
  bool first[3] = {true, true, true};
  static_for<0, 13>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const float x = ctr[TAP_DR[t] * PITCH + TAP_DC[t]];
    static_for<0, 3>([&](auto chc) {
      constexpr int ch = decltype(chc)::value;
      constexpr int wi = KW[KIDX][t][ch];
      if constexpr (wi != 0) {
        constexpr float w = (float)wi * 0.0625f;
        if (first[ch]) { acc[ch] = x * w; first[ch] = false; }
        else if constexpr (EXACT) acc[ch] = __builtin_fmaf(x, w, acc[ch]);
        else acc[ch] = acc[ch] + x * w;
      }
    });
  });
  // border pixels / CFAs with scale != 1: (acc * 16) / (in_scale * t), as in the tile kernel
  const bool need = BORDER && (p.in_scale != 1.f || rr < 2 || rr >= p.H - 2 || cc < 2 || cc >= p.W - 2);
  if (BORDER && __builtin_amdgcn_ballot_w64(need) != 0) {
    float t3[3];
    tile::border_weight<KIDX>(rr, cc, p.H, p.W, t3);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float fixed = (acc[ch] * 16.f) / (p.in_scale * t3[ch]);
      acc[ch] = need ? fixed : acc[ch];
    }
  }
  if (p.has_ccm) {
    const float a = acc[0], b = acc[1], d = acc[2];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
      acc[ch] = (p.ccm[3 * ch] * a + p.ccm[3 * ch + 1] * b) + p.ccm[3 * ch + 2] * d;
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) rgb[ch] = clamp01(acc[ch]);
}

// u / n for 0 <= u < 2^16 and 1 <= n <= 64 without an integer division
MI_DEV int small_div(int u, float inv_n) { return (int)(((float)u + 0.5f) * inv_n); }

constexpr int FILL_NIT = (R_CAP * (C_CAP / 8) + THREADS - 1) / THREADS;

// HOT: 12-bit standard packing with aligned rows, no colour matrix (the configuration of
// Camera16/32.load_packed12), fixed at compile time; see tile::tile_kernel.
template <class E, int PR, int PC, bool HOT = false>
__global__ __launch_bounds__(THREADS) void resize_tile_kernel(const RParams rp_in) {
  constexpr bool EXACT = sizeof(E) == 2;
  RParams rp = rp_in;
  Params& p = rp.t;
  if constexpr (HOT) { p.src_kind = tile::SRC_PACKED12; p.src_fast = 1; p.has_ccm = 0; p.in_scale = 1.f; }
  __shared__ __attribute__((aligned(16))) float lds[R_CAP * PITCH];

  const int tiles_x = (rp.Wd + DW - 1) / DW;
  const int by = blockIdx.x / tiles_x, bx = blockIdx.x - by * tiles_x;
  const int r0d = by * DH, c0d = bx * DW;
  const int r1d = min(r0d + DH, rp.Hd) - 1, c1d = min(c0d + DW, rp.Wd) - 1;     // last dst row / col of the tile

  // source region: quads of the first / last destination row and column, +-2 px of demosaic halo
  const int rb = quad_origin(r0d, rp.s0, p.H) - 2;
  const int re = quad_origin(r1d, rp.s0, p.H) + 1 + 2;
  const int c_lo = quad_origin(c0d, rp.s1, p.W) - 2;
  const int cb = ((c_lo + 8192) & ~7) - 8192;                                    // floor to a multiple of 8
  const int ce = quad_origin(c1d, rp.s1, p.W) + 1 + 2;
  const int nrows = re - rb + 1, nunits = (ce - cb) / 8 + 1;                     // <= R_CAP, <= C_CAP / 8 (host-checked)
  const float inv_units = 1.0f / (float)nunits;

  // ---- fill: unpack the region into LDS (zeros outside the image) -----------------------------
  if constexpr (HOT) {
    // all loads of the lane first (one memory latency per tile), then unpack
    const uint8_t* base = static_cast<const uint8_t*>(p.src);
    const size_t pitch = (size_t)p.W * 3 / 2;
    uint32_t raw[FILL_NIT][3];
#pragma unroll
    for (int it = 0; it < FILL_NIT; ++it) {
      const int u = threadIdx.x + it * THREADS;
      const int lr = small_div(u, inv_units), lu = u - lr * nunits;
      const int r = rb + lr, c = cb + lu * 8;
      raw[it][0] = raw[it][1] = raw[it][2] = 0;
      if (u < nrows * nunits && r >= 0 && r < p.H && c >= 0 && c < p.W) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(base + (size_t)r * pitch + (size_t)c * 3 / 2);
        raw[it][0] = q[0]; raw[it][1] = q[1]; raw[it][2] = q[2];
      }
    }
#pragma unroll
    for (int it = 0; it < FILL_NIT; ++it) {
      const int u = threadIdx.x + it * THREADS;
      if (u >= nrows * nunits) continue;
      const int lr = small_div(u, inv_units), lu = u - lr * nunits;
      uint32_t v[8];
      tile::unpack12x8(raw[it][0], raw[it][1], raw[it][2], false, v);
      float out[8];
      tile::decode_scaled8<E>(v, p.k_decode, out);
      float* d = lds + lr * PITCH + lu * 8;
      *reinterpret_cast<float4*>(d) = make_float4(out[0], out[1], out[2], out[3]);
      *reinterpret_cast<float4*>(d + 4) = make_float4(out[4], out[5], out[6], out[7]);
    }
  } else {
    const uint8_t* base = static_cast<const uint8_t*>(p.src);
    const bool is16 = p.src_kind == tile::SRC_PACKED16;
    const bool ids = p.src_kind == tile::SRC_PACKED12_IDS;
    const size_t pitch = is16 ? (size_t)p.W * 2 : (size_t)p.W * 3 / 2;
    for (int u = threadIdx.x; u < nrows * nunits; u += THREADS) {
      const int lr = small_div(u, inv_units), lu = u - lr * nunits;
      const int r = rb + lr, c = cb + lu * 8;
      uint32_t v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0;
      if (r >= 0 && r < p.H && c >= 0 && c < p.W) {
        const uint8_t* rowp = base + (size_t)r * pitch;
        if (p.src_fast) {                              // W % 8 == 0: the unit is wholly inside
          if (!is16) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(rowp + (size_t)c * 3 / 2);
            tile::unpack12x8(q[0], q[1], q[2], ids, v);
          } else {
            const uint4 d = *reinterpret_cast<const uint4*>(rowp + (size_t)c * 2);
            v[0] = d.x & 0xFFFFu; v[1] = d.x >> 16; v[2] = d.y & 0xFFFFu; v[3] = d.y >> 16;
            v[4] = d.z & 0xFFFFu; v[5] = d.z >> 16; v[6] = d.w & 0xFFFFu; v[7] = d.w >> 16;
          }
        } else {
          const int n = c + 8 <= p.W ? 8 : p.W - c;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (i < n) {
              if (!is16) {
                const uint8_t* q = rowp + (size_t)(c + (i & ~1)) * 3 / 2;
                const uint32_t w = q[0] | (q[1] << 8) | (q[2] << 16);
                uint32_t a, b;
                tile::unpack_pair(w, ids, a, b);
                v[i] = (i & 1) ? b : a;
              } else {
                const uint8_t* q = rowp + (size_t)(c + i) * 2;
                v[i] = q[0] | (q[1] << 8);
              }
            }
          }
        }
      }
      float out[8];
      tile::decode_scaled8<E>(v, p.k_decode, out);
      float* d = lds + lr * PITCH + lu * 8;
      *reinterpret_cast<float4*>(d) = make_float4(out[0], out[1], out[2], out[3]);
      *reinterpret_cast<float4*>(d + 4) = make_float4(out[4], out[5], out[6], out[7]);
    }
  }
  __syncthreads();

  // ---- every lane: 4 consecutive destination pixels of one row (16 lanes across, 16 rows) --------
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int r = r0d + ty, cbase = c0d + 4 * tx;
  // dead lanes compute a valid in-tile pixel (the ballots in demosaic_at are wave-wide) and skip the store
  const int rr_d = r < rp.Hd ? r : r0d;
  // sample_bilinear (interpolate.py:24-34), row part
  const float pr = (float)rr_d / rp.s0;
  const int ir = (int)pr;
  const float fr = pr - (float)ir;
  const int ra = min(ir, p.H - 1), rbm = min(ir + 1, p.H - 1);                    // index_clamped (:20-21)
  const int qr = min(ir, p.H - 2);                                                // the 2x2 quad holding all taps
  const int par_r = (qr + PR) & 1;
  const int row0 = qr + par_r, row1 = qr + 1 - par_r;                             // (row + PR) & 1 == 0 / 1
  const bool ra1 = ((ra + PR) & 1) != 0, rb1 = ((rbm + PR) & 1) != 0;            // site parity of each row tap
  float o[4][3];
  static_for<0, 4>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int c = cbase + j;
    const int cc_d = (r < rp.Hd && c < rp.Wd) ? c : c0d;
    const float pc = (float)cc_d / rp.s1;
    const int ic = (int)pc;
    const float fc = pc - (float)ic;
    const int ca = min(ic, p.W - 1), cbm = min(ic + 1, p.W - 1);
    const int qc = min(ic, p.W - 2);
    const int par_c = (qc + PC) & 1;
    const int col0 = qc + par_c, col1 = qc + 1 - par_c;
    float P[2][2][3];                                                            // [row parity][col parity][rgb]
    // one frame test per quad (its four sites reach rows qr-2..qr+3, columns qc-2..qc+3), one scalar branch
    const bool near_frame = p.in_scale != 1.f || qr < 2 || qr + 1 >= p.H - 2 || qc < 2 || qc + 1 >= p.W - 2;
    if (__builtin_amdgcn_ballot_w64(near_frame) != 0) {
      demosaic_at<0, EXACT, true>(p, lds, rb, cb, row0, col0, P[0][0]);
      demosaic_at<1, EXACT, true>(p, lds, rb, cb, row1, col0, P[1][0]);
      demosaic_at<2, EXACT, true>(p, lds, rb, cb, row0, col1, P[0][1]);
      demosaic_at<3, EXACT, true>(p, lds, rb, cb, row1, col1, P[1][1]);
    } else {
      demosaic_at<0, EXACT, false>(p, lds, rb, cb, row0, col0, P[0][0]);
      demosaic_at<1, EXACT, false>(p, lds, rb, cb, row1, col0, P[1][0]);
      demosaic_at<2, EXACT, false>(p, lds, rb, cb, row0, col1, P[0][1]);
      demosaic_at<3, EXACT, false>(p, lds, rb, cb, row1, col1, P[1][1]);
    }
    const bool ca1 = ((ca + PC) & 1) != 0, cb1 = ((cbm + PC) & 1) != 0;          // site parity of each column tap
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      // the reference stores the full-resolution RGB in the work dtype (scale 1) before resizing
      const float p00 = (float)cast_out<E>(P[0][0][ch]), p10 = (float)cast_out<E>(P[1][0][ch]);
      const float p01 = (float)cast_out<E>(P[0][1][ch]), p11 = (float)cast_out<E>(P[1][1][ch]);
      // rows first: the values at rows ra / rb for each column parity, mixed over rows (:28-33); then
      // the column taps pick their parity.  Same operations on the same operands as mixing
      // src[ra, ca], src[rb, ca] and src[ra, cb], src[rb, cb] directly, with half the selects.
      const float a0 = ra1 ? p10 : p00, b0 = rb1 ? p10 : p00;                    // column parity 0
      const float a1 = ra1 ? p11 : p01, b1 = rb1 ? p11 : p01;                    // column parity 1
      const float m0 = a0 * (1.0f - fr) + b0 * fr;
      const float m1 = a1 * (1.0f - fr) + b1 * fr;
      const float y1 = ca1 ? m1 : m0;                                            // column ca
      const float y2 = cb1 ? m1 : m0;                                            // column cb
      o[j][ch] = y1 * (1.0f - fc) + y2 * fc;                                     // intensity scale 1 (same dtype)
    }
  });
  if (r < rp.Hd) {
    E* dst = static_cast<E*>(p.dst) + ((size_t)r * rp.Wd + cbase) * 3;
    E t[12];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) t[3 * j + ch] = cast_out<E>(o[j][ch]);
    // 4 px = 24 / 48 contiguous bytes: vector stores when the row pitch keeps them aligned
    const bool vec = cbase + 4 <= rp.Wd && (rp.Wd & 3) == 0 && ((uintptr_t)p.dst & 15) == 0;
    if (vec) {
      if constexpr (sizeof(E) == 2) {
        const uint2* sv = reinterpret_cast<const uint2*>(t);
        uint2* dv = reinterpret_cast<uint2*>(dst);
        dv[0] = sv[0]; dv[1] = sv[1]; dv[2] = sv[2];
      } else {
        const uint4* sv = reinterpret_cast<const uint4*>(t);
        uint4* dv = reinterpret_cast<uint4*>(dst);
        dv[0] = sv[0]; dv[1] = sv[1]; dv[2] = sv[2];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (cbase + j < rp.Wd) { dst[3 * j] = t[3 * j]; dst[3 * j + 1] = t[3 * j + 1]; dst[3 * j + 2] = t[3 * j + 2]; }
    }
  }
}

static inline bool hot_ok(const RParams& rp) {
  const Params& p = rp.t;
  return p.src_kind == tile::SRC_PACKED12 && p.src_fast && !p.has_ccm && p.in_scale == 1.f;
}

static inline int num_tiles(int Hd, int Wd) { return ((Wd + DW - 1) / DW) * ((Hd + DH - 1) / DH); }

int launch_rggb(const RParams& p, int work_dtype, hipStream_t s);
int launch_grbg(const RParams& p, int work_dtype, hipStream_t s);
int launch_gbrg(const RParams& p, int work_dtype, hipStream_t s);
int launch_bggr(const RParams& p, int work_dtype, hipStream_t s);
static inline int launch(const RParams& p, int work_dtype, int pattern, hipStream_t s) {
  switch (pattern) {
    case MI_RGGB: return launch_rggb(p, work_dtype, s);
    case MI_GRBG: return launch_grbg(p, work_dtype, s);
    case MI_GBRG: return launch_gbrg(p, work_dtype, s);
    default: return launch_bggr(p, work_dtype, s);
  }
}

}  // namespace rtile
