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
// taps from the LDS region whose element (0, 0) is image (rb, cb).
template <int KIDX, bool EXACT>
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
        const float w = p.wq[tile::wq_index(wi)];
        if (first[ch]) { acc[ch] = x * w; first[ch] = false; }
        else if constexpr (EXACT) acc[ch] = __builtin_fmaf(x, w, acc[ch]);
        else acc[ch] = acc[ch] + x * w;
      }
    });
  });
  // border pixels / CFAs with scale != 1: (acc * 16) / (in_scale * t), as in the tile kernel
  const bool need = p.in_scale != 1.f || rr < 2 || rr >= p.H - 2 || cc < 2 || cc >= p.W - 2;
  if (__builtin_amdgcn_ballot_w64(need) != 0) {
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

template <class E, int PR, int PC>
__global__ __launch_bounds__(THREADS) void resize_tile_kernel(const RParams rp) {
  constexpr bool EXACT = sizeof(E) == 2;
  const Params& p = rp.t;
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

  // ---- fill: unpack the region into LDS (zeros outside the image) -----------------------------
  {
    const uint8_t* base = static_cast<const uint8_t*>(p.src);
    const bool is16 = p.src_kind == tile::SRC_PACKED16;
    const bool ids = p.src_kind == tile::SRC_PACKED12_IDS;
    const size_t pitch = is16 ? (size_t)p.W * 2 : (size_t)p.W * 3 / 2;
    for (int u = threadIdx.x; u < nrows * nunits; u += THREADS) {
      const int lr = u / nunits, lu = u - lr * nunits;
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
#pragma unroll
      for (int i = 0; i < 8; ++i) out[i] = tile::decode_scaled<E>(v[i], p.k_decode);
      float* d = lds + lr * PITCH + lu * 8;
      *reinterpret_cast<float4*>(d) = make_float4(out[0], out[1], out[2], out[3]);
      *reinterpret_cast<float4*>(d + 4) = make_float4(out[4], out[5], out[6], out[7]);
    }
  }
  __syncthreads();

  // ---- 4 destination pixels per lane: consecutive lanes = consecutive columns of one row -------
  E* dst = static_cast<E*>(p.dst);
#pragma unroll 1
  for (int j = 0; j < DW * DH / THREADS; ++j) {
    const int idx = threadIdx.x + j * THREADS;
    const int r = r0d + idx / DW, c = c0d + (idx & (DW - 1));
    const bool live = r < rp.Hd && c < rp.Wd;
    // dead lanes compute a valid in-tile pixel (the ballots below are wave-wide) and skip the store
    const int rr_d = live ? r : r0d, cc_d = live ? c : c0d;
    // sample_bilinear (interpolate.py:24-34)
    const float pr = (float)rr_d / rp.s0, pc = (float)cc_d / rp.s1;
    const int ir = (int)pr, ic = (int)pc;
    const float fr = pr - (float)ir, fc = pc - (float)ic;
    const int ra = min(ir, p.H - 1), rbm = min(ir + 1, p.H - 1);                  // index_clamped (:20-21)
    const int ca = min(ic, p.W - 1), cbm = min(ic + 1, p.W - 1);
    const int qr = min(ir, p.H - 2), qc = min(ic, p.W - 2);                      // the 2x2 quad holding all taps
    // the quad's row / column with site parity 0 and 1
    const int par_r = (qr + PR) & 1, par_c = (qc + PC) & 1;
    const int row0 = qr + par_r, row1 = qr + 1 - par_r;                           // (row + PR) & 1 == 0 / 1
    const int col0 = qc + par_c, col1 = qc + 1 - par_c;
    float P[2][2][3];                                                            // [row parity][col parity][rgb]
    demosaic_at<0, EXACT>(p, lds, rb, cb, row0, col0, P[0][0]);
    demosaic_at<1, EXACT>(p, lds, rb, cb, row1, col0, P[1][0]);
    demosaic_at<2, EXACT>(p, lds, rb, cb, row0, col1, P[0][1]);
    demosaic_at<3, EXACT>(p, lds, rb, cb, row1, col1, P[1][1]);
    // the reference stores the full-resolution RGB in the work dtype (scale 1) before resizing
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) P[a][b][ch] = (float)cast_out<E>(P[a][b][ch]);
    const bool ra1 = ((ra + PR) & 1) != 0, rb1 = ((rbm + PR) & 1) != 0;          // site parity of each tap
    const bool ca1 = ((ca + PC) & 1) != 0, cb1 = ((cbm + PC) & 1) != 0;
    float o[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float p00 = P[0][0][ch], p10 = P[1][0][ch], p01 = P[0][1][ch], p11 = P[1][1][ch];
      const float ta_ca = ra1 ? (ca1 ? p11 : p10) : (ca1 ? p01 : p00);           // src[ra, ca]
      const float tb_ca = rb1 ? (ca1 ? p11 : p10) : (ca1 ? p01 : p00);           // src[rb, ca]
      const float ta_cb = ra1 ? (cb1 ? p11 : p10) : (cb1 ? p01 : p00);           // src[ra, cb]
      const float tb_cb = rb1 ? (cb1 ? p11 : p10) : (cb1 ? p01 : p00);           // src[rb, cb]
      const float y1 = ta_ca * (1.0f - fr) + tb_ca * fr;                         // mix over rows (:28-33)
      const float y2 = ta_cb * (1.0f - fr) + tb_cb * fr;
      o[ch] = y1 * (1.0f - fc) + y2 * fc;                                        // intensity scale 1 (same dtype)
    }
    if (live) {
      E* q = dst + ((size_t)r * rp.Wd + c) * 3;
      q[0] = cast_out<E>(o[0]); q[1] = cast_out<E>(o[1]); q[2] = cast_out<E>(o[2]);
    }
  }
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
