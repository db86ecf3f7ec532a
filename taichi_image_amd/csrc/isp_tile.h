// LDS-tiled Bayer demosaic (bayer.py:115-177 of the reference) with pluggable sources and
// epilogues.  One 256-thread block (4 wave64) produces a 128x32-pixel output tile:
//   1. fill : the CFA tile + 2-px halo is staged in LDS as fp32 holding the work-dtype values
//             (E = f16 or f32: the values are rounded to E exactly where the reference stores its
//             CFA image, then widened).  Packed 12/16-bit sources are unpacked on the way
//             (packed.py:24-44,149-157), 8 pixels (12 or 16 coalesced bytes) per lane, every load
//             of a lane issued before the first use;
//   2. strip: every lane owns a 2-row x 8-col strip, reads its 6x12 window from LDS once
//             (b64 + 2 x b128 + b64 per row) and evaluates the four 13-tap diamond kernels with
//             compile-time weights, accumulating in fp32 in the reference's tap order;
//   3. epilogue per strip row: store RGB (16-byte stores), or feed the tonemap reductions / final
//             map of the fused config-2 pipeline without ever writing the intermediate RGB image.
//
// Arithmetic contract: the demosaic is bit-exact against oracle/isp_oracle.py:bayer_to_rgb.
// The accumulation uses the weights w/16 (an exact power-of-two scaling that commutes with every
// rounding in the chain), so an interior pixel of a scale-1 CFA is finished after the last FMA;
// every other pixel is fixed up as (acc*16) / (in_scale * t) with a correctly rounded division.
// Products of <=16-bit inputs with the weights are exact in fp32, so fmaf == mul-then-add for
// E = f16; E = f32 keeps separate mul and add.
//
// Measured on gfx950 (scratch/fma_bench3.hip): the vector issue rate is ~1 wave-instruction/ns per
// SIMD only while the instruction stream stays under ~4.4 bytes/ns per SIMD, i.e. with 4-byte
// encodings; long bodies of 8-byte encodings run at ~0.55/ns.  The strip code is therefore written
// to compile to VOP2 forms (v_fmac_f32 with SGPR weights) and as few instructions as possible.
#pragma once
#include "isp_common.h"
#include "isp_math.h"

#pragma clang fp contract(off)

#include <type_traits>

namespace tile {

// measurement aid: Params::debug_skip switches parts of the tile kernel off (scripts/ablate*.py); release builds
// compile the tests away (make EXTRA=-DMI_ISP_MEASURE keeps them)
#ifdef MI_ISP_MEASURE
#define MI_DEBUG_SKIP(p) ((p).debug_skip)
#else
#define MI_DEBUG_SKIP(p) 0
#endif

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F> MI_DEV void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

constexpr int TILE_W = 128, TILE_H = 32;
constexpr int STRIP_W = 8, STRIP_H = 2;
constexpr int STRIPS_X = TILE_W / STRIP_W;            // 16 lanes across
constexpr int LDS_COLS = TILE_W + 16;                 // image cols [c0-8, c0+TILE_W+8)
constexpr int LDS_ROWS = TILE_H + 4;                  // image rows [r0-2, r0+TILE_H+2)
constexpr int UNITS = LDS_COLS / 8;                   // 8-px load units per LDS row
constexpr int THREADS = 256;
// LDS row layout: the even 16-byte slots (4 columns each) of a row live in [0, HALF), the odd
// slots in [HALF, 2*HALF).  The lanes of one ds_read_b128 / ds_write_b128 then touch consecutive
// 16-byte slots and two strip rows are a multiple of 64 banks apart: conflict-free for the
// hardware's 16-lane b128 groups (scratch/lds_sim.py; the linear layout cost 2.7x the cycles).
constexpr int HALF = 80;                              // 18 slots * 4 floats + 8 pad
constexpr int PITCH = 2 * HALF;                       // LDS row pitch in floats
__host__ __device__ constexpr int lds_pos(int lc) { return ((lc >> 3) << 2) + (lc & 3) + ((lc >> 2) & 1) * HALF; }

enum SrcKind { SRC_CFA_U8 = 0, SRC_CFA_U16 = 1, SRC_CFA_F16 = 2, SRC_CFA_F32 = 3,
               SRC_PACKED12 = 4, SRC_PACKED12_IDS = 5, SRC_PACKED16 = 6 };
// EPI_STORE_MINMAX: EPI_STORE that also reduces the bounds of what it stores (first pass of the
// "cached" config-2 pipeline: the work-dtype RGB image is written once and re-read by the
// elementwise tonemap passes instead of being recomputed three times).
enum Epi { EPI_STORE = 0, EPI_MINMAX = 1, EPI_STATS = 2, EPI_RH_MINMAX = 3, EPI_RH_STORE = 4, EPI_STORE_MINMAX = 5 };

struct Params {
  const void* src;        // CFA image or packed bytes
  void* dst;              // RGB output (EPI_STORE / EPI_RH_STORE)
  const float* fp;        // FrameParams (device) for the tonemap epilogues
  float* partials;        // per-block partials, SoA with stride part_stride
  int part_stride;
  int H, W;
  int src_kind;
  int src_fast;           // packed sources: rows are 4-byte (12-bit) / 16-byte (16-bit) aligned
  int out_dtype;
  int vec_store;          // dst rows allow 16-byte (8-byte for u8) vector stores
  int has_ccm;
  float ccm[9];
  float in_scale;         // scale_factor of the CFA dtype (types.py:12-18)
  float k_decode;         // f32(scale(E)/4095) or f32(scale(E)/65535) for packed sources
  float out_scale;        // scale_factor of the output dtype
  float gamma_inv, la, ca;
  int debug_skip;         // measurement aid: bit 0 skips the fill, bit 1 skips the strip compute
  // The eight distinct demosaic weights / 16 for the fused resize kernel (isp_resize_tile.h).  The
  // tile kernel uses literal weights instead: measured on gfx950 (scratch/issue_bench.hip), a VALU
  // instruction with an SGPR operand issues at half the rate of one with VGPR / literal operands
  // (v_fmac_f32 v, s, v: 0.57 wave-instr/ns/SIMD; v_fmac_f32 v, literal, v: 0.95).
  float wq[8];
};

// the distinct weight values of KW and their slot in Params::wq
__host__ __device__ constexpr int wq_index(int w) {
  return w == -3 ? 0 : w == -2 ? 1 : w == 1 ? 2 : w == 4 ? 3 : w == 8 ? 4 : w == 10 ? 5 : w == 12 ? 6 : 7;
}
constexpr int WQ_VALUES[8] = {-3, -2, 1, 4, 8, 10, 12, 16};
static inline void set_weights(Params& p) {
  for (int i = 0; i < 8; ++i) p.wq[i] = (float)WQ_VALUES[i] * 0.0625f;
}

// 13-tap diamond, reference order (bayer.py:15-27): (d_row, d_col)
__device__ constexpr int8_t TAP_DR[13] = {-2, -1, -1, -1, 0, 0, 0, 0, 0, 1, 1, 1, 2};
__device__ constexpr int8_t TAP_DC[13] = {0, -1, 0, 1, -2, -1, 0, 1, 2, -1, 0, 1, 0};
// weights [kernel][tap][channel] (bayer.py:30-55): K0 red site, K1 green (red above/below),
// K2 green (red left/right), K3 blue site.  Every channel sums to 16.
__device__ constexpr int8_t KW[4][13][3] = {
    {{0, -2, -3}, {0, 0, 4}, {0, 4, 0}, {0, 0, 4}, {0, -2, -3}, {0, 4, 0}, {16, 8, 12},
     {0, 4, 0}, {0, -2, -3}, {0, 0, 4}, {0, 4, 0}, {0, 0, 4}, {0, -2, -3}},
    {{-2, 0, 1}, {-2, 0, -2}, {8, 0, 0}, {-2, 0, -2}, {1, 0, -2}, {0, 0, 8}, {10, 16, 10},
     {0, 0, 8}, {1, 0, -2}, {-2, 0, -2}, {8, 0, 0}, {-2, 0, -2}, {-2, 0, 1}},
    {{1, 0, -2}, {-2, 0, -2}, {0, 0, 8}, {-2, 0, -2}, {-2, 0, 1}, {8, 0, 0}, {10, 16, 10},
     {8, 0, 0}, {-2, 0, 1}, {-2, 0, -2}, {0, 0, 8}, {-2, 0, -2}, {1, 0, -2}},
    {{-3, -2, 0}, {4, 0, 0}, {0, 4, 0}, {4, 0, 0}, {-3, -2, 0}, {0, 4, 0}, {12, 8, 16},
     {0, 4, 0}, {-3, -2, 0}, {4, 0, 0}, {0, 4, 0}, {4, 0, 0}, {-3, -2, 0}}};

// In-bounds weight sums of a border pixel (the `t` of bayer.py:143-149), tabulated at compile time:
// entry [kernel][row mask][col mask] packs t for R, G, B as three signed bytes, where bit (d + 2) of a
// mask says that offset d in {-2..2} is inside the image.  The sums are small integers (units of 1/16).
struct BorderTable { uint32_t t[4][32][32]; };
constexpr BorderTable make_border_table() {
  constexpr int DR[13] = {-2, -1, -1, -1, 0, 0, 0, 0, 0, 1, 1, 1, 2};
  constexpr int DC[13] = {0, -1, 0, 1, -2, -1, 0, 1, 2, -1, 0, 1, 0};
  BorderTable b = {};
  for (int k = 0; k < 4; ++k)
    for (int rm = 0; rm < 32; ++rm)
      for (int cm = 0; cm < 32; ++cm) {
        int t[3] = {0, 0, 0};
        for (int tap = 0; tap < 13; ++tap)
          if (((rm >> (DR[tap] + 2)) & 1) && ((cm >> (DC[tap] + 2)) & 1))
            for (int ch = 0; ch < 3; ++ch) t[ch] += KW[k][tap][ch];
        b.t[k][rm][cm] = (uint32_t)(t[0] & 0xFF) | ((uint32_t)(t[1] & 0xFF) << 8) | ((uint32_t)(t[2] & 0xFF) << 16);
      }
  return b;
}
__device__ constexpr BorderTable BORDER_T = make_border_table();

// 5-bit mask of the offsets d in {-2..2} for which x + d lies in [0, n)
MI_DEV int inside_mask(int x, int n) {
  int m = 0;
#pragma unroll
  for (int d = -2; d <= 2; ++d) m |= (x + d >= 0 && x + d < n) ? 1 << (d + 2) : 0;
  return m;
}

// ---------------------------------------------------------------------------------------------
// unpack helpers
// ---------------------------------------------------------------------------------------------
// One 24-bit group w = b0 | b1<<8 | b2<<16 -> two 12-bit values.
// standard (packed.py:24-31): p0 = (b1&0xF)<<8 | b0 = w & 0xFFF ; p1 = b2<<4 | b1>>4 = w >> 12
// IDS      (packed.py:37-44): p0 = b0<<4 | (b2&0xF)           ; p1 = b1<<4 | b2>>4
MI_DEV void unpack_pair(uint32_t w, bool ids, uint32_t& p0, uint32_t& p1) {
  if (!ids) {
    p0 = w & 0xFFFu;
    p1 = (w >> 12) & 0xFFFu;
  } else {
    const uint32_t b0 = w & 0xFFu, b1 = (w >> 8) & 0xFFu, b2 = (w >> 16) & 0xFFu;
    p0 = (b0 << 4) | (b2 & 0xFu);
    p1 = (b1 << 4) | (b2 >> 4);
  }
}

// 8 pixels from 12 packed bytes held in three little-endian dwords
MI_DEV void unpack12x8(uint32_t d0, uint32_t d1, uint32_t d2, bool ids, uint32_t (&v)[8]) {
  if (!ids) {
    // the standard layout is plain little-endian bit packing: pixel k = bits [12k, 12k+12)
    v[0] = d0 & 0xFFFu;
    v[1] = (d0 >> 12) & 0xFFFu;
    v[2] = __builtin_amdgcn_alignbit(d1, d0, 24) & 0xFFFu;
    v[3] = (d1 >> 4) & 0xFFFu;
    v[4] = (d1 >> 16) & 0xFFFu;
    v[5] = __builtin_amdgcn_alignbit(d2, d1, 28) & 0xFFFu;
    v[6] = (d2 >> 8) & 0xFFFu;
    v[7] = d2 >> 20;
  } else {
    const uint32_t w0 = d0 & 0xFFFFFFu;
    const uint32_t w1 = (d0 >> 24) | ((d1 & 0xFFFFu) << 8);
    const uint32_t w2 = (d1 >> 16) | ((d2 & 0xFFu) << 16);
    const uint32_t w3 = d2 >> 8;
    unpack_pair(w0, true, v[0], v[1]);
    unpack_pair(w1, true, v[2], v[3]);
    unpack_pair(w2, true, v[4], v[5]);
    unpack_pair(w3, true, v[6], v[7]);
  }
}

// scaled write of packed.py:98-100, cast(f32(v) * f32(scale/4095), E), widened back to fp32
template <class E> MI_DEV float decode_scaled(uint32_t v, float k) { return (float)cast_out<E>((float)v * k); }

// 8 values at once; for the f16 work type the products are rounded in pairs (v_cvt_pk_f16_f32: the same
// RNE conversion of the same fp32 products, half the conversion instructions) and widened back
template <class E> MI_DEV void decode_scaled8(const uint32_t (&v)[8], float k, float (&out)[8]) {
  if constexpr (sizeof(E) == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = (float)v[2 * i] * k, b = (float)v[2 * i + 1] * k;
      uint32_t pk;
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
      half_t h[2];
      __builtin_memcpy(h, &pk, 4);
      out[2 * i] = (float)h[0]; out[2 * i + 1] = (float)h[1];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = decode_scaled<E>(v[i], k);
  }
}

// one 8-px unit (LDS columns 8*lu .. 8*lu+7) = even slot 2*lu and odd slot 2*lu+1 of the row
MI_DEV void lds_store8(float* row, int lu, const float (&v)[8]) {
  *reinterpret_cast<float4*>(row + 4 * lu) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(row + HALF + 4 * lu) = make_float4(v[4], v[5], v[6], v[7]);
}

// ---------------------------------------------------------------------------------------------
// phase 1: fill the LDS tile.  LDS (lr, lc) <-> image (rb + lr, cb + lc), rb = r0-2, cb = c0-8.
// Out-of-image elements are zero (they contribute 0*w, an exact no-op, to the accumulators).
// ---------------------------------------------------------------------------------------------
// General packed path: any width / alignment, byte loads.
template <class E>
MI_DEV void fill_packed(const Params& p, float* lds, int rb, int cb) {
  const uint8_t* base = static_cast<const uint8_t*>(p.src);
  const bool is16 = p.src_kind == SRC_PACKED16;
  const bool ids = p.src_kind == SRC_PACKED12_IDS;
  const size_t pitch = is16 ? (size_t)p.W * 2 : (size_t)p.W * 3 / 2;
  for (int u = threadIdx.x; u < LDS_ROWS * UNITS; u += THREADS) {
    const int lr = u / UNITS, lu = u - lr * UNITS;
    const int r = rb + lr, c = cb + lu * 8;
    float* dst = lds + lr * PITCH;
    const bool inside = r >= 0 && r < p.H && c >= 0 && c < p.W;
    const int n = inside ? (c + 8 <= p.W ? 8 : p.W - c) : 0;     // W is even
    const uint8_t* rowp = base + (size_t)(inside ? r : 0) * pitch;
    if (!is16) {
      const uint8_t* q = rowp + (size_t)(inside ? c : 0) * 3 / 2;  // c % 8 == 0 -> 12-byte multiple
      for (int j = 0; j < 4; ++j) {
        uint32_t a = 0, b = 0;
        if (2 * j < n) {
          const uint32_t w = q[3 * j] | (q[3 * j + 1] << 8) | (q[3 * j + 2] << 16);
          unpack_pair(w, ids, a, b);
        }
        dst[lds_pos(lu * 8 + 2 * j)] = decode_scaled<E>(a, p.k_decode);
        dst[lds_pos(lu * 8 + 2 * j + 1)] = decode_scaled<E>(b, p.k_decode);
      }
    } else {
      const uint8_t* q = rowp + (size_t)(inside ? c : 0) * 2;
      for (int j = 0; j < 8; ++j) {
        uint32_t a = 0;
        if (j < n) a = q[2 * j] | (q[2 * j + 1] << 8);
        dst[lds_pos(lu * 8 + j)] = decode_scaled<E>(a, p.k_decode);
      }
    }
  }
}

// Fast path (W % 8 == 0, aligned base): every unit is either wholly inside the image or wholly
// outside.  All global loads of a lane (up to 3 units = 36/48 bytes) are issued before the first
// use, so one memory latency is paid per tile instead of one per unit; an all-zero unit decodes to
// zeros, so out-of-image units need no branch after the load.
template <class E>
MI_DEV void fill_packed_fast(const Params& p, float* lds, int rb, int cb) {
  constexpr int NUNITS = LDS_ROWS * UNITS;
  constexpr int NIT = (NUNITS + THREADS - 1) / THREADS;
  static_assert(UNITS == 18 && NUNITS < 65536 / 4, "magic division below assumes 18 units per row");
  const uint8_t* base = static_cast<const uint8_t*>(p.src);
  const bool is16 = p.src_kind == SRC_PACKED16;
  const bool ids = p.src_kind == SRC_PACKED12_IDS;
  const size_t pitch = is16 ? (size_t)p.W * 2 : (size_t)p.W * 3 / 2;
  uint4 raw[NIT];
  int off[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int u = threadIdx.x + it * THREADS;
    const int lr = (u * 3641) >> 16;                 // u / 18 for u < 16384
    const int lu = u - lr * UNITS;
    const int r = rb + lr, c = cb + lu * 8;
    off[it] = u < NUNITS ? lr * PITCH + lu * 4 : -1;
    raw[it] = make_uint4(0, 0, 0, 0);
    if (u < NUNITS && r >= 0 && r < p.H && c >= 0 && c < p.W) {
      const uint8_t* rowp = base + (size_t)r * pitch;
      if (!is16) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(rowp + (size_t)c * 3 / 2);
        raw[it].x = q[0]; raw[it].y = q[1]; raw[it].z = q[2];
      } else {
        raw[it] = *reinterpret_cast<const uint4*>(rowp + (size_t)c * 2);
      }
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (off[it] < 0) continue;
    uint32_t v[8];
    const uint4 d = raw[it];
    if (!is16) {
      unpack12x8(d.x, d.y, d.z, ids, v);
    } else {
      v[0] = d.x & 0xFFFFu; v[1] = d.x >> 16; v[2] = d.y & 0xFFFFu; v[3] = d.y >> 16;
      v[4] = d.z & 0xFFFFu; v[5] = d.z >> 16; v[6] = d.w & 0xFFFFu; v[7] = d.w >> 16;
    }
    float out[8];
    decode_scaled8<E>(v, p.k_decode, out);
    lds_store8(lds + off[it], 0, out);
  }
}

// Plain CFA with aligned rows and whole tiles (HOT == 2, 3): the same 8-pixel units as the packed
// fast path, 8 / 16 / 32 bytes per lane and unit, every load issued before the first use.
template <class S>
MI_DEV void fill_cfa_fast(const Params& p, float* lds, int rb, int cb) {
  constexpr int NUNITS = LDS_ROWS * UNITS;
  constexpr int NIT = (NUNITS + THREADS - 1) / THREADS;
  typedef typename std::conditional<sizeof(S) == 1, uint2, uint4>::type Q;   // 8 pixels = 8 / 16 / 32 bytes
  constexpr int NQ = (int)(sizeof(S) * 8 / sizeof(Q));
  const S* base = static_cast<const S*>(p.src);
  Q raw[NIT][NQ];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int u = threadIdx.x + it * THREADS;
    const int lr = (u * 3641) >> 16;                 // u / 18 for u < 16384
    const int lu = u - lr * UNITS;
    const int r = rb + lr, c = cb + lu * 8;
#pragma unroll
    for (int q = 0; q < NQ; ++q) raw[it][q] = Q{};                        // zero bits are 0 in every CFA dtype
    if (u < NUNITS && r >= 0 && r < p.H && c >= 0 && c < p.W) {
      const Q* src4 = reinterpret_cast<const Q*>(base + (size_t)r * p.W + c);
#pragma unroll
      for (int q = 0; q < NQ; ++q) raw[it][q] = src4[q];
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int u = threadIdx.x + it * THREADS;
    if (u >= NUNITS) continue;
    const int lr = (u * 3641) >> 16;
    const int lu = u - lr * UNITS;
    S t[8];
    __builtin_memcpy(t, raw[it], sizeof(t));
    float out[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)t[i];
    lds_store8(lds + lr * PITCH + lu * 4, 0, out);
  }
}

// Plain CFA images (u8 / u16 / f16 / f32): coalesced scalar loads, exact widening to fp32.
template <class S>
MI_DEV void fill_plain(const Params& p, float* lds, int rb, int cb) {
  const S* src = static_cast<const S*>(p.src);
  constexpr int NEED = TILE_W + 4;                     // image cols [c0-2, c0+TILE_W+2)
  static_assert(NEED == 132 && LDS_ROWS == 36, "magic division below assumes 36 x 132");
  for (int i = threadIdx.x; i < LDS_ROWS * NEED; i += THREADS) {
    const int lr = (i * 993) >> 17;                    // i / 132 for i < 36 * 132
    const int k = i - lr * NEED;
    const int lc = k + 6;                              // c0-2 == cb+6
    const int r = rb + lr, c = cb + lc;
    float v = 0.f;
    if (r >= 0 && r < p.H && c >= 0 && c < p.W) v = (float)src[(size_t)r * p.W + c];
    lds[lr * PITCH + lds_pos(lc)] = v;
  }
}

template <class E>
MI_DEV void fill_tile(const Params& p, float* lds, int rb, int cb) {
  switch (p.src_kind) {
    case SRC_CFA_U8: fill_plain<uint8_t>(p, lds, rb, cb); break;
    case SRC_CFA_U16: fill_plain<uint16_t>(p, lds, rb, cb); break;
    case SRC_CFA_F16: fill_plain<half_t>(p, lds, rb, cb); break;
    case SRC_CFA_F32: fill_plain<float>(p, lds, rb, cb); break;
    default:
      if (p.src_fast) fill_packed_fast<E>(p, lds, rb, cb);
      else fill_packed<E>(p, lds, rb, cb);
      break;
  }
}

// ---------------------------------------------------------------------------------------------
// phase 2: the lane's 6 x 12 window, LDS -> registers (window origin = image (r-2, c-2))
// ---------------------------------------------------------------------------------------------
MI_DEV void load_window(const float* lds, int tx, int ty, float (&win)[6][12]) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    // window col j <-> LDS col 8*tx + 6 + j: slots 2tx+1 (upper half), 2tx+2, 2tx+3, 2tx+4 (lower half)
    // volatile: keeps hipcc from narrowing / re-pairing these into ds_read2_b64 / ds_read2_b32,
    // whose 32-bank addressing conflicts on this layout; each stays one conflict-free ds_read_b128
    // (the pointer is cast back to the LDS address space: a volatile access through a generic
    // pointer would become flat_load)
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef const volatile __attribute__((address_space(3))) f4* lds_f4_ptr;
    const float* rp = lds + (2 * ty + k) * PITCH + 4 * tx;
    const f4 a = *(lds_f4_ptr)(rp + HALF);
    const f4 b = *(lds_f4_ptr)(rp + 4);
    const f4 c = *(lds_f4_ptr)(rp + HALF + 4);
    const f4 d = *(lds_f4_ptr)(rp + 8);
    win[k][0] = a.z; win[k][1] = a.w;
    win[k][2] = b.x; win[k][3] = b.y; win[k][4] = b.z; win[k][5] = b.w;
    win[k][6] = c.x; win[k][7] = c.y; win[k][8] = c.z; win[k][9] = c.w;
    win[k][10] = d.x; win[k][11] = d.y;
  }
}

// filter_at (bayer.py:138-155) for the pixel at strip position (I, K): sequential fp32 accumulation
// over the non-zero taps in reference order with the weights w/16.
template <int KIDX, bool EXACT, int I, int K>
MI_DEV void accumulate(const float (&wq)[8], const float (&win)[6][12], float (&acc)[3]) {
  bool first[3] = {true, true, true};
  static_for<0, 13>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const float x = win[I + 2 + TAP_DR[t]][K + 2 + TAP_DC[t]];
    static_for<0, 3>([&](auto cc) {
      constexpr int ch = decltype(cc)::value;
      constexpr int wi = KW[KIDX][t][ch];
      if constexpr (wi != 0) {
        constexpr float w = (float)wi * 0.0625f;      // literal operand: see the note at Params::wq
        (void)wq;
        if (first[ch]) { acc[ch] = x * w; first[ch] = false; }      // == fma(x, w, +0)
        else if constexpr (EXACT) acc[ch] = __builtin_fmaf(x, w, acc[ch]);
        else acc[ch] = acc[ch] + x * w;
      }
    });
  });
}

// in-bounds weight sums for a border pixel (the `t` of bayer.py:143-149): 0/1 masks for the five
// row and five column offsets, then t += mask * w over the non-zero compile-time weights (exact
// small integers in fp32).  No memory access; only strips that touch the image frame run it.
template <int KIDX>
MI_DEV void border_weight(int r, int c, int H, int W, float (&t3)[3]) {
  float rm[5], cm[5];
#pragma unroll
  for (int d = -2; d <= 2; ++d) {
    rm[d + 2] = (r + d >= 0 && r + d < H) ? 1.f : 0.f;
    cm[d + 2] = (c + d >= 0 && c + d < W) ? 1.f : 0.f;
  }
  t3[0] = t3[1] = t3[2] = 0.f;
  static_for<0, 13>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const float m = rm[TAP_DR[t] + 2] * cm[TAP_DC[t] + 2];
    static_for<0, 3>([&](auto cc) {
      constexpr int ch = decltype(cc)::value;
      constexpr int w = KW[KIDX][t][ch];
      if constexpr (w != 0) t3[ch] = __builtin_fmaf(m, (float)w, t3[ch]);
    });
  });
}

// ---------------------------------------------------------------------------------------------
// stores: one strip row = 8 px * 3 ch = 24 elements, contiguous in the (H, W, 3) output
// ---------------------------------------------------------------------------------------------
template <class T>
MI_DEV void store_row(T* dst, const float (&v)[24], int npx, bool vec) {
  T o[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) o[i] = cast_out<T>(v[i]);
  if (vec && npx == 8) {
    if (sizeof(T) == 1) {
      const uint2* s = reinterpret_cast<const uint2*>(o);
      uint2* d = reinterpret_cast<uint2*>(dst);
      d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    } else {
      constexpr int N = (int)(sizeof(T) * 24 / 16);
      const uint4* s = reinterpret_cast<const uint4*>(o);
      uint4* d = reinterpret_cast<uint4*>(dst);
#pragma unroll
      for (int i = 0; i < N; ++i) d[i] = s[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 24; ++i)
      if (i < npx * 3) dst[i] = o[i];      // static indices: keeps o[] in registers
  }
}

// Wave-cooperative store of one strip row of a FULL tile: the wave's 64 lanes (4 strip rows x 16
// strips) hold 4 segments of 128 px, each contiguous in the output.  The 48-byte-per-lane pieces are
// transposed through `stage` (the wave's own slice of the CFA tile buffer, free once every wave has
// its window) so that each store instruction writes long contiguous runs: 6.2 vs 3.7 TB/s measured
// for the per-lane 3 x 16 B pattern (scratch/store_bench.hip).  T is 1 or 2 bytes wide.
// 4-byte outputs have 6 units per lane: their LDS slots are padded to 7 per lane so that the 16-byte
// writes of consecutive lanes fall into different bank groups (WAVE_STAGE_UNITS per wave).
template <class T> struct StagePitch { static constexpr int value = IoUnits<T>::value == 6 ? 7 : IoUnits<T>::value; };
template <class T>
MI_DEV void wave_store_row(T* dst, int W, int row0, int c0, int lane, void* stage, const float (&v)[24]) {
  typedef typename IoUnit<T>::type U;
  constexpr int N = IoUnits<T>::value, P = StagePitch<T>::value;
  static_assert(N == 3 || N == 6, "3 or 6 units per lane");
  T o[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) o[i] = cast_out<T>(v[i]);
  U mine[N];
  __builtin_memcpy(mine, o, sizeof(mine));
  U* lb = static_cast<U*>(stage);
#pragma unroll
  for (int j = 0; j < N; ++j) lb[lane * P + j] = mine[j];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int q = j * 64 + lane;                       // unit index in the wave's 4 segments x 16 N units
    const int sgm = (q >= 16 * N) + (q >= 32 * N) + (q >= 48 * N);
    U* g = reinterpret_cast<U*>(dst + ((size_t)(row0 + 2 * sgm) * W + c0) * 3) + (q - 16 * N * sgm);
    const int src_lane = (N == 3) ? 0 : q / N;         // N == 3: slot index == q
    *g = (N == 3) ? lb[q] : lb[src_lane * P + (q - src_lane * N)];
  }
  __builtin_amdgcn_wave_barrier();
}

// clamp(x, 0, 1) (bayer.py:155) and the RNE conversion to f16 of two values in ONE instruction: the clamp
// output modifier of v_cvt_pk_f16_f32 (rounding is monotone and 0, 1 are representable, so clamping the
// rounded value equals rounding the clamped one; NaN -> 0 like clamp01).  Replaces 2 v_max..clamp + 1 v_cvt_pk.
MI_DEV uint32_t cvt_pk_f16_clamp01(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// wave_store_row for f16 outputs from UNCLAMPED fp32 values (the HOT kernels with unit scale)
MI_DEV void wave_store_row_f16_clamped(half_t* dst, int W, int row0, int c0, int lane, void* stage, const float (&v)[24]) {
  uint4 mine[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
    mine[j] = make_uint4(cvt_pk_f16_clamp01(v[8 * j], v[8 * j + 1]), cvt_pk_f16_clamp01(v[8 * j + 2], v[8 * j + 3]),
                         cvt_pk_f16_clamp01(v[8 * j + 4], v[8 * j + 5]), cvt_pk_f16_clamp01(v[8 * j + 6], v[8 * j + 7]));
  uint4* lb = static_cast<uint4*>(stage);
#pragma unroll
  for (int j = 0; j < 3; ++j) lb[lane * 3 + j] = mine[j];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int q = j * 64 + lane;                       // unit index in the wave's 4 x 48 units
    const int sgm = (q >= 48) + (q >= 96) + (q >= 144);
    uint4* g = reinterpret_cast<uint4*>(dst + ((size_t)(row0 + 2 * sgm) * W + c0) * 3) + (q - 48 * sgm);
    *g = lb[q];
  }
  __builtin_amdgcn_wave_barrier();
}

MI_DEV void store_row_dyn(const Params& p, int r, int c, const float (&v)[24], int npx) {
  if (MI_DEBUG_SKIP(p) & 32) {                      // measurement aid: no global stores
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) s += v[i];
    if (s == 12345.678f) static_cast<float*>(p.dst)[0] = s;
    return;
  }
  const size_t off = ((size_t)r * p.W + c) * 3;
  switch (p.out_dtype) {
    case MI_U8: store_row<uint8_t>(static_cast<uint8_t*>(p.dst) + off, v, npx, p.vec_store); break;
    case MI_U16: store_row<uint16_t>(static_cast<uint16_t*>(p.dst) + off, v, npx, p.vec_store); break;
    case MI_F16: store_row<half_t>(static_cast<half_t*>(p.dst) + off, v, npx, p.vec_store); break;
    default: store_row<float>(static_cast<float*>(p.dst) + off, v, npx, p.vec_store); break;
  }
}

// ---------------------------------------------------------------------------------------------
// the kernel.  PR/PC: parity offsets of the CFA pattern relative to RGGB, so that the kernel used
// at (row, col) is K[((row+PR)&1) + 2*((col+PC)&1)] (bayer.py:92-97,165-175):
// RGGB (0,0), GBRG (1,0), GRBG (0,1), BGGR (1,1).
// ---------------------------------------------------------------------------------------------
// HOT: the configuration of the packed-RAW pipelines, fixed at compile time so that the kernel is
// straight-line code without the generic paths: standard 12-bit packing with aligned rows, whole
// 8-pixel strips (W % 8 == 0; ragged right / bottom tiles store per lane), stores in the work dtype (see hot_spec); HOT == 2: the same for a plain f16 / f32 CFA image.
// measurement aid (make EXTRA=-DMI_TILE_STAMPS): wave 0 of every block leaves s_memtime stamps of its
// phases in workspace rows 2.. (32-bit, 8 per block); see scripts/tile_stamps.py
#ifdef MI_TILE_STAMPS
#define MI_STAMP(i)                                                                                      \
  do {                                                                                                   \
    if (HOT != 0 && threadIdx.x == 0)                                                                      \
      reinterpret_cast<unsigned*>(p.partials + 2 * (size_t)p.part_stride)[blockIdx.x * 8 + (i)] =       \
          (unsigned)__builtin_readcyclecounter();                                                        \
  } while (0)
#else
#define MI_STAMP(i) do {} while (0)
#endif
template <class E> constexpr int dtype_code() { return sizeof(E) == 2 ? (int)MI_F16 : (int)MI_F32; }
template <class E, int PR, int PC, int EPI, int HOT = 0>
__global__ __launch_bounds__(THREADS) void tile_kernel(const Params p_in) {
  constexpr bool EXACT = sizeof(E) == 2;
  Params p = p_in;
  // HOT == 3: integer CFA, u8 for the f16 work type and u16 for f32 (mi_isp_demosaic), same dtype out
  typedef typename std::conditional<HOT == 3, typename std::conditional<sizeof(E) == 2, uint8_t, uint16_t>::type, E>::type CfaT;
  if constexpr (HOT != 0) {
    constexpr int cfa_code = HOT == 3 ? (sizeof(E) == 2 ? (int)MI_U8 : (int)MI_U16) : dtype_code<E>();
    // SRC_CFA_* share the MI_* numbering; HOT == 1 keeps the two 12-bit layouts (standard / IDS) a run-time flag
    p.src_kind = HOT == 1 ? (p.src_kind == SRC_PACKED12_IDS ? (int)SRC_PACKED12_IDS : (int)SRC_PACKED12) : cfa_code;
    p.src_fast = 1; p.debug_skip = 0; p.vec_store = 1;      // the colour matrix stays a run-time (uniform) branch
    p.in_scale = HOT == 3 ? ScaleOf<CfaT>::value : 1.f;
    p.out_dtype = cfa_code; p.out_scale = p.in_scale;
  }
  // the tile buffer doubles as the per-wave staging of the cooperative stores; 4-byte outputs of the
  // specialised kernels need 64 lanes x 7 units x 16 B per wave
  constexpr bool STAGE_F32 = (HOT == 1 || HOT == 2) && sizeof(E) == 4;
  constexpr int WAVE_STAGE_FLOATS = STAGE_F32 ? 64 * 7 * 4 : 64 * 12;
  constexpr int LDS_FLOATS = LDS_ROWS * PITCH > 4 * WAVE_STAGE_FLOATS ? LDS_ROWS * PITCH : 4 * WAVE_STAGE_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  __shared__ float red[4][8];

  // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8).  With a row-major tile order
  // and tiles_x % 8 == 0 every left-edge tile would land on one XCD and every right-edge tile on
  // another, and the edge tiles are the slow ones (border renormalisation).  Rotating each tile row
  // by its row index spreads them over all XCDs (speed only; any mapping is correct).
  // The tile rows are also rotated by half the image so that the (slow) top and bottom rows run in
  // the middle of the grid instead of being its tail.
  const int tiles_x = (p.W + TILE_W - 1) / TILE_W, tiles_y = (p.H + TILE_H - 1) / TILE_H;
  const int gy = blockIdx.x / tiles_x;
  int bx = blockIdx.x - gy * tiles_x + gy % tiles_x;
  if (bx >= tiles_x) bx -= tiles_x;
  int by = gy + tiles_y / 2;
  if (by >= tiles_y) by -= tiles_y;
  const int r0 = by * TILE_H, c0 = bx * TILE_W;

  MI_STAMP(0);
  if constexpr (HOT == 1) fill_packed_fast<E>(p, lds, r0 - 2, c0 - 8);
  else if constexpr (HOT >= 2) fill_cfa_fast<CfaT>(p, lds, r0 - 2, c0 - 8);
  else if (!(MI_DEBUG_SKIP(p) & 1)) fill_tile<E>(p, lds, r0 - 2, c0 - 8);
  MI_STAMP(1);
  __syncthreads();
  MI_STAMP(2);

  const int tx = threadIdx.x & (STRIPS_X - 1), ty = threadIdx.x / STRIPS_X;
  const int r = r0 + STRIP_H * ty, c = c0 + STRIP_W * tx;
  // H, W even -> both rows, pixel pairs in.  HOT: W % 8 == 0, so a strip is all in or all out
  const bool active = r < p.H && c < p.W && (HOT != 0 || !(MI_DEBUG_SKIP(p) & 2));
  const int npx = HOT != 0 ? 8 : (active ? (p.W - c < 8 ? p.W - c : 8) : 0);

  // tonemap scalars (uniform loads); unused ones are dead code per EPI
  float lo = 0.f, inv = 1.f, lo2 = 0.f, inv2 = 1.f;
  ReinhardK rk;
  if (EPI == EPI_STATS || EPI == EPI_RH_MINMAX || EPI == EPI_RH_STORE) { lo = p.fp[FP_LO]; inv = p.fp[FP_INV]; }
  if (EPI == EPI_RH_MINMAX || EPI == EPI_RH_STORE) {
    rk.map_key = p.fp[FP_MAPKEY]; rk.ei = p.fp[FP_EI];
    rk.mean3[0] = p.fp[FP_MEAN3]; rk.mean3[1] = p.fp[FP_MEAN3 + 1]; rk.mean3[2] = p.fp[FP_MEAN3 + 2];
    rk.la = p.la; rk.ca = p.ca;
  }
  if (EPI == EPI_RH_STORE) { lo2 = p.fp[FP_LO2]; inv2 = p.fp[FP_INV2]; }

  float vmin = __builtin_inff(), vmax = -__builtin_inff();
  StatsAcc st; st.init();

  constexpr bool STORES = EPI == EPI_STORE || EPI == EPI_RH_STORE || EPI == EPI_STORE_MINMAX;
  float win[6][12];
  if (active) load_window(lds, tx, ty, win);
  // a full tile with 1/2-byte outputs is stored wave-cooperatively through the (now free) tile buffer
  const bool coop_store = STORES && (STAGE_F32 || p.out_dtype != MI_F32) && r0 + TILE_H <= p.H && c0 + TILE_W <= p.W &&
                          (HOT != 0 || (p.vec_store && !(MI_DEBUG_SKIP(p) & 32)));
  if (STORES) __syncthreads();                          // every wave holds its window
  MI_STAMP(3);
  void* stage = lds + (threadIdx.x >> 6) * WAVE_STAGE_FLOATS;   // 3 KB (7 KB for 4-byte outputs) per wave
  if (active) {
    // every tap of all 16 pixels in bounds, and c / (in_scale * t) == c / 16 ?
    const bool fast = p.in_scale == 1.f && r >= 2 && r + 1 < p.H - 2 && c >= 2 && c + 7 < p.W - 2;
    // Wave-uniform predicates (scalar branches): the slow blocks below are real branches that
    // interior waves skip, not per-lane selects the compiler would flatten into the fast stream.
    const bool wave_has_slow = __builtin_amdgcn_ballot_w64(!fast) != 0 && !(MI_DEBUG_SKIP(p) & 16);
    const bool wave_all_full = HOT != 0 || __builtin_amdgcn_ballot_w64(npx != 8) == 0;

    static_for<0, 2>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      float v[24];                                    // the row's 8 px x RGB, normalised
      if (MI_DEBUG_SKIP(p) & 8) {                       // measurement aid: no accumulation
#pragma unroll
        for (int j = 0; j < 24; ++j) v[j] = win[i + 2][(j % 12)] + win[i + (j & 3)][j >> 1];
      } else {
      static_for<0, 8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int KIDX = ((i + PR) & 1) + 2 * ((k + PC) & 1);
        float acc[3];
        accumulate<KIDX, EXACT, i, k>(p.wq, win, acc);
        v[3 * k] = acc[0]; v[3 * k + 1] = acc[1]; v[3 * k + 2] = acc[2];
      });
      }
      if (wave_has_slow) {
        // border pixels / CFAs with scale != 1: c / (in_scale * t), correctly rounded, with
        // c == acc * 16 exactly and t the in-bounds weight sum (16 when every tap is in bounds),
        // looked up in BORDER_T.  Each pixel position is skipped (scalar branch) when no lane of the
        // wave needs it, e.g. only columns 0-1 of a left-edge strip are border pixels.  All lookups
        // are issued before the first use.
        const int rmask = inside_mask(r + i, p.H);
        // interior pixels of an integer CFA (in_scale 255 / 65535, t = 16): c / (in_scale * 16) as
        // c * RN(1 / d) with one FMA residual correction - equal to the IEEE division for every integer c
        // these sums can take (tests/test_oracle.py::test_scaled_division_trick), a quarter of its cost
        const float d16 = p.in_scale * 16.f, r16 = 1.0f / d16;
        uint32_t tq[8];
        static_for<0, 8>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          constexpr int KIDX = ((i + PR) & 1) + 2 * ((k + PC) & 1);
          const int cmask = inside_mask(c + k, p.W);
          const bool border = rmask != 31 || cmask != 31;
          tq[k] = 0x101010u;                                            // t = 16 for all channels
          if (__builtin_amdgcn_ballot_w64(border) != 0) tq[k] = BORDER_T.t[KIDX][rmask][cmask];
        });
        static_for<0, 8>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          const int cmask = inside_mask(c + k, p.W);
          const bool need = p.in_scale != 1.f || rmask != 31 || cmask != 31;
          if (__builtin_amdgcn_ballot_w64(need) != 0) {
            if (__builtin_amdgcn_ballot_w64(tq[k] != 0x101010u) == 0) {
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) {
                const float cnum = v[3 * k + ch] * 16.f;
                const float q = cnum * r16;
                const float fixed = __builtin_fmaf(__builtin_fmaf(-q, d16, cnum), r16, q);
                v[3 * k + ch] = need ? fixed : v[3 * k + ch];
              }
            } else {
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) {
                const float t = (float)(int)(int8_t)(tq[k] >> (8 * ch));
                const float fixed = (v[3 * k + ch] * 16.f) / (p.in_scale * t);
                v[3 * k + ch] = need ? fixed : v[3 * k + ch];
              }
            }
          }
        });
      }
      if (p.has_ccm) {                                // bayer.py:152-153, sequential fp32 dot
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float a = v[3 * k], b = v[3 * k + 1], d = v[3 * k + 2];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch)
            v[3 * k + ch] = (p.ccm[3 * ch] * a + p.ccm[3 * ch + 1] * b) + p.ccm[3 * ch + 2] * d;
        }
      }

      auto store_row_any = [&](const float (&vals)[24]) {
        if (coop_store) {                               // block-uniform
          const int lane = threadIdx.x & 63, row0 = r0 + 8 * (threadIdx.x >> 6) + i;
          switch (p.out_dtype) {
            case MI_U8: wave_store_row<uint8_t>(static_cast<uint8_t*>(p.dst), p.W, row0, c0, lane, stage, vals); break;
            case MI_U16: wave_store_row<uint16_t>(static_cast<uint16_t*>(p.dst), p.W, row0, c0, lane, stage, vals); break;
            case MI_F32:
              if constexpr (STAGE_F32) wave_store_row<float>(static_cast<float*>(p.dst), p.W, row0, c0, lane, stage, vals);
              break;
            default: wave_store_row<half_t>(static_cast<half_t*>(p.dst), p.W, row0, c0, lane, stage, vals); break;
          }
        } else {
          store_row_dyn(p, r + i, c, vals, npx);
        }
      };
      // the row epilogue, specialised on "all 8 pixels of every lane are live" (scalar branch)
      auto epilogue = [&](auto full_c, auto ca0_c) {
        constexpr bool FULL = decltype(full_c)::value;
        constexpr bool CA0 = decltype(ca0_c)::value;
        if constexpr (EPI == EPI_MINMAX) {
          // bounds of the work-dtype image: clamp (bayer.py:155) and rounding to E are monotone,
          // so they are applied once to the reduced min / max (finalize), not to every pixel
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (FULL || k < npx) {
              vmin = fminf(vmin, fminf(v[3 * k], fminf(v[3 * k + 1], v[3 * k + 2])));
              vmax = fmaxf(vmax, fmaxf(v[3 * k], fmaxf(v[3 * k + 1], v[3 * k + 2])));
            }
          }
        } else if constexpr ((EPI == EPI_STORE || EPI == EPI_STORE_MINMAX) && (HOT == 1 || HOT == 2) && sizeof(E) == 2) {
          // f16 image with unit scale: clamp and conversion are one instruction per pixel pair; the bounds
          // are reduced on the unclamped fp32 values (clamp and rounding are monotone: finalize applies them)
          if constexpr (EPI == EPI_STORE_MINMAX) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              vmin = fminf(vmin, fminf(v[3 * k], fminf(v[3 * k + 1], v[3 * k + 2])));
              vmax = fmaxf(vmax, fmaxf(v[3 * k], fmaxf(v[3 * k + 1], v[3 * k + 2])));
            }
          }
          if (coop_store) {                               // block-uniform: whole tile
            wave_store_row_f16_clamped(static_cast<half_t*>(p.dst), p.W, r0 + 8 * (threadIdx.x >> 6) + i, c0,
                                       threadIdx.x & 63, stage, v);
          } else {                                        // ragged right / bottom tile: per-lane vector stores
#pragma unroll
            for (int j = 0; j < 24; ++j) v[j] = clamp01(v[j]);
            store_row_dyn(p, r + i, c, v, 8);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 24; ++j) v[j] = clamp01(v[j]);                       // bayer.py:155
          if constexpr (EPI == EPI_STORE || EPI == EPI_STORE_MINMAX) {
            if constexpr (EPI == EPI_STORE_MINMAX) {
              // bounds of the stored image: rounding to the output dtype is monotone, applied once
              // to the reduced min / max (finalize)
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                if (FULL || k < npx) {
                  vmin = fminf(vmin, fminf(v[3 * k], fminf(v[3 * k + 1], v[3 * k + 2])));
                  vmax = fmaxf(vmax, fmaxf(v[3 * k], fmaxf(v[3 * k + 1], v[3 * k + 2])));
                }
              }
            }
#pragma unroll
            for (int j = 0; j < 24; ++j) v[j] *= p.out_scale;
            store_row_any(v);
          } else {
            // the reference materialises the demosaiced image in the work dtype (scale 1)
            float row[24];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              float t[3];
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) t[ch] = norm01((float)cast_out<E>(v[3 * k + ch]), lo, inv);
              const bool live = FULL || k < npx;
              if constexpr (EPI == EPI_STATS) {
                if (live) st.add(t[0], t[1], t[2]);
              } else {
                float q[3];
                reinhard_px<CA0>(t, rk, q);
                if constexpr (EPI == EPI_RH_MINMAX) {
                  if (live) {
                    vmin = fminf(vmin, fminf(q[0], fminf(q[1], q[2])));
                    vmax = fmaxf(vmax, fmaxf(q[0], fmaxf(q[1], q[2])));
                  }
                } else {
#pragma unroll
                  for (int ch = 0; ch < 3; ++ch) row[3 * k + ch] = q[ch];
                }
              }
            }
            if constexpr (EPI == EPI_RH_STORE) {
              linear_n<24>(row, lo2, inv2, p.gamma_inv, p.out_scale);                 // tonemap.py:154
              store_row_any(row);
            }
          }
        }
      };
      MI_STAMP(4 + i);                                  // 4: row 0 computed, 5: row 1 computed (before their stores)
      if constexpr (EPI == EPI_RH_MINMAX || EPI == EPI_RH_STORE) {
        if (wave_all_full && rk.ca == 0.f) epilogue(std::true_type{}, std::true_type{});      // the hot path
        else if (rk.ca == 0.f) epilogue(std::false_type{}, std::true_type{});
        else epilogue(std::false_type{}, std::false_type{});
      } else {
        if (wave_all_full) epilogue(std::true_type{}, std::true_type{});
        else epilogue(std::false_type{}, std::true_type{});
      }
    });
  }

  MI_STAMP(6);
  if (MI_DEBUG_SKIP(p) & 4) {                       // measurement aid: no block reduction
    if (vmin + vmax + st.slog == 12345.f) p.partials[blockIdx.x] = vmin;
    return;
  }
  if (EPI == EPI_MINMAX || EPI == EPI_RH_MINMAX || EPI == EPI_STORE_MINMAX) {
    const float v[2] = {vmin, vmax};
    const int op[2] = {0, 1};
    block_reduce_store<2>(v, op, red, p.partials, p.part_stride, blockIdx.x);
    MI_STAMP(7);
  } else if (EPI == EPI_STATS) {
    const float v[7] = {st.gmin, st.gmax, st.slog, st.sgray, st.s0, st.s1, st.s2};
    const int op[7] = {0, 1, 2, 2, 2, 2, 2};
    block_reduce_store<7>(v, op, red, p.partials, p.part_stride, blockIdx.x);
  }
}

// host-side launchers, one per pattern translation unit (isp_tile_p{0..3}.hip)
int launch_rggb(const Params& p, int work_dtype, int epi, hipStream_t stream);
int launch_grbg(const Params& p, int work_dtype, int epi, hipStream_t stream);
int launch_gbrg(const Params& p, int work_dtype, int epi, hipStream_t stream);
int launch_bggr(const Params& p, int work_dtype, int epi, hipStream_t stream);
static inline int launch(const Params& p, int work_dtype, int pattern, int epi, hipStream_t stream) {
  switch (pattern) {
    case MI_RGGB: return launch_rggb(p, work_dtype, epi, stream);
    case MI_GRBG: return launch_grbg(p, work_dtype, epi, stream);
    case MI_GBRG: return launch_gbrg(p, work_dtype, epi, stream);
    default: return launch_bggr(p, work_dtype, epi, stream);
  }
}

// which compile-time specialisation (template parameter HOT) may this launch of epilogue `epi` use?
// 0: none; 1: packed 12-bit source (standard layout, aligned rows); 2: plain f16 / f32 CFA of the work
// dtype with 16-byte aligned rows; 3: u8 / u16 CFA, same dtype out.  All: W % 8 == 0.
static inline int hot_spec(const Params& p, int work_dtype, int epi) {
  const bool stores = epi == EPI_STORE || epi == EPI_STORE_MINMAX;
  const bool common = stores && MI_DEBUG_SKIP(p) == 0 && p.W % 8 == 0 && p.vec_store;
  if (!common) return 0;
  const bool unit = p.in_scale == 1.f && p.out_scale == 1.f && p.out_dtype == work_dtype;
  if (unit && (p.src_kind == SRC_PACKED12 || p.src_kind == SRC_PACKED12_IDS) && p.src_fast) return 1;
  if (epi != EPI_STORE || ((uintptr_t)p.src & 15) != 0) return 0;
  if (unit && p.src_kind == work_dtype && (work_dtype == MI_F16 || work_dtype == MI_F32)) return 2;
  // 3: integer CFA (u8 with the f16 work type, u16 with f32), same dtype out
  const int int_kind = work_dtype == MI_F16 ? (int)MI_U8 : (int)MI_U16;
  if (p.src_kind == int_kind && p.out_dtype == int_kind && p.in_scale == mi_scale_factor(int_kind) &&
      p.out_scale == p.in_scale)
    return 3;
  return 0;
}

static inline int num_tiles(int H, int W) {
  return ((W + TILE_W - 1) / TILE_W) * ((H + TILE_H - 1) / TILE_H);
}

}  // namespace tile
