// ISP.load_packed12 with resize_width / scale (camera_isp.py:333-340,371-373,302-315) on the streaming
// design: unpack -> demosaic -> bilinear resize in ONE pass over the packed frame, f16 work dtype.
//
// The demosaic is isp_stream.h's (register ring, DPP halo exchange, LDS decode table).  Every demosaiced
// row is clamped, rounded to f16 - where the reference stores its full-resolution RGB image - and written
// into a three-row ring in LDS, one ring per wave, 8 bytes per pixel (r, g, b, pad), a lane's 8 pixels contiguous
// and lanes 80 bytes apart (conflict-free 16-byte writes); a destination pixel then reads its two column taps
// of a source row as two aligned 8-byte reads.  Every source pixel is demosaiced exactly once and any scale
// works.  (rtile::resize_tile_kernel, which this replaces for f16, gathered fp32 CFA values per lane and
// evaluated four 13-tap kernels per destination pixel - 65 % LDS bank-conflict cycles - and was limited to the
// scales whose source region fits its LDS tile.)  Measured at scale 0.46875 on a 4K frame: 32.7 us against
// 33.1 us - the arithmetic per destination pixel (12 conversions, 9 three-operation mixes), not the gather, is
// what costs; DESIGN.md 5.1c.
//
// Ownership: a wave owns the source band [col0, col0 + 8 * stride_units) x [r_begin, r_end) and produces the
// destination pixels whose 2 x 2 source quad STARTS there (interpolate.py:24-34: quad origin = min(trunc(i / s),
// n - 2)) - with the band's first and last destination column moved up to the next multiple of 4 (align4): a lane
// emits 4 destination pixels = 24 bytes = three 8-byte stores, and with ragged band ends EVERY wave had a lane whose
// four pixels straddled the end, which sent the whole wave through the element-store path (12 two-byte store
// instructions per row on top of the 3 real ones).  The wave therefore demosaics two more units to the right (the
// quads' second column, and up to 3 destination pixels borrowed from the next band) and one more row below.  Bands are
// 56-62 units wide so that the extra units fit the 64 lanes.
//
// Arithmetic: bit-exact against oracle isp_load_packed12 (same demosaic; p = I / scale by true division,
// mix(x, y, a) = x (1 - a) + y a uncontracted, rows first, then columns).
#pragma once
#include "isp_stream.h"

#pragma clang fp contract(off)

namespace rstrm {

using namespace strm;

constexpr int RING = 3;                       // source rows kept per wave
constexpr int LANE_PITCH = 80;                // bytes per lane in a ring row: 8 pixels x 8 bytes + 16 of padding
constexpr int ROW_BYTES = 64 * LANE_PITCH;

struct RSArgs {
  Params t;                 // source description (tile::Params); t.dst = the (Hd, Wd, 3) f16 output
  int Hd, Wd;
  float s0, s1;             // interpolate.py:60-66: p = (r / s0, c / s1)
  int stride_units;         // units (8 px) a band owns; its wave covers stride_units + 2
  int bands_x, rows_per_wave, n_waves, n_blocks;
  int align4;               // destination columns are dealt to the bands in whole groups of 4 (see the kernel)
  int n_batch;              // several frames in one launch (grid.y = frame): frame y reads srcs[y], writes dsts[y]; 0: t.src / t.dst
  const void* srcs[LOAD_BATCH];
  void* dsts[LOAD_BATCH];
};

// quad origin of destination index i along one axis: min(trunc(i / s), n - 2)   (n >= 2)
MI_DEV int quad_origin(int i, float s, int n) {
  const int q = (int)((float)i / s);
  return q < n - 2 ? q : n - 2;
}
// the first destination index in [0, n_dst] whose quad origin is >= x (origins are non-decreasing in i)
MI_DEV int first_with_origin(int x, float s, int n_src, int n_dst) {
  int c = (int)((float)x * s) - 2;
  c = c < 0 ? 0 : c;
  while (c < n_dst && quad_origin(c, s, n_src) < x) ++c;
  return c < n_dst ? c : n_dst;
}

template <int PR, int PC>
__global__ __launch_bounds__(THREADS, 2) void resize_kernel(const RSArgs a) {
  typedef half_t E;
  const Params& p = a.t;
  __shared__ __attribute__((aligned(16))) uint4 ring_all[WAVES][RING][ROW_BYTES / 16];
  __shared__ float lut[4096];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = blockIdx.x * WAVES + wave;
  const bool wave_ok = g < a.n_waves;
  const int by = g / a.bands_x, bx = g - by * a.bands_x;
  const int col0 = bx * a.stride_units * 8;            // first source column of the band
  const int c0 = col0 + lane * 8;
  const int r_begin = by * a.rows_per_wave;
  const int r_own_end = wave_ok ? (r_begin + a.rows_per_wave < p.H ? r_begin + a.rows_per_wave : p.H) : r_begin;
  const int r_last = r_own_end < p.H ? r_own_end : p.H - 1;         // last row to demosaic (one beyond the owned ones)
  const bool col_ok = wave_ok && c0 < p.W && lane <= a.stride_units + 1;
  uint4 (*ring)[ROW_BYTES / 16] = ring_all[wave];

  // ---- packed source rows (as stream_kernel) ----
  const uint32_t pitch = (uint32_t)p.W * 3 / 2;
  const void* const src_p = a.n_batch > 0 ? a.srcs[blockIdx.y] : p.src;      // (blockIdx.y itself: see strm::SArgs)
  void* const dst_p = a.n_batch > 0 ? a.dsts[blockIdx.y] : p.dst;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src_p), 0, (int)((uint32_t)p.H * pitch), 0x00020000);
  const uint32_t col_off = col_ok ? (uint32_t)c0 * 3 / 2 : INVALID_OFF;
  const bool last_lane = lane == a.stride_units + 1 || lane == 63;
  const bool ext_ok = col_ok && ((lane == 0 && c0 > 0) || (last_lane && c0 + 8 < p.W));
  const uint32_t ext_off = ext_ok ? (uint32_t)c0 * 3 / 2 + (lane == 0 ? -4 : 12) : INVALID_OFF;
  auto load_row = [&](int r, uint32_t (&d)[4]) {
    const uint32_t row_off = (r >= 0 && r < p.H && r <= r_last + 2) ? (uint32_t)r * pitch : INVALID_OFF;
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    const u3 q = __builtin_amdgcn_raw_buffer_load_b96(rsrc, col_off + row_off, 0, 0);
    d[0] = q.x; d[1] = q.y; d[2] = q.z;
    d[3] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, ext_off + row_off, 0, 0);
  };
  // the lane right of the band's last lane is inactive and decodes zeros - but the band's last lane needs the REAL two
  // pixels to its right: it takes them from its edge dword like lane 63 does (decode_row_edge below)
  auto decode = [&](const uint32_t (&d)[4], WinRow& row) {
    uint32_t v[8];
    tile::unpack12x8(d[0], d[1], d[2], false, v);
    const uint32_t w = lane == 0 ? d[3] >> 8 : d[3] & 0xFFFFFFu;
    float own[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) own[i] = lut[v[i]];
    const float x0 = lut[w & 0xFFFu], x1 = lut[w >> 12];
    const float l0 = from_left(own[6], x0), l1 = from_left(own[7], x1);
    const float r0 = from_right(own[0], x0), r1 = from_right(own[1], x1);
    row.v[0] = l0; row.v[1] = l1;
#pragma unroll
    for (int j = 0; j < 8; ++j) row.v[2 + j] = own[j];
    row.v[10] = last_lane ? x0 : r0;
    row.v[11] = last_lane ? x1 : r1;
  };

  // ---- destination geometry of this wave ----
  const int own_c_end = bx + 1 < a.bands_x ? col0 + a.stride_units * 8 : p.W;      // quad origins [col0, own_c_end)
  auto band_edge = [&](int src_col) {                  // first destination column of the band that starts at src_col
    const int c = first_with_origin(src_col, a.s1, p.W, a.Wd);
    const int c4 = a.align4 ? (c + 3) & ~3 : c;
    return c4 < a.Wd ? c4 : a.Wd;
  };
  const int cd_begin = wave_ok && bx > 0 ? band_edge(col0) : 0;
  const int cd_end = wave_ok ? (bx + 1 < a.bands_x ? band_edge(own_c_end) : a.Wd) : 0;
  const int rd_begin = wave_ok ? first_with_origin(r_begin, a.s0, p.H, a.Hd) : 0;
  const int rd_end = wave_ok ? (r_own_end < p.H ? first_with_origin(r_own_end, a.s0, p.H, a.Hd) : a.Hd) : 0;
  const __amdgpu_buffer_rsrc_t drsrc =
      __builtin_amdgcn_make_buffer_rsrc(dst_p, 0, (int)((uint32_t)a.Hd * (uint32_t)a.Wd * 6u), 0x00020000);
  const bool vec_rows = (a.Wd & 3) == 0;               // 4 pixels = 24 bytes stay 8-byte aligned in every row
  const int cg0 = cd_begin & ~3;                       // group base: lane l owns destination columns cg0 + 4 l + 256 k + j

  // column taps of the lane's pixels (interpolate.py:24-34), hoisted for the first group of 256 columns
  // ring row layout: the 8 pixels of a lane are 64 contiguous bytes, lanes are 80 bytes apart - the 16-byte writes of 8
  // consecutive lanes then fall into 8 different bank quads (at a 64-byte stride they would collide 4-fold)
  auto px_off = [&](int x) { return (uint32_t)(((x - col0) >> 3) * LANE_PITCH + ((x - col0) & 7) * 8); };
  struct ColTap { uint32_t off0, off1; float fc; bool sel_a, sel_b, valid; };
  auto col_tap = [&](int c) {
    ColTap t;
    t.valid = c >= cd_begin && c < cd_end;
    const int cc = t.valid ? c : cd_begin < a.Wd ? cd_begin : 0;
    const float pc = (float)cc / a.s1;
    const int ic = (int)pc;
    t.fc = pc - (float)ic;
    const int qc = ic < p.W - 2 ? ic : p.W - 2;
    const int ca = ic < p.W - 1 ? ic : p.W - 1, cb = ic + 1 < p.W - 1 ? ic + 1 : p.W - 1;      // index_clamped
    t.off0 = px_off(qc); t.off1 = px_off(qc + 1);
    t.sel_a = ca != qc; t.sel_b = cb != qc;
    return t;
  };
  ColTap tap0[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) tap0[j] = col_tap(cg0 + 4 * lane + j);
  // the usual case, for the whole wave: the first column tap is the quad's first column, the second its second
  // (index_clamped, interpolate.py:20-21, only bites in the image's last column) - no selects in the emission then
  bool plain0 = true;
#pragma unroll
  for (int j = 0; j < 4; ++j) plain0 = plain0 && !tap0[j].sel_a && tap0[j].sel_b;
  const bool plain_wave = __builtin_amdgcn_ballot_w64(!plain0) == 0;

  MI_SSTAMP(0);
  // ---- prologue ----
  float wq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wq[i] = vgpr(wq_value(i));
  WinRow win[6];
  uint32_t raw[3][2][4];
  {
    uint32_t pro[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) load_row(r_begin - 2 + q, pro[q]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      load_row(r_begin + 2 + 2 * j, raw[j][0]);
      load_row(r_begin + 3 + 2 * j, raw[j][1]);
    }
    for (int e = threadIdx.x; e < 4096; e += THREADS) lut[e] = tile::decode_scaled<E>((uint32_t)e, p.k_decode);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) decode(pro[q], win[q]);
  }
  MI_SSTAMP(1);
  const bool is_left = col_ok && c0 == 0, is_right = col_ok && c0 + 8 == p.W;
  const bool any_left = __builtin_amdgcn_ballot_w64(is_left) != 0, any_right = __builtin_amdgcn_ballot_w64(is_right) != 0;
  int next_r = rd_begin;                               // next destination row to emit

  // one destination row: both source rows of its quads are in the ring
  auto emit = [&](int r) __attribute__((always_inline)) {
    const float pr = (float)r / a.s0;                  // wave-uniform
    const int ir = (int)pr;
    const float fr = vgpr(pr - (float)ir), fr1 = vgpr(1.0f - (pr - (float)ir));
    const int ra = ir < p.H - 1 ? ir : p.H - 1, rb = ir + 1 < p.H - 1 ? ir + 1 : p.H - 1;      // index_clamped
    const char* row_a = reinterpret_cast<const char*>(ring[ra % RING]);
    const char* row_b = reinterpret_cast<const char*>(ring[rb % RING]);
    auto group = [&](int gbase, const ColTap (&tap)[4], auto plain_c) __attribute__((always_inline)) {
      constexpr bool PLAIN = decltype(plain_c)::value;
      float of[12];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint2 a0 = *reinterpret_cast<const uint2*>(row_a + tap[j].off0), a1 = *reinterpret_cast<const uint2*>(row_a + tap[j].off1);
        const uint2 b0 = *reinterpret_cast<const uint2*>(row_b + tap[j].off0), b1 = *reinterpret_cast<const uint2*>(row_b + tap[j].off1);
        const float fc = tap[j].fc, fc1 = 1.0f - tap[j].fc;
        // rows first (interpolate.py:28-33): m = x (1 - a) + y a, two products and a sum, each rounded (no contraction).
        // A product takes its f16 factor straight from the packed word (v_fma_mix_f32 with a zero addend: the half is
        // widened exactly, the product rounded once to fp32 - the bits of converting first and multiplying, without the
        // twelve conversions per pixel; all factors are >= 0, so the zero addend changes nothing).
        auto channel = [&](auto hi_c, uint32_t xa, uint32_t xb, uint32_t xa1, uint32_t xb1) __attribute__((always_inline)) {
          constexpr bool HI = decltype(hi_c)::value;
          float pa, pb, qa, qb;
          if constexpr (HI) {
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(pa) : "v"(xa), "v"(fr1));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(pb) : "v"(xb), "v"(fr));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(qa) : "v"(xa1), "v"(fr1));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(qb) : "v"(xb1), "v"(fr));
          } else {
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(pa) : "v"(xa), "v"(fr1));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(pb) : "v"(xb), "v"(fr));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(qa) : "v"(xa1), "v"(fr1));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(qb) : "v"(xb1), "v"(fr));
          }
          const float m0 = pa + pb, m1 = qa + qb;
          // then the column taps pick their column
          const float y1 = PLAIN ? m0 : (tap[j].sel_a ? m1 : m0);
          const float y2 = PLAIN ? m1 : (tap[j].sel_b ? m1 : m0);
          return f32_rounded(y1 * fc1 + y2 * fc);                                   // intensity scale 1 (same dtype)
        };
        of[3 * j + 0] = channel(std::false_type{}, a0.x, b0.x, a1.x, b1.x);       // r: low half of the first word
        of[3 * j + 1] = channel(std::true_type{}, a0.x, b0.x, a1.x, b1.x);        // g: its high half
        of[3 * j + 2] = channel(std::false_type{}, a0.y, b0.y, a1.y, b1.y);       // b: low half of the second word
      }
      half_t oh[12];                                        // RNE to f16, in pairs (v_cvt_pk_f16_f32)
      uint32_t opk[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(opk[k]) : "v"(of[2 * k]), "v"(of[2 * k + 1]));
      __builtin_memcpy(oh, opk, 24);
      uint32_t outp[6];
      __builtin_memcpy(outp, oh, 24);
      const int cl = gbase + 4 * lane;
      const bool all4 = vec_rows && tap[0].valid && tap[3].valid;
      const uint32_t base = ((uint32_t)r * (uint32_t)a.Wd + (uint32_t)cl) * 6u;
      typedef uint32_t u2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int k = 0; k < 3; ++k)
        __builtin_amdgcn_raw_buffer_store_b64(u2{outp[2 * k], outp[2 * k + 1]}, drsrc, all4 ? base + 8u * k : INVALID_OFF, 0, 0);
      // lanes at the band's ends (and images whose rows do not keep the 8-byte alignment): element stores, behind a
      // wave-uniform branch (most waves have none)
      if (__builtin_amdgcn_ballot_w64(!all4 && (tap[0].valid || tap[1].valid || tap[2].valid || tap[3].valid)) != 0) {
#pragma unroll
        for (int e = 0; e < 12; ++e) {
          const bool ok = !all4 && tap[e / 3].valid;
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(uint16_t, oh[e]), drsrc, ok ? base + 2u * e : INVALID_OFF, 0, 0);
        }
      }
    };
    if (plain_wave) group(cg0, tap0, std::true_type{});
    else group(cg0, tap0, std::false_type{});
    if (cg0 + 256 < cd_end) {                            // more than 256 destination columns per band: upscaling
      asm volatile("" ::: "memory");                     // keep this a real branch (the taps are recomputed inside)
      for (int gbase = cg0 + 256; gbase < cd_end; gbase += 256) {
        ColTap tap[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) tap[j] = col_tap(gbase + 4 * lane + j);
        group(gbase, tap, std::false_type{});
      }
    }
  };

  const bool younger = blockIdx.x * 2 >= gridDim.x;
  auto body = [&](auto ph_c, int i) {
    constexpr int PH = decltype(ph_c)::value;
    const int r = r_begin + 2 * i;
    decode(raw[PH][0], win[(2 * PH + 4) % 6]);
    decode(raw[PH][1], win[(2 * PH + 5) % 6]);
    load_row(r + 8, raw[PH][0]);
    load_row(r + 9, raw[PH][1]);
    if (r > r_last) return;                            // wave-uniform
    if (i == 3) MI_SSTAMP(2);
    WinRow w6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) w6[k] = win[(2 * PH + k) % 6];
    static_for<0, 2>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      const int row = r + I;
      if (row > r_last) return;
#if MI_STREAM_PRIO
      if (younger == (I == 1)) asm volatile("s_setprio 1"); else asm volatile("s_setprio 0");   // (strm::stream_kernel)
#endif
      float v[24];
      accumulate_row<PR, PC, I, true>(w6, wq, v);
      if (row < 2 || row >= p.H - 2) border_fix_rows<PR, PC, I>(v, tile::inside_mask(row, p.H), is_left, is_right);
      else if (any_left || any_right) border_fix_cols<PR, PC, I>(v, is_left, is_right, any_left, any_right);
      if (p.has_ccm) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float x = v[3 * k], y = v[3 * k + 1], z = v[3 * k + 2];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch)
            v[3 * k + ch] = (p.ccm[3 * ch] * x + p.ccm[3 * ch + 1] * y) + p.ccm[3 * ch + 2] * z;
        }
      }
      // the full-resolution pixel as the reference stores it (clamped, rounded to f16), 8 bytes per pixel
      uint4 px[4];
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const uint32_t rg0 = tile::cvt_pk_f16_clamp01(v[3 * k], v[3 * k + 1]), b0 = tile::cvt_pk_f16_clamp01(v[3 * k + 2], 0.f);
        const uint32_t rg1 = tile::cvt_pk_f16_clamp01(v[3 * k + 3], v[3 * k + 4]), b1 = tile::cvt_pk_f16_clamp01(v[3 * k + 5], 0.f);
        px[k / 2] = make_uint4(rg0, b0, rg1, b1);
      }
      uint4* dstrow = ring[row % RING] + lane * (LANE_PITCH / 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) dstrow[k] = px[k];
    });
    __builtin_amdgcn_wave_barrier();
    if (i == 3) MI_SSTAMP(3);
    // rows up to min(r + 1, r_last) are in the ring: emit the destination rows whose quads end there
    const int have = r + 1 < r_last ? r + 1 : r_last;
    while (next_r < rd_end) {
      const int qr = quad_origin(next_r, a.s0, p.H);
      if (qr + 1 > have) break;
      emit(next_r);
      ++next_r;
    }
    __builtin_amdgcn_wave_barrier();
    if (i == 3) MI_SSTAMP(4);
  };

  const int n_pairs = (r_last + 1 - r_begin + 1) / 2;
  for (int i = 0; i < n_pairs; i += 3) {
    body(std::integral_constant<int, 0>{}, i);
    body(std::integral_constant<int, 1>{}, i + 1);
    body(std::integral_constant<int, 2>{}, i + 2);
  }
#if MI_STREAM_PRIO
  asm volatile("s_setprio 0");
#endif
  MI_SSTAMP(5);
}

// ---- host side ---------------------------------------------------------------------------------
static inline bool supported(const Params& p, int work_dtype, const void* dst, int Hd, int Wd, float s0, float s1) {
  return work_dtype == MI_F16 && strm::supported(p, work_dtype) && p.H >= 2 && p.W >= 8 && Hd > 0 && Wd > 0 && s0 > 0.f &&
         s1 > 0.f && ((uintptr_t)dst & 7) == 0 && (int64_t)Hd * Wd * 6 < (int64_t)INVALID_OFF;
}

static inline void geometry(int H, int W, RSArgs& a) {
  const int units = W / 8;
  a.bands_x = units > 2 ? (units - 2 + 61) / 62 : 1;
  a.stride_units = units > 2 ? (units - 2 + a.bands_x - 1) / a.bands_x : 1;
  // the last band may own two units more than the others need: its wave still fits (stride_units + 2 <= 64 lanes cover
  // every unit because bands_x * stride_units >= units - 2).
  // align4: a band borrows up to 3 destination columns from its right neighbour; their quads reach at most
  // 3 * ceil(1 / s1) + 1 source columns past the band's own ones, which the second extra unit must hold
  const int step = (int)ceilf(1.0f / a.s1);
  a.align4 = 3 * step + 2 <= 16 ? 1 : 0;
  int rpw = (int)(((long)H * a.bands_x + 2047) / 2048);
  rpw = (rpw + 1) / 2 * 2;
  if (rpw < 4) rpw = 4;
  a.rows_per_wave = rpw;
  const int bands_y = (H + rpw - 1) / rpw;
  a.n_waves = a.bands_x * bands_y;
  a.n_blocks = (a.n_waves + WAVES - 1) / WAVES;
}

int launch_rggb(const RSArgs& a, hipStream_t stream);
int launch_grbg(const RSArgs& a, hipStream_t stream);
int launch_gbrg(const RSArgs& a, hipStream_t stream);
int launch_bggr(const RSArgs& a, hipStream_t stream);
static inline int launch(const RSArgs& a, int pattern, hipStream_t stream) {
  switch (pattern) {
    case MI_RGGB: return launch_rggb(a, stream);
    case MI_GRBG: return launch_grbg(a, stream);
    case MI_GBRG: return launch_gbrg(a, stream);
    default: return launch_bggr(a, stream);
  }
}

}  // namespace rstrm
