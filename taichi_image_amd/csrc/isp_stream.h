// Streaming Bayer demosaic for packed 12-bit frames: the register-resident form of bayer.py:115-177.
//
// One wave64 owns a band of 512 columns (8 pixels = one 12-byte packed unit per lane) and walks down
// `rows_per_wave` rows of it.  Nothing of the stencil goes through LDS:
//   * vertical taps: the six decoded CFA rows a row pair needs (r-2 .. r+3) stay in registers as a ring
//     of 6 x 12 fp32 values per lane; each iteration decodes two new rows and retires two (the loop body
//     exists in three rotations so that the ring is addressed statically, no register moves);
//   * horizontal taps: the two pixels on either side of a lane's unit come from the neighbouring lanes
//     with DPP wave shifts (v_mov_b32 wave_shr:1 / wave_shl:1), four moves per row; lanes 0 and 63 take
//     them from one extra dword of the packed row instead;
//   * packed rows are loaded three row pairs ahead of their use (global_load_dwordx3 per lane and row,
//     contiguous across the wave) - the only wait in steady state is the one software-pipelined load;
//   * no workgroup barrier: the four waves of a block are independent until the final reduction.
// LDS is used only to transpose a wave's output row (64 lanes x 24 elements) into wave-contiguous
// 16-byte stores (scratch/store_bench.hip: 6.2 vs 3.7 TB/s for the per-lane 3 x 16 B pattern).
//
// Arithmetic: identical to tile::tile_kernel (same accumulate<>, same border renormalisation table) and
// therefore bit-exact against oracle/isp_oracle.py:bayer_to_rgb.
//
// Epilogues (one kernel per value): the passes of the fused config-2 chain (test/pipeline.py:26-32,
// tonemap.py:135-168) that re-derive the demosaiced image from the packed frame instead of storing and
// re-reading a 6 B/px intermediate, and the plain store of ISP.load_packed12 (camera_isp.py:333-340).
#pragma once
#include "isp_tile.h"
#include "isp_finalize.h"
#include <stdlib.h>

#pragma clang fp contract(off)

namespace strm {

using tile::Params;
using tile::static_for;

constexpr int THREADS = 256, WAVES = THREADS / 64;
constexpr int BAND = 512;                       // columns per wave: 64 lanes x 8 px
constexpr int STAGE_U4 = 64 * 7;                // per-wave store staging: 64 lanes x 6 x 16 B (4-byte outputs), slot pitch 7

enum Epi {
  S_STORE = 0,         // demosaic -> work-dtype RGB image (ISP.load_packed12)
  S_BOUNDS = 1,        // bounds of the work-dtype image (tonemap.py:146) + the statistics of :147-149 computed
                       // on the assumption that the bounds turn out to be exactly (0, 1) (see S_STATS)
  S_STATS = 2,         // statistics of the normalised image (tonemap.py:147-149) for bounds other than (0, 1):
                       // exits at once when S_BOUNDS' assumption held
  S_RH_MINMAX = 3,     // bounds of the Reinhard image (tonemap.py:150-153)
  S_RH_STORE = 4,      // final map (tonemap.py:154), any output dtype
  S_STORE_BOUNDS = 5   // S_STORE + S_BOUNDS in one pass (the "cached" chain's first pass)
};

// Partial rows (SoA, stride part_stride) a pass leaves, one entry per block:
//   S_BOUNDS / S_STORE_BOUNDS: rows 0,1 = min,max; rows 2..8 = gmin,gmax,slog2,sgray,s0,s1,s2 (speculative)
//   S_STATS:                   rows 2..8 (overwrites the speculative ones)
//   S_RH_MINMAX:               rows 9,10
constexpr int ROW_BOUNDS = 0, ROW_STATS = 2, ROW_BOUNDS2 = 9;
constexpr int LOAD_BATCH = 8;                       // frames per batched S_STORE / resize launch
// rows 20..47: the tagged rows of the whole-frame kernel's grid barriers (isp_mega.h)
constexpr int MEGA_ROW_BASE = 20, PART_ROWS = 48;

struct SArgs {
  Params t;
  int rows_per_wave;      // even
  int bands_x, n_waves;
  int n_blocks;           // == partial count of every pass of this geometry
  float n_px, intensity;  // for the pulled finalize steps
  float* fp_w;            // FrameParams (device), written by block 0 of the pulling passes
  int bounds_post;        // FIN_BOUNDS post-processing of the raw bounds: 1 = clamp, 2 = clamp + f16 rounding
  // S_STORE: the stride-8 subsample ISP.update_metering gathers from the image (camera_isp.py:168-170: image[::8, ::8]),
  // written on the way - (ceil(H / 8), ceil(W / 8), 3) work-dtype elements - or NULL.  A lane's unit starts at a
  // multiple of 8 columns, so the sample of a row r % 8 == 0 is the lane's first pixel.
  void* sub;
  int sub_w;              // ceil(W / 8)
  // S_STORE, several frames in one launch (grid.y = frame; mi_isp_load_packed_batch: the cameras of a group): frame y reads
  // srcs[y], writes dsts[y] (and subs[y]).  n_batch == 0: one frame, t.src / t.dst / sub.  Indexed by blockIdx.y itself
  // (a computed index would send this struct through scratch).
  int n_batch;
  const void* srcs[LOAD_BATCH];
  void* dsts[LOAD_BATCH];
  void* subs[LOAD_BATCH];
};

// ---------------------------------------------------------------------------------------------
// wave shifts: lane i takes the value of lane i-1 (shr) / i+1 (shl); lane 0 / lane 63 keep `old`
// ---------------------------------------------------------------------------------------------
MI_DEV float from_left(float v, float old) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}
MI_DEV float from_right(float v, float old) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                0x130 /* wave_shl:1 */, 0xF, 0xF, false));
}

// VALU cost model of gfx950 these kernels are written to (scratch/occ_bench.hip, scratch/issue_bench.hip; cycles of
// one SIMD per wave64 instruction with two or more waves resident, ~2.1 GHz under load):
//   2 cycles: VOP1 / VOP2 encodings with VGPR, literal or inline-constant operands - v_fmac/mul/add/sub_f32, v_and/or,
//             shifts, v_mov, v_add_u32, v_fmamk/fmaak;
//   4 cycles: everything else - any instruction with an SGPR operand, VOP3 / VOP3P / SDWA / DPP forms, v_min/max(3),
//             every v_cvt_*, v_bfe, v_perm, v_alignbit, v_fma_mix_f32 (the f16-window variant of this kernel that read
//             halves through v_fma_mix ran no faster than converting once: half-rate instructions);
//   8 cycles: v_log/exp/rcp_f32.
// So: weights and scalars live in VGPRs, the window is fp32 and the 21 FMAs per pixel are plain v_fmac_f32.

// the eight distinct demosaic weights / 16, in tile::wq_index order
__host__ __device__ constexpr float wq_value(int i) {
  return (i == 0 ? -3.f : i == 1 ? -2.f : i == 2 ? 1.f : i == 3 ? 4.f : i == 4 ? 8.f : i == 5 ? 10.f : i == 6 ? 12.f : 16.f) * 0.0625f;
}

// an opaque VGPR copy of a uniform value: keeps the compiler from folding it back into an SGPR / literal operand
MI_DEV float vgpr(float x) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x)); return r; }

// The window row of a lane: image columns c0-2 .. c0+9 of one row, work-dtype values widened to fp32.
struct WinRow { float v[12]; };

// one packed row of this lane: d[0..2] = its 8 pixels, d[3] = the dword holding the two pixels beyond the
// band's edge (lanes 0 and 63 only) -> window row.  The scaled decode of packed.py:98-100 - f32(v) * f32(1/4095),
// rounded to the work dtype, widened back - has 4096 possible results: they are tabulated once per block in LDS
// (16 KB, filled with exactly that arithmetic) and a pixel costs one ds_read_b32 instead of four half-rate
// conversions; the table index is the unpacked 12-bit value itself.
MI_DEV void decode_row(const uint32_t (&d)[4], const float (&lut)[4096], int lane, WinRow& row) {
  uint32_t v[8];
  tile::unpack12x8(d[0], d[1], d[2], false, v);
  // lane 0: the last pixel pair of the unit to the left = the upper 3 bytes of the dword before this unit;
  // lane 63: the first pair of the unit to the right = the lower 3 bytes of the dword after it
  const uint32_t w = lane == 0 ? d[3] >> 8 : d[3] & 0xFFFFFFu;
  float own[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) own[i] = lut[v[i]];
  const float x0 = lut[w & 0xFFFu], x1 = lut[w >> 12];
  row.v[0] = from_left(own[6], x0);
  row.v[1] = from_left(own[7], x1);
#pragma unroll
  for (int j = 0; j < 8; ++j) row.v[2 + j] = own[j];
  row.v[10] = from_right(own[0], x0);
  row.v[11] = from_right(own[1], x1);
}

// filter_at (bayer.py:138-155) for the 8 pixels of strip row I of a 6-row window: per pixel and channel a sequential
// fp32 accumulation over the non-zero taps in reference order with the weights w/16 (as tile::accumulate).  The loop
// nest is taps outermost, pixels and channels inside: consecutive instructions belong to different accumulation
// chains (up to 24 of them), so no instruction waits for its predecessor's result; within each chain the order of
// the additions is unchanged.  wq: the eight distinct weights / 16 in VGPRs (tile::wq_index).
template <int PR, int PC, int I, bool EXACT>
MI_DEV void accumulate_row(const WinRow (&win)[6], const float (&wq)[8], float (&v)[24]) {
  bool first[24];
#pragma unroll
  for (int j = 0; j < 24; ++j) first[j] = true;
  static_for<0, 13>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    constexpr int row = I + 2 + tile::TAP_DR[t];
    static_for<0, 8>([&](auto kc) {
      constexpr int K = decltype(kc)::value;
      constexpr int KIDX = ((I + PR) & 1) + 2 * ((K + PC) & 1);
      constexpr int col = K + 2 + tile::TAP_DC[t];
      static_for<0, 3>([&](auto cc) {
        constexpr int ch = decltype(cc)::value;
        constexpr int wi = tile::KW[KIDX][t][ch];
        if constexpr (wi != 0) {
          const float w = wq[tile::wq_index(wi)];
          const float x = win[row].v[col];
          float& acc = v[3 * K + ch];
          if (first[3 * K + ch]) acc = x * w;                               // == fma(x, w, +0)
          else if constexpr (EXACT) acc = __builtin_fmaf(x, w, acc);
          else acc = acc + x * w;
          first[3 * K + ch] = false;
        }
      });
    });
  });
}

// ---------------------------------------------------------------------------------------------
// wave-contiguous store of one output row of the band: lane l holds elements [24 l, 24 l + 24) as N 16-byte
// (8-byte for u8) units.  The units leave through a raw buffer resource of the output image: `lane_off[j]` is the
// byte offset of unit j * 64 + lane inside the band's row, or an offset beyond the image for units of lanes right
// of it (such stores are dropped by the hardware - no branch around the store), `row_base` the byte offset of the
// band's row (a scalar).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t INVALID_OFF = 0x40000000u;       // 1 GiB: beyond any frame the stream kernels accept

// AUX: cache policy bits of the store.  ST_STREAM = nt ("non-temporal"): a final output nobody on the chip reads again is
// streamed to memory instead of being allocated in the write-back L2 - measured on the whole-frame kernel's phase D
// (75.5 MB per frame, write-bound): 49.9 -> 46.7 us per frame (`sc1`, write-through, gives the same; nt on the packed
// frame's LOADS gives nothing).  ST_KEEP = the default policy, for images the next kernel reads back.
constexpr int ST_KEEP = 0, ST_STREAM = 2;
#ifndef MI_ST_LOAD                 /* the image ISP.load_packed12 stores (the metering and the first tonemap pass read it back) */
#define MI_ST_LOAD ST_KEEP
#endif
template <int UB, int AUX> MI_DEV void buffer_store_unit(__amdgpu_buffer_rsrc_t rsrc, const void* val, uint32_t voff, uint32_t soff) {
  if constexpr (UB == 16) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 x; __builtin_memcpy(&x, val, 16);
    __builtin_amdgcn_raw_buffer_store_b128(x, rsrc, voff, soff, AUX);
  } else {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 x; __builtin_memcpy(&x, val, 8);
    __builtin_amdgcn_raw_buffer_store_b64(x, rsrc, voff, soff, AUX);
  }
}

template <class U, int N, int AUX = ST_KEEP>
MI_DEV void wave_store_units(__amdgpu_buffer_rsrc_t rsrc, uint32_t row_base, const uint32_t (&lane_off)[6], int lane,
                             uint4* stage_, const U (&mine)[N]) {
  U* lb = reinterpret_cast<U*>(stage_);
  // 6 units per lane (4-byte outputs): slots padded to 7 so that consecutive lanes' 16-byte writes fall into
  // different bank groups (as tile::StagePitch)
  constexpr int P = N == 6 ? 7 : N;
#pragma unroll
  for (int j = 0; j < N; ++j) lb[lane * P + j] = mine[j];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int q = j * 64 + lane;
    const int sl = N == 6 ? q / 6 : 0;
    const U val = N == 6 ? lb[sl * P + (q - sl * 6)] : lb[q];
    buffer_store_unit<(int)sizeof(U), AUX>(rsrc, &val, lane_off[j], row_base);
  }
  __builtin_amdgcn_wave_barrier();
}

template <class T, int AUX = ST_KEEP>
MI_DEV void wave_store_row_t(__amdgpu_buffer_rsrc_t rsrc, uint32_t row_base, const uint32_t (&lane_off)[6], int lane,
                             uint4* stage, const float (&v)[24]) {
  typedef typename IoUnit<T>::type U;
  constexpr int N = IoUnits<T>::value;
  T o[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) o[i] = cast_out<T>(v[i]);
  U mine[N];
  __builtin_memcpy(mine, o, sizeof(mine));
  wave_store_units<U, N, AUX>(rsrc, row_base, lane_off, lane, stage, mine);
}

// ---------------------------------------------------------------------------------------------
// statistics of tonemap.py:78-103, two pixels per call.  min / max are taken on the gray values themselves and the
// clamp max(gray, 1e-4) (monotone) is applied once to the reduced values; the channel sums are only accumulated when
// something consumes them (RGB: color_adapt != 0 - otherwise mean3 == gray_mean, tonemap.py:119).
// ---------------------------------------------------------------------------------------------
#ifndef MI_STREAM_PRIO
#define MI_STREAM_PRIO 1
#endif
#pragma clang fp contract(fast)
struct Stats2 {
  float gmin, gmax, slog, sgray, s0, s1, s2;
  MI_DEV void init() {
    gmin = __builtin_inff(); gmax = -__builtin_inff();
    slog = sgray = s0 = s1 = s2 = 0.f;
  }
  template <bool RGB>
  MI_DEV void add2(float a0, float a1, float a2, float b0, float b1, float b2) {
    const float ga = rgb_gray(a0, a1, a2), gb = rgb_gray(b0, b1, b2);
    gmin = __builtin_fminf(gmin, __builtin_fminf(ga, gb));          // v_min3_f32
    gmax = __builtin_fmaxf(gmax, __builtin_fmaxf(ga, gb));
    slog += hw_log2(fmaxf(ga, 1e-4f));            // sum of log2; scaled by ln2 when combined
    slog += hw_log2(fmaxf(gb, 1e-4f));
    sgray += ga; sgray += gb;
    if constexpr (RGB) { s0 += a0; s0 += b0; s1 += a1; s1 += b1; s2 += a2; s2 += b2; }
  }
  // a lane's row of eight pixels: one logarithm (of the product of the clamped gray values) instead of eight - add8_gray
  template <bool RGB>
  MI_DEV void add8(const float (&t)[24]) {
    float c[8];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const float ga = rgb_gray(t[3 * k], t[3 * k + 1], t[3 * k + 2]), gb = rgb_gray(t[3 * k + 3], t[3 * k + 4], t[3 * k + 5]);
      gmin = __builtin_fminf(gmin, __builtin_fminf(ga, gb));          // v_min3_f32
      gmax = __builtin_fmaxf(gmax, __builtin_fmaxf(ga, gb));
      c[k] = fmaxf(ga, 1e-4f); c[k + 1] = fmaxf(gb, 1e-4f);
      sgray += ga; sgray += gb;
      if constexpr (RGB) { s0 += t[3 * k]; s0 += t[3 * k + 3]; s1 += t[3 * k + 1]; s1 += t[3 * k + 4]; s2 += t[3 * k + 2]; s2 += t[3 * k + 5]; }
    }
    slog += hw_log2(((c[0] * c[1]) * (c[2] * c[3])) * ((c[4] * c[5]) * (c[6] * c[7])));
  }
  // the same from two gray values (the whole-frame kernel takes them straight from the packed f16 pixels)
  // (min3 / max3 / max as asm: ga and gb come out of inline asm, and for values of unknown origin the compiler quiets
  // possible signalling NaNs with a v_max x, x before every min / max - 1.5 instructions per pixel for nothing)
  MI_DEV void add2_gray(float ga, float gb) {
    asm("v_min3_f32 %0, %0, %1, %2" : "+v"(gmin) : "v"(ga), "v"(gb));
    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(gmax) : "v"(ga), "v"(gb));
    float ca, cb;
    asm("v_max_f32 %0, 0x38d1b717, %1" : "=v"(ca) : "v"(ga));          // max(gray, 1e-4)
    asm("v_max_f32 %0, 0x38d1b717, %1" : "=v"(cb) : "v"(gb));
    slog += hw_log2(ca);
    slog += hw_log2(cb);
    sgray += ga; sgray += gb;
  }
  // The eight gray values of a lane's row at once.  sum(log2 g) = log2(prod g): one logarithm per row instead of eight
  // (a transcendental issues at a fraction of the rate of a multiply).  The clamped values lie in [1e-4, ~1], so the
  // product of eight stays above 1e-32 - a normal fp32 number; seven roundings of 2^-24 move log2 of the product by
  // ~6e-7 absolute, the same order as the error of eight hardware logarithms (~1e-7 each, relative to values up to 13),
  // and the scalars' contract is 1e-4.
  MI_DEV void add8_gray(const float (&g)[8]) {
    float c[8];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      asm("v_min3_f32 %0, %0, %1, %2" : "+v"(gmin) : "v"(g[k]), "v"(g[k + 1]));
      asm("v_max3_f32 %0, %0, %1, %2" : "+v"(gmax) : "v"(g[k]), "v"(g[k + 1]));
      asm("v_max_f32 %0, 0x38d1b717, %1" : "=v"(c[k]) : "v"(g[k]));          // max(gray, 1e-4)
      asm("v_max_f32 %0, 0x38d1b717, %1" : "=v"(c[k + 1]) : "v"(g[k + 1]));
      sgray += g[k]; sgray += g[k + 1];
    }
    slog += hw_log2(((c[0] * c[1]) * (c[2] * c[3])) * ((c[4] * c[5]) * (c[6] * c[7])));
  }
  MI_DEV void finish() { gmin = fmaxf(gmin, 1e-4f); gmax = fmaxf(gmax, 1e-4f); }
};
#pragma clang fp contract(off)

// Block reduction without a barrier: every wave reduces its values in registers (DPP), leaves them in its own LDS
// row and counts itself in; the wave that arrives last combines the rows in wave order (deterministic) and stores the
// block's partial.  Waves that finish early retire instead of waiting for the slowest one (the edge bands).
// `arrived` must be zero before the first arrival (set by the kernel prologue, followed by its only barrier).
template <int NV>
MI_DEV void block_reduce_store_nb(const float (&v)[NV], const int (&op)[NV], float (*red)[16], unsigned* arrived,
                                  float* partials, int stride, int block, int wave, int lane) {
  static_assert(NV <= 16, "staging row too narrow");
  float r[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) r[k] = op[k] == 0 ? wave_min(v[k]) : (op[k] == 1 ? wave_max(v[k]) : wave_sum(v[k]));
  unsigned before = 0;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) red[wave][k] = r[k];
    before = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  before = __builtin_amdgcn_readfirstlane(before);
  if (before == WAVES - 1) {                          // wave-uniform: this wave arrived last
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      float x = red[0][k];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) {
        const float o = red[w][k];
        x = op[k] == 0 ? fminf(x, o) : (op[k] == 1 ? fmaxf(x, o) : x + o);
      }
      mine = lane == k ? x : mine;
    }
    if (lane < NV) partials[(size_t)lane * stride + block] = mine;
  }
}

// ---------------------------------------------------------------------------------------------
// pulled finalize: every block folds the producer's per-block partials itself (identical arithmetic in
// every block) instead of a one-block launch between two passes; sh_fp = the FrameParams to work with.
// FIN_BOUNDS: rows {min, max}; FIN_STATS: 7 rows; FIN_BOUNDS2: rows {min, max}.
// ---------------------------------------------------------------------------------------------
template <int FIN>
MI_DEV void pull(const SArgs& a, const float* rows, float* sh_fp, double (*sh_tot)[WAVES]) {
  constexpr int nrows = FIN == ew::FIN_STATS ? 7 : 2;
  const float fp_mine = threadIdx.x < FP_COUNT ? a.t.fp[threadIdx.x] : 0.f;
  float mn = __builtin_inff(), mx = -__builtin_inff();
  double sum[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  constexpr int U = nrows == 2 ? 4 : 2;
  for (int base = 0; base < a.n_blocks; base += U * THREADS) {
    float v[U][nrows];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * THREADS + threadIdx.x;
      const bool ok = i < a.n_blocks;
#pragma unroll
      for (int k = 0; k < nrows; ++k)
        v[u][k] = ok ? rows[(size_t)k * a.t.part_stride + i] : (k == 0 ? __builtin_inff() : k == 1 ? -__builtin_inff() : 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      mn = fminf(mn, v[u][0]); mx = fmaxf(mx, v[u][1]);
#pragma unroll
      for (int k = 2; k < nrows; ++k) sum[k - 2] += (double)v[u][k];
    }
  }
  if (threadIdx.x < FP_COUNT) sh_fp[threadIdx.x] = fp_mine;
  mn = wave_min(mn); mx = wave_max(mx);
  if (nrows == 7) {
#pragma unroll
    for (int k = 0; k < 5; ++k) sum[k] = wave_sum(sum[k]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh_tot[0][wave] = (double)mn; sh_tot[1][wave] = (double)mx;
    if (nrows == 7) {
#pragma unroll
      for (int k = 0; k < 5; ++k) sh_tot[k + 2][wave] = sum[k];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < nrows; ++k) {
      double r = sh_tot[k][0];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) r = k == 0 ? fmin(r, sh_tot[k][w]) : (k == 1 ? fmax(r, sh_tot[k][w]) : r + sh_tot[k][w]);
      tot[k] = r;
    }
    ew::FinArgs fa = {};
    fa.fp = sh_fp; fa.n_px = a.n_px; fa.intensity = a.intensity; fa.la = a.t.la; fa.ca = a.t.ca;
    fa.bounds_post = a.bounds_post;
    ew::finalize_scalars<true>(FIN, fa, tot);
  }
  __syncthreads();
}


// ---- border renormalisation (bayer.py:143-155): v = acc * 16 / t for the pixels whose 13-tap diamond leaves the
// image, t = the in-bounds weight sum (tile::BORDER_T).  No vector-memory lookups here: a load in this path would
// make the wave drain its prefetched rows (s_waitcnt vmcnt(0)) - measured 3x on the edge bands of every frame.

// x * 16 / T for a compile-time weight sum T: q = a * RN(1/T) with one exact residual step (a = 16 x).  Equal to the
// IEEE quotient a / T for every finite float a with a normal quotient (|a| >= 2^-120) or a == +0 and every T that occurs
// (exhaustive check over all 2^32 bit patterns: tests/test_oracle.py::test_border_division_by_reciprocal runs
// oracle/check_recip_div.c; a pixel accumulator is zero or at least 2^-29 in magnitude).
template <int T> MI_DEV float div16_by(float x) {
  if constexpr (T == 16) {
    return x;
  } else {
    constexpr float d = (float)T, y = 1.0f / (float)T;
    const float a = x * 16.f;
    const float q = a * y;
    const float e = __builtin_fmaf(-q, d, a);
    return __builtin_fmaf(e, y, q);
  }
}

// rows whose vertical taps are all inside the image: only pixels 0, 1 of the image's first unit (lane `is_left`) and
// pixels 6, 7 of its last unit (`is_right`) are border pixels, and their weight sums are compile-time constants
template <int PR, int PC, int I>
MI_DEV void border_fix_cols(float (&v)[24], bool is_left, bool is_right, bool any_left, bool any_right) {
  auto fix = [&](auto kc, auto cm, bool mine) {
    constexpr int k = decltype(kc)::value, cmask = decltype(cm)::value;
    constexpr int KIDX = ((I + PR) & 1) + 2 * ((k + PC) & 1);
    constexpr uint32_t tq = tile::BORDER_T.t[KIDX][31][cmask];
    static_for<0, 3>([&](auto cc) {
      constexpr int ch = decltype(cc)::value;
      constexpr int T = (int)(int8_t)(tq >> (8 * ch));
      if constexpr (T != 16) v[3 * k + ch] = mine ? div16_by<T>(v[3 * k + ch]) : v[3 * k + ch];
    });
  };
  if (any_left) {                                     // wave-uniform
    fix(std::integral_constant<int, 0>{}, std::integral_constant<int, 28>{}, is_left);   // offsets -2, -1 outside
    fix(std::integral_constant<int, 1>{}, std::integral_constant<int, 30>{}, is_left);   // offset -2 outside
  }
  if (any_right) {
    fix(std::integral_constant<int, 6>{}, std::integral_constant<int, 15>{}, is_right);  // offset +2 outside
    fix(std::integral_constant<int, 7>{}, std::integral_constant<int, 7>{}, is_right);   // offsets +1, +2 outside
  }
}

// {T, RN(1/T)} per [kernel][row mask slot][column mask slot][channel], masks in the order {31, 28, 30, 15, 7} (inside,
// offsets -2 -1 outside, -2 outside, +2 outside, +1 +2 outside): the only masks an image of at least 4 rows and
// columns has.  Compile-time only: the kernels take their divisors from it as template constants.
struct DY { float d, y; };
struct BorderDY { DY t[4][5][5][3]; };
constexpr BorderDY make_border_dy() {
  constexpr int MASKS[5] = {31, 28, 30, 15, 7};
  constexpr tile::BorderTable b = tile::make_border_table();
  BorderDY r = {};
  for (int k = 0; k < 4; ++k)
    for (int ri = 0; ri < 5; ++ri)
      for (int ci = 0; ci < 5; ++ci)
        for (int ch = 0; ch < 3; ++ch) {
          const int T = (int)(int8_t)(b.t[k][MASKS[ri]][MASKS[ci]] >> (8 * ch));
          r.t[k][ri][ci][ch].d = (float)T;
          r.t[k][ri][ci][ch].y = 1.0f / (float)T;     // 10 <= T <= 22 (static_assert below)
        }
  return r;
}
constexpr bool border_dy_in_range() {
  constexpr BorderDY r = make_border_dy();
  for (int k = 0; k < 4; ++k)
    for (int ri = 0; ri < 5; ++ri)
      for (int ci = 0; ci < 5; ++ci)
        for (int ch = 0; ch < 3; ++ch)
          if (r.t[k][ri][ci][ch].d < 10.f || r.t[k][ri][ci][ch].d > 22.f) return false;
  return true;
}
static_assert(border_dy_in_range(), "a weight sum outside the range oracle/check_recip_div.c checks");
constexpr BorderDY BORDER_DY = make_border_dy();

// the image's first and last two rows (rmask != 31, wave-uniform): every pixel of the row is renormalised with the
// reciprocal sequence of div16_by<T>, all sums compile-time constants.  The row's parity I fixes which two rows it can
// be (H is even and every wave starts on an even row: strm::supported, geometry): I = 0 -> row 0 (mask 28) or row
// H - 2 (mask 15), I = 1 -> row 1 (30) or row H - 1 (7); one wave-uniform branch picks the copy.  History: 24 IEEE
// divisions per row (+ 1.5 us on the band), then a {T, 1/T} table in constant memory behind scalar loads - the scalar
// cache misses cost a memory round trip per border row (+ 1 - 1.6 us on the top and bottom bands of the whole-frame
// kernel's phase A, which its first barrier waits for).
template <int PR, int PC, int I, int RS>
MI_DEV void border_fix_rows_ct(float (&v)[24], bool is_left, bool is_right) {
  static_for<0, 8>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int KIDX = ((I + PR) & 1) + 2 * ((k + PC) & 1);
    constexpr int cs = k == 0 ? 1 : k == 1 ? 2 : k == 6 ? 3 : k == 7 ? 4 : 0;     // this pixel's column-border slot, if any
    static_for<0, 3>([&](auto cc) {
      constexpr int ch = decltype(cc)::value;
      constexpr int Tm = (int)BORDER_DY.t[KIDX][RS][0][ch].d, Te = (int)BORDER_DY.t[KIDX][RS][cs][ch].d;
      const float x = v[3 * k + ch];
      float q = div16_by<Tm>(x);
      if constexpr (cs != 0 && Te != Tm) {
        const float qe = div16_by<Te>(x);
        q = (cs <= 2 ? is_left : is_right) ? qe : q;
      }
      v[3 * k + ch] = q;
    });
  });
}
template <int PR, int PC, int I>
MI_DEV void border_fix_rows(float (&v)[24], int rmask, bool is_left, bool is_right) {
  if (rmask == (I == 0 ? 28 : 30)) border_fix_rows_ct<PR, PC, I, I == 0 ? 1 : 2>(v, is_left, is_right);     // wave-uniform
  else border_fix_rows_ct<PR, PC, I, I == 0 ? 3 : 4>(v, is_left, is_right);
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// measurement aid (make EXTRA=-DMI_STREAM_STAMPS): lane 0 of every wave leaves s_memtime stamps of its phases in the
// workspace behind the partial rows (16 per wave, 32-bit); see scripts/stream_stamps.py
#ifdef MI_STREAM_STAMPS
// -DMI_STAMP_REALTIME: the 100 MHz counter all XCDs share (absolute timelines) instead of the per-CU cycle counter
#ifdef MI_STAMP_REALTIME
#define MI_STAMP_NOW() ((unsigned)__builtin_amdgcn_s_memrealtime())
#else
#define MI_STAMP_NOW() ((unsigned)__builtin_readcyclecounter())
#endif
#define MI_SSTAMP(i)                                                                                          \
  do {                                                                                                        \
    if (lane == 0 && wave_ok && p.partials)                                                                   \
      reinterpret_cast<unsigned*>(p.partials + (size_t)PART_ROWS * p.part_stride)[g * 16 + (i)] = MI_STAMP_NOW();  \
  } while (0)
#else
#define MI_SSTAMP(i) do {} while (0)
#define MI_STAMP_NOW() 0u
#endif

template <class E, int PR, int PC, int EPI>
__global__ __launch_bounds__(THREADS, 2) void stream_kernel(const SArgs a) {
  constexpr bool EXACT = sizeof(E) == 2;
  constexpr bool STORES = EPI == S_STORE || EPI == S_RH_STORE || EPI == S_STORE_BOUNDS;
  constexpr bool BOUNDS = EPI == S_BOUNDS || EPI == S_STORE_BOUNDS;
  const Params& p = a.t;
  __shared__ __attribute__((aligned(16))) uint4 stage_all[STORES ? WAVES : 1][STORES ? STAGE_U4 : 1];
  __shared__ float red[WAVES][16];
  __shared__ float sh_fp[FP_COUNT];
  __shared__ double sh_tot[7][WAVES];
  __shared__ unsigned arrived;
  __shared__ float lut[4096];                         // the decoded value of every 12-bit code (decode_row)
  if (threadIdx.x == 0) arrived = 0;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // a scalar: the row arithmetic below is SALU work
  const int g = blockIdx.x * WAVES + wave;
  const bool wave_ok = g < a.n_waves;
  const int by = g / a.bands_x, bx = g - by * a.bands_x;
  const int c0 = bx * BAND + lane * 8;
  const int r_begin = by * a.rows_per_wave;
  const int r_end = wave_ok ? (r_begin + a.rows_per_wave < p.H ? r_begin + a.rows_per_wave : p.H) : r_begin;
  const bool col_ok = wave_ok && c0 < p.W;
  const int active_lanes = !wave_ok ? 0 : (p.W - bx * BAND >= BAND ? 64 : (p.W - bx * BAND) / 8);
  uint4* stage = stage_all[STORES ? wave : 0];

  // Packed rows come through a raw buffer resource: an offset at or beyond the frame's size reads as zero, so rows
  // above / below the image, lanes right of it and the edge dword of the 62 lanes that do not need one are all
  // "loaded" by the same two unconditional instructions (no branches: the compiler can count the loads in flight,
  // and zero bits decode to 0, i.e. out-of-image taps contribute 0 * w).  Offsets: a per-lane column part plus a
  // per-row scalar part, either of which is INVALID (1 GiB, beyond any frame) when its coordinate is outside.
  constexpr uint32_t INVALID = INVALID_OFF;
  const uint32_t pitch = (uint32_t)p.W * 3 / 2;
  const bool batched = EPI == S_STORE && a.n_batch > 0;
  const void* const src_p = batched ? a.srcs[blockIdx.y] : p.src;
  void* const dst_p = batched ? a.dsts[blockIdx.y] : p.dst;
  void* const sub_p = batched ? a.subs[blockIdx.y] : a.sub;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src_p), 0, (int)((uint32_t)p.H * pitch), 0x00020000);
  const uint32_t col_off = col_ok ? (uint32_t)c0 * 3 / 2 : INVALID;
  // lane 0: the dword before its unit (pixels c0-2, c0-1 in its upper 3 bytes); lane 63: the dword after (c0+8, c0+9)
  const bool ext_ok = col_ok && ((lane == 0 && c0 > 0) || (lane == 63 && c0 + 8 < p.W));
  const uint32_t ext_off = ext_ok ? (uint32_t)c0 * 3 / 2 + (lane == 0 ? -4 : 12) : INVALID;

  // output rows leave through a buffer resource too (wave_store_units): unit j * 64 + lane of the band's row
  const int osz = EPI == S_RH_STORE ? (int)mi_dtype_size_dev(p.out_dtype) : (int)sizeof(E);
  const int unit_bytes = osz == 1 ? 8 : 16, units_per_lane = 24 * osz / unit_bytes;
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(
      dst_p, 0, STORES ? (int)((uint32_t)p.H * (uint32_t)p.W * 3u * (uint32_t)osz) : 0, 0x00020000);
  uint32_t lane_off[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
    lane_off[j] = (j * 64 + lane) < active_lanes * units_per_lane ? (uint32_t)(j * 64 + lane) * unit_bytes : INVALID_OFF;
  const uint32_t out_pitch = (uint32_t)p.W * 3u * (uint32_t)osz, band_base = (uint32_t)bx * BAND * 3u * (uint32_t)osz;

  auto load_row = [&](int r, uint32_t (&d)[4]) {
    const uint32_t row_off = (r >= 0 && r < p.H && r < r_end + 2) ? (uint32_t)r * pitch : INVALID;     // scalar
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    const u3 q = __builtin_amdgcn_raw_buffer_load_b96(rsrc, col_off + row_off, 0, 0);
    d[0] = q.x; d[1] = q.y; d[2] = q.z;
    d[3] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, ext_off + row_off, 0, 0);
  };

  MI_SSTAMP(0);
  // ---- prologue: every load of the first four window rows and of the first three row pairs in flight ----
  if constexpr (EPI == S_STATS) {
    // the bounds first: when S_BOUNDS' assumption held there is nothing to do (block 0 still publishes them)
    pull<ew::FIN_BOUNDS>(a, p.partials + (size_t)ROW_BOUNDS * p.part_stride, sh_fp, sh_tot);
    if (blockIdx.x == 0 && threadIdx.x < FP_COUNT && a.fp_w && ew::finalize_writes(ew::FIN_BOUNDS, threadIdx.x))
      a.fp_w[threadIdx.x] = sh_fp[threadIdx.x];
    if (sh_fp[FP_LO] == 0.f && sh_fp[FP_INV] == 1.f) return;
  }
  // uniform operands of the hot loop as VGPRs (an SGPR operand halves a VALU instruction's rate)
  float wq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wq[i] = vgpr(wq_value(i));
  WinRow win[6];                                     // ring: image row (r_begin - 2 + q) lives in slot q % 6
  uint32_t raw[3][2][4];                             // ring: row pair j (rows r_begin + 2 + 2j, + 3 + 2j) in slot j % 3
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
    MI_SSTAMP(1);
    __syncthreads();                                  // the table is complete, `arrived` is zero for everyone; the kernel's only barrier besides pull()
    // the pass's scalars while the loads fly
    if constexpr (EPI == S_RH_MINMAX) pull<ew::FIN_STATS>(a, p.partials + (size_t)ROW_STATS * p.part_stride, sh_fp, sh_tot);
    if constexpr (EPI == S_RH_STORE) pull<ew::FIN_BOUNDS2>(a, p.partials + (size_t)ROW_BOUNDS2 * p.part_stride, sh_fp, sh_tot);
#pragma unroll
    for (int q = 0; q < 4; ++q) decode_row(pro[q], lut, lane, win[q]);
  }
  MI_SSTAMP(2);

  float lo = 0.f, inv = 1.f, lo2 = 0.f, inv2 = 1.f;
  ReinhardK rk;
  rk.la = p.la; rk.ca = p.ca; rk.map_key = 1.f; rk.ei = 1.f; rk.mean3[0] = rk.mean3[1] = rk.mean3[2] = 0.f;
  if constexpr (EPI == S_STATS || EPI == S_RH_MINMAX || EPI == S_RH_STORE) { lo = sh_fp[FP_LO]; inv = sh_fp[FP_INV]; }
  // bounds exactly (0, 1): clamp((x - 0) * 1, 0, 1) is the identity on the clamped image
  const bool unit = lo == 0.f && inv == 1.f;
  const bool ca0 = p.ca == 0.f;
  // per-pixel operands live in VGPRs: a VALU instruction with an SGPR operand issues at half rate
  lo = vgpr(lo); inv = vgpr(inv);
  rk.la = vgpr(rk.la); rk.ca = vgpr(rk.ca);
  if constexpr (EPI == S_RH_MINMAX || EPI == S_RH_STORE) {
    rk.map_key = vgpr(sh_fp[FP_MAPKEY]); rk.ei = vgpr(sh_fp[FP_EI]);
    rk.mean3[0] = vgpr(sh_fp[FP_MEAN3]); rk.mean3[1] = vgpr(sh_fp[FP_MEAN3 + 1]); rk.mean3[2] = vgpr(sh_fp[FP_MEAN3 + 2]);
  }
  if constexpr (EPI == S_RH_STORE) { lo2 = vgpr(sh_fp[FP_LO2]); inv2 = vgpr(sh_fp[FP_INV2]); }
  const float out_scale = vgpr(p.out_scale);
  if constexpr (EPI == S_RH_MINMAX || EPI == S_RH_STORE) {
    // block 0 publishes the pulled scalars for the passes after this one
    constexpr int FIN = EPI == S_RH_MINMAX ? (int)ew::FIN_STATS : (int)ew::FIN_BOUNDS2;
    if (blockIdx.x == 0 && threadIdx.x < FP_COUNT && a.fp_w && ew::finalize_writes(FIN, threadIdx.x))
      a.fp_w[threadIdx.x] = sh_fp[threadIdx.x];
  }

  float vmin = __builtin_inff(), vmax = -__builtin_inff();
  Stats2 st; st.init();
  const bool want_rgb = p.ca != 0.f;                 // channel means feed mean3 only then (tonemap.py:119)

  // the lanes holding the image's first / last unit of a row (their pixels 0, 1 / 6, 7 are border pixels)
  const bool is_left = col_ok && c0 == 0, is_right = col_ok && c0 + 8 == p.W;
  const bool any_left = __builtin_amdgcn_ballot_w64(is_left) != 0, any_right = __builtin_amdgcn_ballot_w64(is_right) != 0;

  const bool younger = blockIdx.x * 2 >= gridDim.x;   // (with all blocks resident: the second block of its CU)
  // ---- one row pair: rotation PH of the ring (PH = pair index % 3) ----
  auto body = [&](auto ph_c, int i) {
    constexpr int PH = decltype(ph_c)::value;
    const int r = r_begin + 2 * i;
    // the pair's two new window rows (image rows r + 2, r + 3), loaded three pairs ago
    decode_row(raw[PH][0], lut, lane, win[(2 * PH + 4) % 6]);
    decode_row(raw[PH][1], lut, lane, win[(2 * PH + 5) % 6]);
    load_row(r + 8, raw[PH][0]);
    load_row(r + 9, raw[PH][1]);
    if (r >= r_end) return;                           // wave-uniform: a dead pair of the last rotation
    WinRow w6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) w6[k] = win[(2 * PH + k) % 6];

    static_for<0, 2>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      const int row = r + I;
#if MI_STREAM_PRIO
      // issue-priority turns between the two waves of a SIMD (isp_mega.h, prio_turn): the second half of the grid holds
      // priority in the odd rows, the first half in the even ones.  Only the load kernel: measured per 4K frame, S_STORE
      // 22.3 -> 21.4 us; the passes of the multi-pass chain, two kernels at a time on two streams, got slower with it
      // (66.6 -> 67.8 us per frame) - "second half of the grid" says nothing about who shares a SIMD there.
      if constexpr (EPI == S_STORE) {
        if (younger == (I == 1)) asm volatile("s_setprio 1"); else asm volatile("s_setprio 0");
      }
#endif
      float v[24];
      accumulate_row<PR, PC, I, EXACT>(w6, wq, v);
#ifndef MI_STREAM_ANALYZE   /* reading aid: the hot path alone (scratch compile only) */
      if (row < 2 || row >= p.H - 2) border_fix_rows<PR, PC, I>(v, tile::inside_mask(row, p.H), is_left, is_right);
      else if (any_left || any_right) border_fix_cols<PR, PC, I>(v, is_left, is_right, any_left, any_right);
#endif
#ifndef MI_STREAM_ANALYZE
      if (p.has_ccm) {                                // bayer.py:152-153, sequential fp32 dot
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float x = v[3 * k], y = v[3 * k + 1], z = v[3 * k + 2];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch)
            v[3 * k + ch] = (p.ccm[3 * ch] * x + p.ccm[3 * ch + 1] * y) + p.ccm[3 * ch + 2] * z;
        }
      }
#endif
      if constexpr (BOUNDS) {
        // bounds of the work-dtype image: clamp (bayer.py:155) and rounding to E are monotone, so they are applied
        // once to the reduced min / max (finalize, bounds_post), not to every pixel
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          vmin = fminf(vmin, fminf(v[3 * k], fminf(v[3 * k + 1], v[3 * k + 2])));
          vmax = fmaxf(vmax, fmaxf(v[3 * k], fmaxf(v[3 * k + 1], v[3 * k + 2])));
        }
      }
      // the demosaiced pixel as the reference materialises it: clamped, rounded to the work dtype
      float t[24];
      uint32_t pk[12];
      if constexpr (sizeof(E) == 2) {
#pragma unroll
        for (int j = 0; j < 12; ++j) pk[j] = tile::cvt_pk_f16_clamp01(v[2 * j], v[2 * j + 1]);
        if constexpr (EPI != S_STORE) {
#pragma unroll
          for (int j = 0; j < 12; ++j) {
            half_t h[2];
            __builtin_memcpy(h, &pk[j], 4);
            t[2 * j] = (float)h[0]; t[2 * j + 1] = (float)h[1];
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 24; ++j) t[j] = clamp01(v[j]);
      }
      if constexpr (EPI == S_STORE) {
        if (sub_p && (row & 7) == 0 && col_ok) {         // (row: wave-uniform)
          E* sp = static_cast<E*>(sub_p) + ((size_t)(row >> 3) * a.sub_w + (size_t)(c0 >> 3)) * 3;
          if constexpr (sizeof(E) == 2) {
            uint16_t* s16 = reinterpret_cast<uint16_t*>(sp);
            s16[0] = (uint16_t)(pk[0] & 0xFFFFu); s16[1] = (uint16_t)(pk[0] >> 16); s16[2] = (uint16_t)(pk[1] & 0xFFFFu);
          } else {
            sp[0] = (E)t[0]; sp[1] = (E)t[1]; sp[2] = (E)t[2];
          }
        }
      }
      if constexpr (EPI == S_STORE || EPI == S_STORE_BOUNDS) {
        const uint32_t row_base = (uint32_t)row * out_pitch + band_base;
        if constexpr (sizeof(E) == 2) {
          uint4 mine[3];
          __builtin_memcpy(mine, pk, sizeof(mine));
          wave_store_units<uint4, 3, MI_ST_LOAD>(drsrc, row_base, lane_off, lane, stage, mine);
        } else {
          uint4 mine[6];
          __builtin_memcpy(mine, t, sizeof(mine));
          wave_store_units<uint4, 6, MI_ST_LOAD>(drsrc, row_base, lane_off, lane, stage, mine);
        }
      }
      if constexpr (BOUNDS) {
        // statistics of tonemap.py:147-149 on the assumption lo = 0, hi = 1 (norm01 is then the identity)
        if (want_rgb) {
          st.add8<true>(t);
        } else {
          st.add8<false>(t);
        }
      }
      if constexpr (EPI == S_STATS) {
        float n[24];
#pragma unroll
        for (int j = 0; j < 24; ++j) n[j] = norm01(t[j], lo, inv);
        st.add8<true>(n);
      }
      if constexpr (EPI == S_RH_MINMAX || EPI == S_RH_STORE) {
        auto tone = [&](auto unit_c, auto ca0_c) {
          constexpr bool UNIT = decltype(unit_c)::value, CA0 = decltype(ca0_c)::value;
          float o[24];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float x[3], q[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) x[ch] = UNIT ? t[3 * k + ch] : norm01(t[3 * k + ch], lo, inv);
            reinhard_px<CA0>(x, rk, q);
            if constexpr (EPI == S_RH_MINMAX) {
              vmin = fminf(vmin, fminf(q[0], fminf(q[1], q[2])));
              vmax = fmaxf(vmax, fmaxf(q[0], fmaxf(q[1], q[2])));
            } else {
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) o[3 * k + ch] = q[ch];
            }
          }
          if constexpr (EPI == S_RH_STORE) {
            linear_n<24>(o, lo2, inv2, p.gamma_inv, out_scale);                     // tonemap.py:154
            const uint32_t row_base = (uint32_t)row * out_pitch + band_base;
            switch (p.out_dtype) {
              case MI_U8: wave_store_row_t<uint8_t, ST_STREAM>(drsrc, row_base, lane_off, lane, stage, o); break;
              case MI_U16: wave_store_row_t<uint16_t, ST_STREAM>(drsrc, row_base, lane_off, lane, stage, o); break;
              case MI_F16: {                                // pairs leave through v_cvt_pk_f16_f32 (half the conversions)
                uint32_t pk2[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk2[j]) : "v"(o[2 * j]), "v"(o[2 * j + 1]));
                uint4 mine[3];
                __builtin_memcpy(mine, pk2, sizeof(mine));
                wave_store_units<uint4, 3, ST_STREAM>(drsrc, row_base, lane_off, lane, stage, mine);
                break;
              }
              default: wave_store_row_t<float, ST_STREAM>(drsrc, row_base, lane_off, lane, stage, o); break;
            }
          }
        };
        if (ca0) {
          if (unit) tone(std::true_type{}, std::true_type{});
          else tone(std::false_type{}, std::true_type{});
        } else {
          tone(std::false_type{}, std::false_type{});
        }
      }
    });
  };

  const int n_pairs = (r_end - r_begin) / 2;
  for (int i = 0; i < n_pairs; i += 3) {
    body(std::integral_constant<int, 0>{}, i);
    MI_SSTAMP(3 + (i < 9 ? i : 9));
    body(std::integral_constant<int, 1>{}, i + 1);
    MI_SSTAMP(4 + (i < 9 ? i : 9));
    body(std::integral_constant<int, 2>{}, i + 2);
    MI_SSTAMP(5 + (i < 9 ? i : 9));
  }
#if MI_STREAM_PRIO
  if constexpr (EPI == S_STORE) asm volatile("s_setprio 0");
#endif
  MI_SSTAMP(15);

  // ---- reductions: one partial per block ----
  st.finish();
  if (!col_ok) { vmin = __builtin_inff(); vmax = -__builtin_inff(); st.init(); }   // lanes beyond the image saw zeros
  if constexpr (BOUNDS) {
    const float v9[9] = {vmin, vmax, st.gmin, st.gmax, st.slog, st.sgray, st.s0, st.s1, st.s2};
    const int op[9] = {0, 1, 0, 1, 2, 2, 2, 2, 2};
    block_reduce_store_nb<9>(v9, op, red, &arrived, p.partials + (size_t)ROW_BOUNDS * p.part_stride, p.part_stride,
                             blockIdx.x, wave, lane);
  } else if constexpr (EPI == S_STATS) {
    const float v7[7] = {st.gmin, st.gmax, st.slog, st.sgray, st.s0, st.s1, st.s2};
    const int op[7] = {0, 1, 2, 2, 2, 2, 2};
    block_reduce_store_nb<7>(v7, op, red, &arrived, p.partials + (size_t)ROW_STATS * p.part_stride, p.part_stride,
                             blockIdx.x, wave, lane);
  } else if constexpr (EPI == S_RH_MINMAX) {
    const float v2[2] = {vmin, vmax};
    const int op[2] = {0, 1};
    block_reduce_store_nb<2>(v2, op, red, &arrived, p.partials + (size_t)ROW_BOUNDS2 * p.part_stride, p.part_stride,
                             blockIdx.x, wave, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// The stride-8 subsample ISP.update_metering reads (camera_isp.py:168-170: image[::8, ::8]) straight from the PACKED
// frame, without the image: only the rows r % 8 == 0 are demosaiced - five packed rows in, one sample per unit out.
// For callers that want a camera group's metering BEFORE its images exist (mi_isp_camera_group_reinhard: the Reinhard
// scalars come from the metering, so a kernel that tone-maps while it demosaics needs them first).  The same decode
// table, window, accumulate_row, border fixes, colour matrix, clamp and rounding as stream_kernel<S_STORE>: the samples
// are the bits its epilogue writes (tests/: compared with the subsample of the loaded image).  A wave = one sample row
// of one band; 1 / 8 of the demosaic and 5 / 8 of the decode of a full pass.
struct SubArgs {
  Params t;
  int bands_x, n_waves, n_blocks;
  int sub_w;              // ceil(W / 8)
  int n_batch;            // >= 1: grid.y = frame; frame y reads srcs[y], writes subs[y]
  const void* srcs[LOAD_BATCH];
  void* subs[LOAD_BATCH];
};

template <class E, int PR, int PC>
__global__ __launch_bounds__(THREADS) void sub_kernel(const SubArgs a) {
  constexpr bool EXACT = sizeof(E) == 2;
  const Params& p = a.t;
  __shared__ float lut[4096];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = blockIdx.x * WAVES + wave;
  const bool wave_ok = g < a.n_waves;
  const int by = g / a.bands_x, bx = g - by * a.bands_x;
  const int c0 = bx * BAND + lane * 8;
  const int row = by * 8;                              // the sample row (even: strip row 0 of its row pair)
  const bool col_ok = wave_ok && c0 < p.W;
  const uint32_t pitch = (uint32_t)p.W * 3 / 2;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.srcs[blockIdx.y]), 0,
                                                                        (int)((uint32_t)p.H * pitch), 0x00020000);
  const uint32_t col_off = col_ok ? (uint32_t)c0 * 3 / 2 : INVALID_OFF;
  const bool ext_ok = col_ok && ((lane == 0 && c0 > 0) || (lane == 63 && c0 + 8 < p.W));
  const uint32_t ext_off = ext_ok ? (uint32_t)c0 * 3 / 2 + (lane == 0 ? -4 : 12) : INVALID_OFF;
  uint32_t raw[5][4];
#pragma unroll
  for (int q = 0; q < 5; ++q) {                        // rows row - 2 .. row + 2; outside the image: zeros (as stream_kernel)
    const int r = row - 2 + q;
    const uint32_t row_off = (wave_ok && r >= 0 && r < p.H) ? (uint32_t)r * pitch : INVALID_OFF;
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    const u3 d = __builtin_amdgcn_raw_buffer_load_b96(rsrc, col_off + row_off, 0, 0);
    raw[q][0] = d.x; raw[q][1] = d.y; raw[q][2] = d.z;
    raw[q][3] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, ext_off + row_off, 0, 0);
  }
  for (int e = threadIdx.x; e < 4096; e += THREADS) lut[e] = tile::decode_scaled<E>((uint32_t)e, p.k_decode);
  __syncthreads();
  float wq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wq[i] = vgpr(wq_value(i));
  WinRow w6[6];
#pragma unroll
  for (int q = 0; q < 5; ++q) decode_row(raw[q], lut, lane, w6[q]);
#pragma unroll
  for (int j = 0; j < 12; ++j) w6[5].v[j] = 0.f;         // (strip row 0 does not look at the window's sixth row)
  const bool is_left = col_ok && c0 == 0, is_right = col_ok && c0 + 8 == p.W;
  const bool any_left = __builtin_amdgcn_ballot_w64(is_left) != 0, any_right = __builtin_amdgcn_ballot_w64(is_right) != 0;
  float v[24];
  accumulate_row<PR, PC, 0, EXACT>(w6, wq, v);
  if (row < 2 || row >= p.H - 2) border_fix_rows<PR, PC, 0>(v, tile::inside_mask(row, p.H), is_left, is_right);
  else if (any_left || any_right) border_fix_cols<PR, PC, 0>(v, is_left, is_right, any_left, any_right);
  float x = v[0], y = v[1], z = v[2];                  // the lane's first pixel = column c0 = a multiple of 8
  if (p.has_ccm) {                                     // bayer.py:152-153, sequential fp32 dot
    float o[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) o[ch] = (p.ccm[3 * ch] * x + p.ccm[3 * ch + 1] * y) + p.ccm[3 * ch + 2] * z;
    x = o[0]; y = o[1]; z = o[2];
  }
  if (!col_ok || row >= p.H) return;
  E* sp = static_cast<E*>(a.subs[blockIdx.y]) + ((size_t)by * a.sub_w + (size_t)(c0 >> 3)) * 3;
  if constexpr (sizeof(E) == 2) {
    const uint32_t p0 = tile::cvt_pk_f16_clamp01(x, y), p1 = tile::cvt_pk_f16_clamp01(z, 0.f);
    uint16_t* s16 = reinterpret_cast<uint16_t*>(sp);
    s16[0] = (uint16_t)(p0 & 0xFFFFu); s16[1] = (uint16_t)(p0 >> 16); s16[2] = (uint16_t)(p1 & 0xFFFFu);
  } else {
    sp[0] = (E)clamp01(x); sp[1] = (E)clamp01(y); sp[2] = (E)clamp01(z);
  }
}

static inline void sub_geometry(int H, int W, SubArgs& a) {
  a.bands_x = (W + BAND - 1) / BAND;
  a.sub_w = (W + 7) / 8;
  a.n_waves = a.bands_x * ((H + 7) / 8);
  a.n_blocks = (a.n_waves + WAVES - 1) / WAVES;
}

// ---- host side ---------------------------------------------------------------------------------
// geometry: bands of 512 columns; rows per wave chosen so that the grid holds about 2 waves per SIMD
static inline void geometry(int H, int W, SArgs& a) {
  a.bands_x = (W + BAND - 1) / BAND;
  int target_waves = 2048;
#ifdef MI_ISP_MEASURE
  static const int env_waves = getenv("MI_ISP_STREAM_WAVES") ? atoi(getenv("MI_ISP_STREAM_WAVES")) : 0;
  if (env_waves > 0) target_waves = env_waves;
#endif
  int rpw = (int)(((long)H * a.bands_x + target_waves - 1) / target_waves);
  rpw = (rpw + 1) / 2 * 2;                            // whole row pairs (a last, partial rotation of the ring skips its dead pairs)
  if (rpw < 4) rpw = 4;
  a.rows_per_wave = rpw;
  const int bands_y = (H + rpw - 1) / rpw;
  a.n_waves = a.bands_x * bands_y;
  a.n_blocks = (a.n_waves + WAVES - 1) / WAVES;
}

// what the stream kernels handle: 12-bit packed sources in the standard layout with 4-byte aligned rows, whole
// 8-pixel units, unit scale, frames below 1 GiB (the rest stays with tile::tile_kernel)
static inline bool supported(const Params& p, int work_dtype) {
  // H >= 4: the border rows then only have the masks BORDER_DY tabulates
  return p.src_kind == tile::SRC_PACKED12 && p.src_fast && p.W % 8 == 0 && p.H % 2 == 0 && p.H >= 4 && p.in_scale == 1.f &&
         (work_dtype == MI_F16 || work_dtype == MI_F32) && (int64_t)p.H * p.W * 3 / 2 < (int64_t)0x40000000;
}

int launch_sub_rggb(const SubArgs& a, int work_dtype, hipStream_t stream);
int launch_sub_grbg(const SubArgs& a, int work_dtype, hipStream_t stream);
int launch_sub_gbrg(const SubArgs& a, int work_dtype, hipStream_t stream);
int launch_sub_bggr(const SubArgs& a, int work_dtype, hipStream_t stream);
static inline int launch_sub(const SubArgs& a, int work_dtype, int pattern, hipStream_t stream) {
  switch (pattern) {
    case MI_RGGB: return launch_sub_rggb(a, work_dtype, stream);
    case MI_GRBG: return launch_sub_grbg(a, work_dtype, stream);
    case MI_GBRG: return launch_sub_gbrg(a, work_dtype, stream);
    default: return launch_sub_bggr(a, work_dtype, stream);
  }
}
int launch_rggb(const SArgs& a, int work_dtype, int epi, hipStream_t stream);
int launch_grbg(const SArgs& a, int work_dtype, int epi, hipStream_t stream);
int launch_gbrg(const SArgs& a, int work_dtype, int epi, hipStream_t stream);
int launch_bggr(const SArgs& a, int work_dtype, int epi, hipStream_t stream);
static inline int launch(const SArgs& a, int work_dtype, int pattern, int epi, hipStream_t stream) {
  switch (pattern) {
    case MI_RGGB: return launch_rggb(a, work_dtype, epi, stream);
    case MI_GRBG: return launch_grbg(a, work_dtype, epi, stream);
    case MI_GBRG: return launch_gbrg(a, work_dtype, epi, stream);
    default: return launch_bggr(a, work_dtype, epi, stream);
  }
}

}  // namespace strm
