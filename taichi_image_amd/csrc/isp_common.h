// Shared host/device helpers for libmi355_isp.so (gfx950 only, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mi_isp.h"

typedef _Float16 half_t;

// ---- host-side error handling ---------------------------------------------------------------
void mi_set_error(const char* fmt, ...);   // defined in isp_api.hip (thread-local buffer)

#define MI_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      mi_set_error(__VA_ARGS__);         \
      return 1;                          \
    }                                    \
  } while (0)

#define MI_HIP(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      mi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                                   \
    }                                                                             \
  } while (0)

#define MI_LAUNCH_CHECK() MI_HIP(hipGetLastError())

static inline bool mi_valid_dtype(int d) { return d >= MI_U8 && d <= MI_F32; }
static inline size_t mi_dtype_size(int d) { return d == MI_U8 ? 1 : (d == MI_F32 ? 4 : 2); }
static inline float mi_scale_factor(int d) { return d == MI_U8 ? 255.f : (d == MI_U16 ? 65535.f : 1.f); }
__host__ __device__ static inline size_t mi_dtype_size_dev(int d) { return d == MI_U8 ? 1 : (d == MI_F32 ? 4 : 2); }
static inline bool mi_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ---- device helpers ---------------------------------------------------------------------------
#define MI_DEV __device__ __forceinline__

// ti.cast(f32 -> T): RNE for floats, truncation for unsigned ints.  NaN -> 0 and saturation are
// the library's definition where the reference is undefined (fptoui poison).
template <class T> MI_DEV T cast_out(float x);
template <> MI_DEV float cast_out<float>(float x) { return x; }
// f32 -> f16 is a separate RNE conversion of the already-rounded f32 value (what ti.cast of an f32
// expression does on every Taichi backend).  The empty asm keeps hipcc from folding a preceding
// multiply into v_fma_mixlo_f16, which rounds the exact product ONCE to f16 and so differs from
// the two-step result for about 1 in 2^13 values (measured on gfx950).
MI_DEV float f32_rounded(float x) { asm("" : "+v"(x)); return x; }
template <> MI_DEV half_t cast_out<half_t>(float x) { return (half_t)f32_rounded(x); }
template <> MI_DEV uint8_t cast_out<uint8_t>(float x) {
  x = fminf(fmaxf(x, 0.f), 255.f);  // fmaxf(NaN, 0) == 0
  return (uint8_t)(unsigned)x;
}
template <> MI_DEV uint16_t cast_out<uint16_t>(float x) {
  x = fminf(fmaxf(x, 0.f), 65535.f);
  return (uint16_t)(unsigned)x;
}

template <class T> MI_DEV float to_f32(T v) { return (float)v; }

template <class T> struct ScaleOf { static constexpr float value = 1.f; };
template <> struct ScaleOf<uint8_t> { static constexpr float value = 255.f; };
template <> struct ScaleOf<uint16_t> { static constexpr float value = 65535.f; };

// x / scale_factor(T) for an integer-valued x of an integer type T (0 <= x <= scale), correctly rounded
// like the IEEE division it replaces: q = x * r with r = RN(1 / scale), one FMA residual correction
// (3 instructions instead of the ~12 of v_div_scale / v_rcp / v_div_fmas / v_div_fixup).  Exhaustively
// equal to x / scale for every u8 and u16 value (tests/test_oracle.py::test_scaled_division_trick).
template <class T> MI_DEV float div_scale(float x) {
  constexpr float d = ScaleOf<T>::value;
  if constexpr (d == 1.f) {
    return x;
  } else {
    constexpr float r = 1.0f / d;
    const float q = x * r;
    const float e = __builtin_fmaf(-q, d, x);
    return __builtin_fmaf(e, r, q);
  }
}

// NaN-ignoring min/max (v_min_f32 / v_max_f32 semantics, == llvm.minnum/maxnum)
MI_DEV float nmin(float a, float b) { return fminf(a, b); }
MI_DEV float nmax(float a, float b) { return fmaxf(a, b); }

// Wave64 reductions in registers: four DPP steps fold each row of 16 lanes (every lane of a row ends
// up with the row's result), row_bcast15 / row_bcast31 fold the four rows into lane 63, and a
// v_readlane hands the result to every lane as a scalar.  No LDS traffic: ds_bpermute-based shuffles
// cost a dependent LDS round trip per step (12-42 of them per block here).
template <int CTRL, int ROW_MASK = 0xF> MI_DEV float dpp_mov(float v, float old) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                CTRL, ROW_MASK, 0xF, false));
}
// quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140,
// row_bcast15 = 0x142 (rows 1 and 3 take lane 15 of the row before), row_bcast31 = 0x143 (rows 2, 3 take lane 31).
// Lanes outside the row mask get IDENT, the identity of OP.
#define MI_WAVE_REDUCE(OP, IDENT)                          \
  v = OP(v, dpp_mov<0xB1>(v, v));                          \
  v = OP(v, dpp_mov<0x4E>(v, v));                          \
  v = OP(v, dpp_mov<0x141>(v, v));                         \
  v = OP(v, dpp_mov<0x140>(v, v));                         \
  v = OP(v, dpp_mov<0x142, 0xA>(v, IDENT));                \
  v = OP(v, dpp_mov<0x143, 0xC>(v, IDENT));                \
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
MI_DEV float op_add(float a, float b) { return a + b; }
MI_DEV float wave_min(float v) { MI_WAVE_REDUCE(fminf, __builtin_inff()) }
MI_DEV float wave_max(float v) { MI_WAVE_REDUCE(fmaxf, -__builtin_inff()) }
MI_DEV float wave_sum(float v) { MI_WAVE_REDUCE(op_add, 0.f) }
#undef MI_WAVE_REDUCE
// fp64 sum for the pulled finalize (rgb_pass_kernel prologue): the halves ride on two DPP moves
template <int CTRL, int ROW_MASK = 0xF> MI_DEV double dpp_mov(double v, double old) {
  const long long b = __builtin_bit_cast(long long, v), o = __builtin_bit_cast(long long, old);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)b, CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
MI_DEV double wave_sum(double v) {
  v += dpp_mov<0xB1>(v, v);
  v += dpp_mov<0x4E>(v, v);
  v += dpp_mov<0x141>(v, v);
  v += dpp_mov<0x140>(v, v);
  v += dpp_mov<0x142, 0xA>(v, 0.0);
  v += dpp_mov<0x143, 0xC>(v, 0.0);
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// Block reduction of up to RW values per thread for 256-thread blocks (4 waves): each wave
// shuffles down to one value, the per-wave results meet in LDS, thread k < NV combines them
// and stores partial[k * stride + block].  op[k]: 0 = min, 1 = max, 2 = sum.
template <int NV, int RW>
MI_DEV void block_reduce_store(const float (&v)[NV], const int (&op)[NV], float (*red)[RW],
                               float* partials, int stride, int block) {
  static_assert(NV <= RW, "staging row too narrow");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float r = op[k] == 0 ? wave_min(v[k]) : (op[k] == 1 ? wave_max(v[k]) : wave_sum(v[k]));
    if (lane == 0) red[wave][k] = r;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int k = threadIdx.x;
    float r = red[0][k];
    const int nw = blockDim.x >> 6;
    for (int w = 1; w < nw; ++w) {
      float o = red[w][k];
      r = op[k] == 0 ? fminf(r, o) : (op[k] == 1 ? fmaxf(r, o) : r + o);
    }
    partials[(size_t)k * stride + block] = r;
  }
}

// ---- wave-cooperative 24-element-per-lane IO ----------------------------------------------------------
// Interleaved RGB puts 8 pixels = 24 elements = 48 bytes (f16) in each lane.  Accessed directly that
// is three 16-byte accesses per lane at a 48-byte lane stride, which the memory pipeline serves at
// 3.7 TB/s (measured, scratch/store_bench.hip) against 6.2 TB/s for accesses that are contiguous
// across the wave.  These helpers move a wave's 64 x 24 elements between global memory and
// registers with wave-contiguous accesses, transposing through a per-wave LDS buffer
// (no block barrier: LDS operations of one wave execute in order).
template <class T> struct IoUnit { typedef uint4 type; };
template <> struct IoUnit<uint8_t> { typedef uint2 type; };
template <class T> struct IoUnits { static constexpr int value = (int)(sizeof(T) * 24 / sizeof(typename IoUnit<T>::type)); };
constexpr int WAVE_IO_BYTES = 64 * 24 * 4;          // per-wave LDS staging, sized for 4-byte elements

template <class T>
MI_DEV void wave_load24(const T* gptr, int lane, void* lbuf_, T (&t)[24]) {
  typedef typename IoUnit<T>::type U;
  constexpr int N = IoUnits<T>::value;
  U* lbuf = static_cast<U*>(lbuf_);
  const U* g = reinterpret_cast<const U*>(gptr);
#pragma unroll
  for (int j = 0; j < N; ++j) lbuf[j * 64 + lane] = g[j * 64 + lane];
  __builtin_amdgcn_wave_barrier();
  U mine[N];
#pragma unroll
  for (int j = 0; j < N; ++j) mine[j] = lbuf[lane * N + j];
  __builtin_memcpy(t, mine, sizeof(mine));
  __builtin_amdgcn_wave_barrier();
}

// NT: stream the stores to memory (final outputs nobody on the chip reads back; see strm::ST_STREAM)
template <bool NT, class U> MI_DEV void store_unit(U* p, const U& v) {
  if constexpr (NT) {                                  // (the builtin takes clang vector types, not HIP's uint2 / uint4)
    typedef uint32_t vec_t __attribute__((ext_vector_type(sizeof(U) / 4)));
    vec_t x;
    __builtin_memcpy(&x, &v, sizeof(U));
    __builtin_nontemporal_store(x, reinterpret_cast<vec_t*>(p));
  } else {
    *p = v;
  }
}
template <class T, bool NT = false>
MI_DEV void wave_store24(T* gptr, int lane, void* lbuf_, const T (&t)[24]) {
  typedef typename IoUnit<T>::type U;
  constexpr int N = IoUnits<T>::value;
  U* lbuf = static_cast<U*>(lbuf_);
  U mine[N];
  __builtin_memcpy(mine, t, sizeof(mine));
#pragma unroll
  for (int j = 0; j < N; ++j) lbuf[lane * N + j] = mine[j];
  __builtin_amdgcn_wave_barrier();
  if constexpr (sizeof(T) == 1) {
    // 1-byte elements: the wave's 1536 bytes leave as 96 16-byte units (lanes 0..63, then 0..31) instead of 192
    // 8-byte ones when the destination allows it (wave-uniform test)
    if ((reinterpret_cast<uintptr_t>(gptr) & 15) == 0) {
      const uint4* l4 = reinterpret_cast<const uint4*>(lbuf_);
      uint4* g4 = reinterpret_cast<uint4*>(gptr);
      store_unit<NT>(g4 + lane, l4[lane]);
      if (lane < 32) store_unit<NT>(g4 + 64 + lane, l4[64 + lane]);
      __builtin_amdgcn_wave_barrier();
      return;
    }
  }
  U* g = reinterpret_cast<U*>(gptr);
#pragma unroll
  for (int j = 0; j < N; ++j) store_unit<NT>(g + j * 64 + lane, lbuf[j * 64 + lane]);
  __builtin_amdgcn_wave_barrier();
}

// ---- per-frame workspace layout (floats) ---------------------------------------------------------
// [0, 64)            : FrameParams -- scalars produced by the finalize steps (finalize_kernel, or pulled into the
//                      consuming pass's prologue), read by later passes
// [64, 64 + 8*cap)   : per-block partials, SoA: partial[k][block]
enum {
  FP_LO = 0, FP_HI = 1, FP_INV = 2,                 // bounds of the input image, 1/(hi-lo)
  FP_BMIN = 3, FP_BMAX = 4, FP_LMEAN = 5, FP_GMEAN = 6, FP_RMEAN = 7, /* 8, 9 */
  FP_MAPKEY = 10, FP_EI = 11, FP_MEAN3 = 12, /* 13, 14 */
  FP_LO2 = 15, FP_HI2 = 16, FP_INV2 = 17,           // bounds of the Reinhard image
  FP_MAXOUT = 18,                                   // ISP reinhard: max(1e-6, max p)
  FP_COUNT = 64
};
static inline int mi_partial_cap(int H, int W) {
  // enough blocks for the tile kernels (128x32 px tiles) and the elementwise reductions
  long tiles = (long)((W + 127) / 128) * ((H + 31) / 32);
  long cap = tiles < 4096 ? 4096 : tiles;
  return (int)cap;
}
