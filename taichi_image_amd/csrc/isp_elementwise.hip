// Elementwise / gather kernels of the camera-ISP path and the small finalize kernels that turn
// per-block partials into the scalars of the next pass.  All of these are HBM- (or launch-)
// bound byte movers: wide coalesced accesses, no LDS tiling, wave-shuffle reductions.
#include "isp_elementwise.h"
#include "isp_math.h"

#include <type_traits>
#include <atomic>
#include <mutex>
#include <cstdlib>

#pragma clang fp contract(off)

namespace {

constexpr int EW_THREADS = 256;
// RGB passes: few fat blocks (2 per CU), so that the per-block prologue that folds the previous
// pass's partials (pull_finalize) reads little in total and leaves few partials itself
constexpr int PASS_THREADS = 512;
constexpr int PASS_MAX_BLOCKS = 1024;              // capacity of the partial rows; pass_blocks() picks the grid

static inline int grid_for(int64_t work_items, int cap = 256 * 16) {
  int64_t b = (work_items + EW_THREADS - 1) / EW_THREADS;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

template <class F> static int dispatch_dtype(int dtype, F&& f) {
  switch (dtype) {
    case MI_U8: return f((uint8_t)0);
    case MI_U16: return f((uint16_t)0);
    case MI_F16: return f((half_t)0);
    default: return f((float)0);
  }
}

// unpack helpers shared with the tile kernels ------------------------------------------------
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

template <class T> MI_DEV T write_value(uint32_t v, bool scaled, float k) {
  // packed.py:98-104: scaled -> cast(f32(v) * k); direct -> numeric conversion
  return scaled ? cast_out<T>((float)v * k) : cast_out<T>((float)v);
}

// ---------------------------------------------------------------------------------------------
// K1 decode12 (packed.py:92-131).  Fast path: 8 px per lane from three dwords.
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(EW_THREADS) void decode12_kernel(const uint8_t* __restrict__ enc,
                                                              T* __restrict__ out, int64_t n_pairs,
                                                              int scaled, int ids, float k, int fast) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_groups = fast ? n_pairs / 4 : 0;
  for (int64_t g = tid; g < n_groups; g += stride) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(enc + g * 12);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    const uint32_t w[4] = {d0 & 0xFFFFFFu, (d0 >> 24) | ((d1 & 0xFFFFu) << 8),
                           (d1 >> 16) | ((d2 & 0xFFu) << 16), d2 >> 8};
    T o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t p0, p1;
      unpack_pair(w[j], ids, p0, p1);
      o[2 * j] = write_value<T>(p0, scaled, k);
      o[2 * j + 1] = write_value<T>(p1, scaled, k);
    }
    T* dst = out + g * 8;
    if (sizeof(T) == 1) {
      *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(o);
    } else if (sizeof(T) == 2) {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
    } else {
      reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(o)[0];
      reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(o)[1];
    }
  }
  for (int64_t j = n_groups * 4 + tid; j < n_pairs; j += stride) {
    const uint8_t* q = enc + j * 3;
    const uint32_t w = q[0] | (q[1] << 8) | (q[2] << 16);
    uint32_t p0, p1;
    unpack_pair(w, ids, p0, p1);
    out[2 * j] = write_value<T>(p0, scaled, k);
    out[2 * j + 1] = write_value<T>(p1, scaled, k);
  }
}

// K2 decode16 (packed.py:135-172)
template <class T>
__global__ __launch_bounds__(EW_THREADS) void decode16_kernel(const uint8_t* __restrict__ enc,
                                                              T* __restrict__ out, int64_t n,
                                                              int scaled, float k, int fast) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_groups = fast ? n / 8 : 0;
  for (int64_t g = tid; g < n_groups; g += stride) {
    const uint4 d = *reinterpret_cast<const uint4*>(enc + g * 16);
    const uint32_t v[8] = {d.x & 0xFFFFu, d.x >> 16, d.y & 0xFFFFu, d.y >> 16,
                           d.z & 0xFFFFu, d.z >> 16, d.w & 0xFFFFu, d.w >> 16};
    T o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = write_value<T>(v[j], scaled, k);
    T* dst = out + g * 8;
    if (sizeof(T) == 1) {
      *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(o);
    } else if (sizeof(T) == 2) {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
    } else {
      reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(o)[0];
      reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(o)[1];
    }
  }
  for (int64_t i = n_groups * 8 + tid; i < n; i += stride) {
    const uint32_t v = enc[2 * i] | (enc[2 * i + 1] << 8);
    out[i] = write_value<T>(v, scaled, k);
  }
}

// K3 encode12 (packed.py:60-89) -- test-input generator, one pair per lane
template <class T>
__global__ __launch_bounds__(EW_THREADS) void encode12_kernel(const T* __restrict__ values,
                                                              uint8_t* __restrict__ enc,
                                                              int64_t n_pairs, int scaled, int ids,
                                                              float k) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_pairs; j += stride) {
    uint32_t p[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float x = (float)values[2 * j + e];
      if (scaled) {
        const float r = roundf(x * k);  // ti.round: half away from zero
        p[e] = (uint32_t)fminf(fmaxf(r, 0.f), 65535.f);
      } else {
        p[e] = (uint32_t)values[2 * j + e] & 0xFFFFu;
      }
    }
    uint8_t* q = enc + 3 * j;
    if (!ids) {  // packed.py:13-20
      q[0] = p[0] & 0xFF;
      q[1] = ((p[1] & 0xF) << 4) | ((p[0] >> 8) & 0xFF);
      q[2] = (p[1] >> 4) & 0xFF;
    } else {     // packed.py:48-55
      q[0] = (p[0] >> 4) & 0xFF;
      q[1] = (p[1] >> 4) & 0xFF;
      q[2] = ((p[0] & 0xF) << 4) | (p[1] & 0xF);
    }
  }
}

// K11 loaders (camera_isp.py:82-99)
template <class T>
__global__ __launch_bounds__(EW_THREADS) void load_convert_kernel(const void* __restrict__ src,
                                                                  T* __restrict__ dst, int64_t n,
                                                                  int mode) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float x;
    if (mode == MI_LOAD_16U) x = (float)static_cast<const uint16_t*>(src)[i] / 65535.0f;
    else if (mode == MI_LOAD_32F) x = static_cast<const float*>(src)[i];
    else x = (float)static_cast<const uint16_t*>(src)[i];
    dst[i] = cast_out<T>(x);
  }
}

// K5 rgb_to_bayer (bayer.py:101-112): channel index per site from pixel_orders (bayer.py:85-90)
template <class T>
__global__ __launch_bounds__(EW_THREADS) void mosaic_kernel(const T* __restrict__ rgb,
                                                            T* __restrict__ cfa, int H, int W,
                                                            int order4) {
  const int64_t n = (int64_t)H * W;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int r = (int)(i / W), c = (int)(i - (int64_t)r * W);
    const int site = (r & 1) * 2 + (c & 1);          // (r0c0, r0c1, r1c0, r1c1)
    const int ch = (order4 >> (2 * site)) & 3;
    cfa[i] = rgb[i * 3 + ch];
  }
}

// K6 bilinear (interpolate.py:19-34,59-66)
template <class TI, class TO>
__global__ __launch_bounds__(EW_THREADS) void resize_kernel(const TI* __restrict__ src,
                                                            TO* __restrict__ dst, int Hs, int Ws,
                                                            int Hd, int Wd, float s0, float s1,
                                                            float intensity) {
  const int64_t n = (int64_t)Hd * Wd;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int r = (int)(i / Wd), c = (int)(i - (int64_t)r * Wd);
    const float pr = (float)r / s0, pc = (float)c / s1;
    const int ir = (int)pr, ic = (int)pc;
    const float fr = pr - (float)ir, fc = pc - (float)ic;
    const int r0 = min(max(ir, 0), Hs - 1), r1 = min(max(ir + 1, 0), Hs - 1);
    const int c0 = min(max(ic, 0), Ws - 1), c1 = min(max(ic + 1, 0), Ws - 1);
    const TI* a = src + ((size_t)r0 * Ws + c0) * 3;
    const TI* b = src + ((size_t)r1 * Ws + c0) * 3;
    const TI* cc = src + ((size_t)r0 * Ws + c1) * 3;
    const TI* d = src + ((size_t)r1 * Ws + c1) * 3;
    TO* o = dst + i * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float y1 = (float)a[ch] * (1.0f - fr) + (float)b[ch] * fr;
      const float y2 = (float)cc[ch] * (1.0f - fr) + (float)d[ch] * fr;
      o[ch] = cast_out<TO>((y1 * (1.0f - fc) + y2 * fc) * intensity);
    }
  }
}

// interpolate.py:36-54: source index for destination (r, c), shape = destination shape
MI_DEV void transformed(int Hd, int Wd, int r, int c, int t, int& sr, int& sc) {
  switch (t) {
    case MI_T_ROTATE_90: sr = Wd - c - 1; sc = r; break;
    case MI_T_ROTATE_180: sr = Hd - r - 1; sc = Wd - c - 1; break;
    case MI_T_ROTATE_270: sr = c; sc = Hd - r - 1; break;
    case MI_T_TRANSPOSE: sr = c; sc = r; break;
    case MI_T_FLIP_VERT: sr = Hd - r - 1; sc = c; break;
    case MI_T_FLIP_HORIZ: sr = r; sc = Wd - c - 1; break;
    case MI_T_TRANSVERSE: sr = Wd - c - 1; sc = Hd - r - 1; break;
    default: sr = r; sc = c; break;
  }
}

// K7 transform (interpolate.py:93-125)
template <class T>
__global__ __launch_bounds__(EW_THREADS) void transform_kernel(const T* __restrict__ src,
                                                               T* __restrict__ dst, int Hs, int Ws,
                                                               int Hd, int Wd, int t) {
  const int64_t n = (int64_t)Hd * Wd;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int r = (int)(i / Wd), c = (int)(i - (int64_t)r * Wd);
    int sr, sc;
    transformed(Hd, Wd, r, c, t, sr, sc);
    const T* s = src + ((size_t)sr * Ws + sc) * 3;
    T* o = dst + i * 3;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
  }
}

// ---------------------------------------------------------------------------------------------
// color/yuv_420.py: RGB <-> planar YUV 4:2:0 (the step after the path).  One thread per
// 2x2 block.  The reference's quirks are reproduced: the BGR-named matrix is applied to rgb.bgr
// (Y = 0.299 B + 0.587 G + 0.114 R, yuv_420.py:12-27) and tm.clamp(0, 1, x) is min(1, x) (:54-57,88).
// Layout (yuv_420.py:95-103): yuv is (H * 3 / 2, W); rows [0, H) = Y, the rest = two (H/2, W/2) planes,
// plane 0 = yuv.z, plane 1 = yuv.y (:58-59).
// ---------------------------------------------------------------------------------------------
MI_DEV void ycrcb_from_rgb(float r, float g, float b, float (&yuv)[3]) {
  // YCrCb_T_bgr @ (b, g, r) + (0, 0.5, 0.5), rows as (m0 * v0 + m1 * v1) + m2 * v2 in fp32
  yuv[0] = (0.299f * b + 0.587f * g) + 0.114f * r;
  yuv[1] = ((-0.168736f * b + -0.331264f * g) + 0.5f * r) + 0.5f;
  yuv[2] = ((0.5f * b + -0.418688f * g) + -0.081312f * r) + 0.5f;
}

template <class TI, class TO>
__global__ __launch_bounds__(EW_THREADS) void rgb_yuv420_kernel(const TI* __restrict__ src, TO* __restrict__ yuv,
                                                                int H, int W) {
  const int hb = H / 2, wb = W / 2;
  const float in_scale = ScaleOf<TI>::value, out_scale = ScaleOf<TO>::value;
  TO* yp = yuv;
  TO* plane0 = yuv + (size_t)H * W;
  TO* plane1 = plane0 + (size_t)hb * wb;
  const int64_t n = (int64_t)hb * wb, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int br = (int)(i / wb), bc = (int)(i - (int64_t)br * wb);
    float su = 0.f, sv = 0.f;
#pragma unroll
    for (int dr = 0; dr < 2; ++dr)
#pragma unroll
      for (int dc = 0; dc < 2; ++dc) {                 // ti.ndrange(2, 2) order (yuv_420.py:52)
        const size_t px = (size_t)(2 * br + dr) * W + (2 * bc + dc);
        const TI* p = src + px * 3;
        float t[3];
        ycrcb_from_rgb(div_scale<TI>((float)p[0]), div_scale<TI>((float)p[1]), div_scale<TI>((float)p[2]), t);
        yp[px] = cast_out<TO>(fminf(1.0f, t[0]) * out_scale);
        su = su + t[1]; sv = sv + t[2];
      }
    plane1[i] = cast_out<TO>(fminf(1.0f, su / 4.0f) * out_scale);
    plane0[i] = cast_out<TO>(fminf(1.0f, sv / 4.0f) * out_scale);
  }
}

template <class TI, class TO>
__global__ __launch_bounds__(EW_THREADS) void yuv420_rgb_kernel(const TI* __restrict__ yuv, TO* __restrict__ rgb,
                                                                int H, int W) {
  const int hb = H / 2, wb = W / 2;
  const float in_scale = ScaleOf<TI>::value, out_scale = ScaleOf<TO>::value;
  const TI* yp = yuv;
  const TI* plane0 = yuv + (size_t)H * W;
  const TI* plane1 = plane0 + (size_t)hb * wb;
  // bgr_T_YCrCb = inverse(YCrCb_T_bgr) evaluated in float64 (yuv_420.py:18), rounded to fp32
  const float m[9] = {1.0f, -1.2188942e-06f, 1.4019996f, 1.0f, -0.34413567f, -0.71413618f,
                      1.0f, 1.7720001f, 4.0629806e-07f};
  const int64_t n = (int64_t)H * W, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int r = (int)(i / W), c = (int)(i - (int64_t)r * W);
    const size_t ib = (size_t)(r / 2) * wb + (c / 2);
    const float y = div_scale<TI>((float)yp[i]);
    const float u = div_scale<TI>((float)plane1[ib]) - 0.5f, v = div_scale<TI>((float)plane0[ib]) - 0.5f;
    float bgr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) bgr[k] = (m[3 * k] * y + m[3 * k + 1] * u) + m[3 * k + 2] * v;
    TO* q = rgb + (size_t)i * 3;
    q[0] = cast_out<TO>(fminf(1.0f, bgr[2]) * out_scale);
    q[1] = cast_out<TO>(fminf(1.0f, bgr[1]) * out_scale);
    q[2] = cast_out<TO>(fminf(1.0f, bgr[0]) * out_scale);
  }
}

// ---------------------------------------------------------------------------------------------
// 8-pixel (24-element) vector IO on (H, W, 3) images
// ---------------------------------------------------------------------------------------------
// The 24 elements of an aligned whole group as raw 16-byte (8-byte for u8) units: issued early and
// kept opaque until used, so that the unpacking (and with it the wait for the data) stays at the use.
template <class T> struct Raw24 { typename IoUnit<T>::type u[IoUnits<T>::value]; };
template <class T> MI_DEV void load24_raw(const T* p, Raw24<T>& r) {
  typedef typename IoUnit<T>::type U;
  const U* s = reinterpret_cast<const U*>(p);
#pragma unroll
  for (int i = 0; i < IoUnits<T>::value; ++i) r.u[i] = s[i];
}
template <class T> MI_DEV void raw_to_float(const Raw24<T>& r, float (&v)[24]) {
  // The words pass through an empty volatile asm first: without it LLVM folds the unpacking (shifts,
  // conversions) into the loop-carried value, i.e. moves it - and the s_waitcnt for the data - up to
  // right behind the prefetching loads, which serialises load latency and compute again.
  uint32_t w[sizeof(r.u) / 4];
  __builtin_memcpy(w, r.u, sizeof(w));
#pragma unroll
  for (size_t i = 0; i < sizeof(w) / 4; ++i) asm volatile("" : "+v"(w[i]));
  T t[24];
  __builtin_memcpy(t, w, sizeof(t));
#pragma unroll
  for (int i = 0; i < 24; ++i) v[i] = (float)t[i];
}

template <class T> MI_DEV void load24(const T* p, float (&v)[24], int npx, bool vec) {
  if (vec && npx == 8) {
    T t[24];
    if (sizeof(T) == 1) {
      const uint2* s = reinterpret_cast<const uint2*>(p);
      uint2* d = reinterpret_cast<uint2*>(t);
      d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    } else {
      constexpr int N = (int)(sizeof(T) * 24 / 16);
      const uint4* s = reinterpret_cast<const uint4*>(p);
      uint4* d = reinterpret_cast<uint4*>(t);
#pragma unroll
      for (int i = 0; i < N; ++i) d[i] = s[i];
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = (float)t[i];
  } else {
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = i < npx * 3 ? (float)p[i] : 0.f;
  }
}

template <class T> MI_DEV void store24(T* p, const float (&v)[24], int npx, bool vec) {
  T o[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) o[i] = cast_out<T>(v[i]);
  if (vec && npx == 8) {
    if (sizeof(T) == 1) {
      const uint2* s = reinterpret_cast<const uint2*>(o);
      uint2* d = reinterpret_cast<uint2*>(p);
      d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    } else {
      constexpr int N = (int)(sizeof(T) * 24 / 16);
      const uint4* s = reinterpret_cast<const uint4*>(o);
      uint4* d = reinterpret_cast<uint4*>(p);
#pragma unroll
      for (int i = 0; i < N; ++i) d[i] = s[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 24; ++i)
      if (i < npx * 3) p[i] = o[i];
  }
}

// rgb -> yuv420, aligned images (W % 8 == 0, 16-byte aligned rows): one lane = 2 rows x 8 pixels =
// two 24-element vector loads, two 8-element Y stores, 4 + 4 chroma stores.
template <class T, int N> MI_DEV void store_vec(T* p, const T (&v)[N]) {
  constexpr int BYTES = (int)sizeof(T) * N;
  if constexpr (BYTES % 16 == 0) {
#pragma unroll
    for (int i = 0; i < BYTES / 16; ++i) reinterpret_cast<uint4*>(p)[i] = reinterpret_cast<const uint4*>(v)[i];
  } else if constexpr (BYTES % 8 == 0) {
#pragma unroll
    for (int i = 0; i < BYTES / 8; ++i) reinterpret_cast<uint2*>(p)[i] = reinterpret_cast<const uint2*>(v)[i];
  } else {
#pragma unroll
    for (int i = 0; i < BYTES / 4; ++i) reinterpret_cast<uint32_t*>(p)[i] = reinterpret_cast<const uint32_t*>(v)[i];
  }
}

template <class TI, class TO>
__global__ __launch_bounds__(EW_THREADS) void rgb_yuv420_vec_kernel(const TI* __restrict__ src, TO* __restrict__ yuv,
                                                                    int H, int W) {
  const int hb = H / 2, wb = W / 2, groups = W / 8;
  const float in_scale = ScaleOf<TI>::value, out_scale = ScaleOf<TO>::value;
  TO* yp = yuv;
  TO* plane0 = yuv + (size_t)H * W;
  TO* plane1 = plane0 + (size_t)hb * wb;
  const int64_t n = (int64_t)hb * groups, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int br = (int)(i / groups), g = (int)(i - (int64_t)br * groups);
    float a[2][24];
    load24<TI>(src + ((size_t)(2 * br) * W + 8 * g) * 3, a[0], 8, true);
    load24<TI>(src + ((size_t)(2 * br + 1) * W + 8 * g) * 3, a[1], 8, true);
    float u[2][8], v[2][8];
    TO yo[2][8];
#pragma unroll
    for (int dr = 0; dr < 2; ++dr)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float t[3];
        ycrcb_from_rgb(div_scale<TI>(a[dr][3 * k]), div_scale<TI>(a[dr][3 * k + 1]), div_scale<TI>(a[dr][3 * k + 2]), t);
        yo[dr][k] = cast_out<TO>(fminf(1.0f, t[0]) * out_scale);
        u[dr][k] = t[1]; v[dr][k] = t[2];
      }
    TO uo[4], vo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // the reference's order: (0,0), (0,1), (1,0), (1,1), starting from 0 (yuv_420.py:50-55)
      const float su = (((0.f + u[0][2 * c]) + u[0][2 * c + 1]) + u[1][2 * c]) + u[1][2 * c + 1];
      const float sv = (((0.f + v[0][2 * c]) + v[0][2 * c + 1]) + v[1][2 * c]) + v[1][2 * c + 1];
      uo[c] = cast_out<TO>(fminf(1.0f, su / 4.0f) * out_scale);
      vo[c] = cast_out<TO>(fminf(1.0f, sv / 4.0f) * out_scale);
    }
    store_vec<TO, 8>(yp + (size_t)(2 * br) * W + 8 * g, yo[0]);
    store_vec<TO, 8>(yp + (size_t)(2 * br + 1) * W + 8 * g, yo[1]);
    store_vec<TO, 4>(plane1 + (size_t)br * wb + 4 * g, uo);
    store_vec<TO, 4>(plane0 + (size_t)br * wb + 4 * g, vo);
  }
}

// ---------------------------------------------------------------------------------------------
// Generic RGB-image pass used by the stateless tonemaps (tonemap.py) and the ISP tonemaps
// (camera_isp.py:177-227).  MODE selects the per-pixel work; reductions leave one partial per
// block.
// ---------------------------------------------------------------------------------------------
// cache policy of the passes' stores: the in-place p of ISP pass 1 is read back by pass 2 (kept), the outputs of the
// storing passes are final (streamed)
#ifndef MI_NT_P1
#define MI_NT_P1 false
#endif
#ifndef MI_NT_OUT
#define MI_NT_OUT true
#endif
enum PassMode {
  PM_MINMAX = 0,        // bounds_func (util.py:50-60)
  PM_STATS = 1,         // linear_func(gamma 1) + metering_func (tonemap.py:78-103)
  PM_RH_MINMAX = 2,     // + reinhard_func, bounds of the result (tonemap.py:150-153)
  PM_RH_STORE = 3,      // + final linear_func (tonemap.py:154)
  PM_LINEAR_STORE = 4,  // tonemap_linear / ISP linear_kernel: linear_func straight to dst
  PM_ISP_RH_P1 = 5,     // camera_isp.py:198-213: p written back in place, max(p)
  PM_ISP_RH_P2 = 6,     // camera_isp.py:215-218: (p/max_out)^(1/gamma)*255 -> u8
  PM_ISP_RH_P2R = 7     // the same from the UNTOUCHED image: p recomputed (camera_isp.py:198-211) and rounded to the image dtype as
                        // the write-back would have, for pass 1 run without its write-back (ISP.tonemap_reinhard(write_back=False),
                        // an extension: the reference's images are mutated, these are not; the u8 outputs are the same bits)
};

struct PassArgs {
  const void* src; void* dst; void* inplace;
  const float* fp; float* partials; int part_stride;
  int64_t n_px;
  int vec_in, vec_out;
  float gamma_inv, la, ca, out_scale;
  int transform, H, W;      // optional orientation transform fused into the store (u8 ISP outputs)
  // batched launch (grid.y = image): per-image source / destination pointers and max_out scalars
  int batched;
  const float* maxouts;     // PM_ISP_RH_P2: max_out per image (camera_isp.py:190,213)
  ew::PtrList srcs, dsts;
  // Pulled finalize: instead of a one-block finalize launch between two passes, every block of the
  // consuming pass folds the producer's per-block partials itself in its prologue (identical
  // arithmetic in every block, so every block derives the same scalars); block 0 also publishes
  // them to `fp_w` for the passes after this one.  pull_mode < 0: scalars come from `fp`.
  int pull_mode = -1;       // ew::FinMode or -1
  const float* pull_partials; int pull_stride, pull_n, pull_bounds_post;
  float pull_npx, pull_intensity;
  float* fp_w;
  // ISP Reinhard (batched): pass 1 derives its scalars from the metering state in its prologue instead of
  // a 1-thread prep launch, pass 2 folds its image's partial maxima instead of a maxout launch
  const float* isp_state9;  // PM_ISP_RH_P1: camera_isp.py:186-195 evaluated per block (NULL: read fp)
  int no_nan;               // the source image holds no NaN (written by the tile kernel's clamping store)
  int pull_maxout_n;        // PM_ISP_RH_P2: partial maxima per image at partials[part_stride + y * n + i] (0: read maxouts)
  int no_writeback;         // PM_ISP_RH_P1: p is not written back over the image (pass 2 is then PM_ISP_RH_P2R)
  int part_flip;            // PM_ISP_RH_P2, batched: grid row y reads the partial maxima of row n - 1 - y (pass 1 ran over the
                            // reversed list, see mi_isp_reinhard_batch)
};

// (isp_reinhard_scalars - camera_isp.py:186-195 - lives in isp_math.h: the camera-group kernel of isp_mega_cam.h derives the
// same scalars)

// Prologue of a pass with a pulled finalize: sh_fp = the FrameParams this block works with.
template <int FIN>
MI_DEV void pull_finalize(const PassArgs& a, float* sh_fp, double (*sh_tot)[PASS_THREADS / 64]) {
  constexpr int mode = FIN;
  constexpr int nrows = mode == ew::FIN_STATS ? 7 : 2;
  // every load of the prologue is issued before the first use: one memory latency, which the
  // caller's prefetch of its first image group shares
  const float fp_mine = threadIdx.x < FP_COUNT ? a.fp[threadIdx.x] : 0.f;
  constexpr int U = nrows == 2 ? 8 : 2;              // entries per thread and round (<= 16 loads)
  float mn = __builtin_inff(), mx = -__builtin_inff();
  double sum[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int base = 0; base < a.pull_n; base += U * PASS_THREADS) {
    float v[U][nrows];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * PASS_THREADS + threadIdx.x;
      const bool ok = i < a.pull_n;
#pragma unroll
      for (int k = 0; k < nrows; ++k)
        v[u][k] = ok ? a.pull_partials[(size_t)k * a.pull_stride + i] : (k == 0 ? __builtin_inff() : k == 1 ? -__builtin_inff() : 0.f);
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
  // wave 0: lane k folds row k over the waves, the rows meet in lane 0 through DPP-free readlanes
  if (wave == 0) {
    double r = 0.0;
    if (lane < nrows) {
      r = sh_tot[lane][0];
      for (int w = 1; w < PASS_THREADS / 64; ++w)
        r = lane == 0 ? fmin(r, sh_tot[lane][w]) : (lane == 1 ? fmax(r, sh_tot[lane][w]) : r + sh_tot[lane][w]);
    }
    double tot[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const long long b = __builtin_bit_cast(long long, r);
      const int lo = __builtin_amdgcn_readlane((int)b, k), hi = __builtin_amdgcn_readlane((int)(b >> 32), k);
      tot[k] = k < nrows ? __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo) : 0.0;
    }
    if (lane == 0 && !(a.pull_mode & 0x100)) {
      ew::FinArgs fa = {};
      fa.fp = sh_fp; fa.n_px = a.pull_npx; fa.intensity = a.pull_intensity; fa.la = a.la; fa.ca = a.ca;
      fa.bounds_post = a.pull_bounds_post;
      ew::finalize_scalars<true>(mode, fa, tot);
    }
  }
  __syncthreads();
}

// per-wave LDS staging of the wave-cooperative stores decides whether two 1024-thread blocks fit a CU
template <class TI, class TO, int MODE> constexpr int pass_lds_bytes() {
  const bool uses_io = MODE == PM_RH_STORE || MODE == PM_LINEAR_STORE || MODE == PM_ISP_RH_P2 || MODE == PM_ISP_RH_P2R || MODE == PM_ISP_RH_P1;
  return uses_io ? (PASS_THREADS / 64) * 64 * 24 * (int)(sizeof(TI) > sizeof(TO) ? sizeof(TI) : sizeof(TO)) : 0;
}
template <class TI, class TO, int MODE>
__global__ __launch_bounds__(PASS_THREADS, 4) void rgb_pass_kernel(const PassArgs a) {
#pragma clang fp contract(fast)
  __shared__ float red[PASS_THREADS / 64][8];
  // (The pointer lists are indexed by blockIdx.y itself and nothing derived from it: with a computed index the compiler
  // copies the 1.2 KB argument struct into scratch, per thread, to index it - 2.4 x the time of both ISP passes.  The
  // host orders the lists instead.)
  const TI* src = static_cast<const TI*>(a.batched ? a.srcs.p[blockIdx.y] : a.src);
  TO* dst = static_cast<TO*>(a.batched ? const_cast<void*>(a.dsts.p[blockIdx.y]) : a.dst);
  TI* inplace = a.batched ? const_cast<TI*>(src) : static_cast<TI*>(a.inplace);
  const int pblock = a.batched ? blockIdx.y * gridDim.x + blockIdx.x : blockIdx.x;
  const int64_t n_groups = (a.n_px + 7) / 8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr bool P2ANY = MODE == PM_ISP_RH_P2 || MODE == PM_ISP_RH_P2R;
  constexpr bool STORES = MODE == PM_RH_STORE || MODE == PM_LINEAR_STORE || P2ANY;

  // the pass's scalars: from the pulled finalize of the producer's partials, or from FrameParams
  __shared__ float sh_fp[FP_COUNT];
  __shared__ double sh_tot[7][PASS_THREADS / 64];
  // which finalize a pass pulls is fixed by what it consumes
  constexpr int PULL_FIN = MODE == PM_STATS || MODE == PM_LINEAR_STORE ? (int)ew::FIN_BOUNDS
                           : MODE == PM_RH_MINMAX ? (int)ew::FIN_STATS
                           : MODE == PM_RH_STORE ? (int)ew::FIN_BOUNDS2 : -1;
  const bool pulled = PULL_FIN >= 0 && a.pull_mode >= 0;
  const int lane = threadIdx.x & 63;

  // Whole groups with aligned buffers and no transform take the FULL path (straight-line code, wave-
  // cooperative stores); the rest (a ragged tail, unaligned views, transformed stores) the general one.
  // FULL needs all 64 groups of the wave whole: n_full is a multiple of 64 groups and `tid` strides by
  // whole waves, so `g < n_full` is wave-uniform.
  const bool can_full = a.vec_in && (!STORES || (a.vec_out && a.transform == MI_T_NONE));
  const int64_t n_full = can_full ? (a.n_px / 8) / 64 * 64 : 0;
  // the first group's loads are in flight while the prologue folds the producer's partials

  constexpr int PREFETCH = 1;                       // deeper (2-3 groups) measured slower on gfx950
  Raw24<TI> raws[PREFETCH];
#pragma unroll
  for (int d = 0; d < PREFETCH; ++d)
    if (tid + d * stride < n_full) load24_raw<TI>(src + (tid + d * stride) * 24, raws[d]);

  if constexpr (PULL_FIN >= 0) {
    if (pulled) pull_finalize<PULL_FIN>(a, sh_fp, sh_tot);
  }
  bool isp_prep = false;
  if constexpr (MODE == PM_ISP_RH_P1 || MODE == PM_ISP_RH_P2R) {
    isp_prep = a.isp_state9 != nullptr;
    if (isp_prep) {
      if (threadIdx.x == 0) isp_reinhard_scalars(a.isp_state9, sh_fp, a.pull_intensity, a.ca);
      __syncthreads();
    }
  }
  auto fpv = [&](int i) { return (pulled || isp_prep) ? sh_fp[i] : a.fp[i]; };

  float lo = 0.f, inv = 1.f, lo2 = 0.f, inv2 = 1.f, maxout_inv = 1.f;
  ReinhardK rk;
  rk.la = a.la; rk.ca = a.ca; rk.map_key = 1.f; rk.ei = 1.f; rk.mean3[0] = rk.mean3[1] = rk.mean3[2] = 0.f;
  if (MODE != PM_MINMAX && MODE != PM_ISP_RH_P2) { lo = fpv(FP_LO); inv = fpv(FP_INV); }
  if (MODE == PM_RH_MINMAX || MODE == PM_RH_STORE || MODE == PM_ISP_RH_P1 || MODE == PM_ISP_RH_P2R) {
    rk.map_key = fpv(FP_MAPKEY); rk.ei = fpv(FP_EI);
    rk.mean3[0] = fpv(FP_MEAN3); rk.mean3[1] = fpv(FP_MEAN3 + 1); rk.mean3[2] = fpv(FP_MEAN3 + 2);
  }
  if (MODE == PM_RH_STORE) { lo2 = fpv(FP_LO2); inv2 = fpv(FP_INV2); }
  if constexpr (P2ANY) {
    if constexpr (MODE == PM_ISP_RH_P2R) __syncthreads();     // (the scalars above were read out of sh_fp, which the fold re-uses)
    if (a.batched && a.pull_maxout_n > 0) {
      // max_out of this block's image (camera_isp.py:190,213): fold the partial maxima pass 1 left
      const int prow = a.part_flip ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
      const float* pm = a.partials + a.part_stride + (size_t)prow * a.pull_maxout_n;
      float m = -__builtin_inff();
      for (int i = threadIdx.x; i < a.pull_maxout_n; i += PASS_THREADS) m = fmaxf(m, pm[i]);
      m = wave_max(m);
      if ((threadIdx.x & 63) == 0) sh_fp[threadIdx.x >> 6] = m;
      __syncthreads();
      m = sh_fp[0];
#pragma unroll
      for (int w = 1; w < PASS_THREADS / 64; ++w) m = fmaxf(m, sh_fp[w]);
      maxout_inv = 1.0f / fmaxf(1e-6f, m);
    } else {
      maxout_inv = 1.0f / (a.batched ? a.maxouts[blockIdx.y] : a.fp[FP_MAXOUT]);
    }
  }

  float vmin = __builtin_inff(), vmax = -__builtin_inff();
  StatsAcc st; st.init();

  constexpr bool USES_IO = STORES || MODE == PM_ISP_RH_P1;
  constexpr int IO_BYTES = 64 * 24 * (int)(sizeof(TI) > sizeof(TO) ? sizeof(TI) : sizeof(TO));   // per-wave staging
  __shared__ __attribute__((aligned(16))) unsigned char io_buf[USES_IO ? PASS_THREADS / 64 : 1][IO_BYTES];
  void* wbuf = io_buf[USES_IO ? threadIdx.x >> 6 : 0];

  // One group = 8 pixels = 24 elements.  FULL: the wave's 64 groups are whole, aligned and
  // contiguous, no orientation transform -> wave-contiguous IO through LDS and straight-line code
  // without per-pixel tests.  CA0: color_adapt == 0 (one pow/px).
  auto group = [&](auto full_c, auto ca0_c, auto unit_c, int64_t g, Raw24<TI>& raw) {
    constexpr bool UNIT = decltype(unit_c)::value;       // bounds exactly (0, 1): norm01 is the identity
    constexpr bool FULL = decltype(full_c)::value;
    constexpr bool CA0 = decltype(ca0_c)::value;
    const int64_t px0 = g * 8;
    const int npx = FULL ? 8 : (int)(a.n_px - px0 < 8 ? a.n_px - px0 : 8);
    float v[24], o[24];
    // loads stay per-lane (3 x 16 B at a 48-B lane stride): measured as fast as wave-contiguous
    // loads through LDS for reads; the stores below do go through LDS (3.7 -> 6.2 TB/s)
    if constexpr (FULL) {
      // software pipeline: this group was loaded one iteration ago; start the next one now
      raw_to_float<TI>(raw, v);
      if (g + PREFETCH * stride < n_full) load24_raw<TI>(src + (g + PREFETCH * stride) * 24, raw);
    } else {
      load24<TI>(src + px0 * 3, v, npx, a.vec_in);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool live = FULL || k < npx;
      float x[3] = {v[3 * k], v[3 * k + 1], v[3 * k + 2]};
      if (MODE == PM_MINMAX) {
        if (live) {
          vmin = fminf(vmin, fminf(x[0], fminf(x[1], x[2])));
          vmax = fmaxf(vmax, fmaxf(x[0], fmaxf(x[1], x[2])));
        }
      } else if (MODE == PM_LINEAR_STORE) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[3 * k + ch] = x[ch];
      } else if (MODE == PM_ISP_RH_P1) {
        // camera_isp.py:200: no clamp on the normalised value here
        float t[3], q[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) t[ch] = isp_norm(x[ch], lo, inv);
        reinhard_px<CA0>(t, rk, q);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[3 * k + ch] = q[ch];
        if (live) vmax = fmaxf(vmax, fmaxf(q[0], fmaxf(q[1], q[2])));
      } else if (MODE == PM_ISP_RH_P2) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[3 * k + ch] = x[ch] * maxout_inv;
      } else if (MODE == PM_ISP_RH_P2R) {
        // p as pass 1 computes it (camera_isp.py:200-210), rounded to the image dtype as the write-back stores it (:211)
        float t[3], q[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) t[ch] = isp_norm(x[ch], lo, inv);
        reinhard_px<CA0>(t, rk, q);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[3 * k + ch] = (float)cast_out<TI>(q[ch]) * maxout_inv;
      } else {
        float t[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) t[ch] = UNIT ? x[ch] : norm01(x[ch], lo, inv);
        if (MODE == PM_STATS) {
          if (live) st.add(t[0], t[1], t[2]);
        } else {
          float q[3];
          reinhard_px<CA0>(t, rk, q);
          if (MODE == PM_RH_MINMAX) {
            if (live) {
              vmin = fminf(vmin, fminf(q[0], fminf(q[1], q[2])));
              vmax = fmaxf(vmax, fmaxf(q[0], fmaxf(q[1], q[2])));
            }
          } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) o[3 * k + ch] = q[ch];
          }
        }
      }
    }
    if (P2ANY) {                                     // camera_isp.py:217-218 (no clamp there)
      if (a.gamma_inv != 1.f) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 24; ++j) o[j] = hw_pow(o[j], a.gamma_inv);
      }
#pragma unroll
      for (int j = 0; j < 24; ++j) o[j] *= 255.f;
    }
    if (MODE == PM_LINEAR_STORE) linear_n<24>(o, lo, inv, a.gamma_inv, a.out_scale);
    if (MODE == PM_RH_STORE) linear_n<24>(o, lo2, inv2, a.gamma_inv, a.out_scale);
    if (MODE == PM_ISP_RH_P1 && a.no_writeback) {
      // (nothing to store: pass 2 recomputes p from the image)
    } else if (MODE == PM_ISP_RH_P1) {
      if constexpr (FULL) {
        TI ot[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) ot[i] = cast_out<TI>(o[i]);
        wave_store24<TI, MI_NT_P1>(inplace + (px0 - (int64_t)lane * 8) * 3, lane, wbuf, ot);
      } else {
        store24<TI>(inplace + px0 * 3, o, npx, a.vec_in);
      }
    } else if (STORES) {
      if constexpr (FULL) {
        TO ot[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) ot[i] = cast_out<TO>(o[i]);
        wave_store24<TO, MI_NT_OUT>(dst + (px0 - (int64_t)lane * 8) * 3, lane, wbuf, ot);
      } else if (a.transform == MI_T_NONE) {
        store24<TO>(dst + px0 * 3, o, npx, a.vec_out);
      } else {
        // scatter: source pixel (r, c) -> the destination pixel that reads it (inverse of
        // interpolate.py:36-54); destination dims (Hd, Wd)
        const bool swap = a.transform == MI_T_ROTATE_90 || a.transform == MI_T_ROTATE_270 ||
                          a.transform == MI_T_TRANSPOSE;
        const int Hd = swap ? a.W : a.H, Wd = swap ? a.H : a.W;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (k >= npx) break;
          const int64_t px = px0 + k;
          const int r = (int)(px / a.W), c = (int)(px - (int64_t)r * a.W);
          int dr, dc;
          switch (a.transform) {
            case MI_T_ROTATE_90: dr = c; dc = Wd - 1 - r; break;        // sr = Wd-c'-1, sc = r'
            case MI_T_ROTATE_180: dr = Hd - 1 - r; dc = Wd - 1 - c; break;
            case MI_T_ROTATE_270: dr = Hd - 1 - c; dc = r; break;       // sr = c', sc = Hd-r'-1
            case MI_T_TRANSPOSE: dr = c; dc = r; break;
            case MI_T_FLIP_VERT: dr = Hd - 1 - r; dc = c; break;
            case MI_T_FLIP_HORIZ: dr = r; dc = Wd - 1 - c; break;
            default: dr = Hd - 1 - c; dc = Wd - 1 - r; break;           // transverse (square only)
          }
          TO* q = dst + ((size_t)dr * Wd + dc) * 3;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) q[ch] = cast_out<TO>(o[3 * k + ch]);
        }
      }
    }
  };

  // (only for images the tile kernel wrote - a.no_nan: its clamp has already mapped NaN to 0, which norm01 would do here)
  const bool unit_bounds = (MODE == PM_STATS || MODE == PM_RH_MINMAX || MODE == PM_RH_STORE) && a.no_nan && lo == 0.f && inv == 1.f;
  // the colour-adapt variant (three pows per pixel) only exists for the modes that evaluate reinhard_px
  constexpr bool HAS_REINHARD = MODE == PM_RH_MINMAX || MODE == PM_RH_STORE || MODE == PM_ISP_RH_P1 || MODE == PM_ISP_RH_P2R;
  if (!HAS_REINHARD || rk.ca == 0.f) {
    for (int64_t g = tid; g < n_full; g += PREFETCH * stride) {
#pragma unroll
      for (int d = 0; d < PREFETCH; ++d)
        if (g + d * stride < n_full) {
          // image bounds exactly (0, 1) - every frame with a clipped pixel at both ends: clamp((x - 0) * 1) == x
          if (unit_bounds) group(std::true_type{}, std::true_type{}, std::true_type{}, g + d * stride, raws[d]);
          else group(std::true_type{}, std::true_type{}, std::false_type{}, g + d * stride, raws[d]);
        }
    }
    for (int64_t g = n_full + tid; g < n_groups; g += stride) group(std::false_type{}, std::true_type{}, std::false_type{}, g, raws[0]);
  } else {
    // colour adaptation (three pows per pixel): whole groups keep the wave-cooperative IO
    for (int64_t g = tid; g < n_full; g += PREFETCH * stride) {
#pragma unroll
      for (int d = 0; d < PREFETCH; ++d)
        if (g + d * stride < n_full) group(std::true_type{}, std::false_type{}, std::false_type{}, g + d * stride, raws[d]);
    }
    for (int64_t g = n_full + tid; g < n_groups; g += stride) group(std::false_type{}, std::false_type{}, std::false_type{}, g, raws[0]);
  }

  if constexpr (PULL_FIN >= 0) {
    // block 0 publishes the pulled scalars for the passes after this one (off the critical path)
    if (pulled && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < FP_COUNT && a.fp_w &&
        ew::finalize_writes(PULL_FIN, threadIdx.x))
      a.fp_w[threadIdx.x] = sh_fp[threadIdx.x];
  }
  if (MODE == PM_MINMAX || MODE == PM_RH_MINMAX || MODE == PM_ISP_RH_P1) {
    const float v2[2] = {vmin, vmax};
    const int op[2] = {0, 1};
    block_reduce_store<2>(v2, op, red, a.partials, a.part_stride, pblock);
  } else if (MODE == PM_STATS) {
    const float v7[7] = {st.gmin, st.gmax, st.slog, st.sgray, st.s0, st.s1, st.s2};
    const int op[7] = {0, 1, 2, 2, 2, 2, 2};
    block_reduce_store<7>(v7, op, red, a.partials, a.part_stride, pblock);
  }
}

// ---------------------------------------------------------------------------------------------
// ISP Reinhard (camera_isp.py:177-218) of a whole camera group in ONE persistent launch (round 4).
//
// The two passes of rgb_pass_kernel<PM_ISP_RH_P1 / P2> communicate through memory: pass 1 writes p back over the image
// (camera_isp.py:211 - the reference mutates its input, so that write stays), pass 2 reads it again for the final map,
// and a kernel boundary sits between them because max_out (camera_isp.py:213) is a reduction over the whole image.
// Here a resident grid (ISPF_BPC blocks per CU, checked against the occupancy query) keeps every image's ROUNDED p - the
// very bytes it stores - in registers: 12 packed dwords per 8-pixel group (f16), `iters` groups per thread.  HBM sees the
// image in, p out (the in-place semantics) and the u8 result out; the 2 B/value re-read of p and one launch are gone.
//   P1(k)  Reinhard of image k: p stored in place and kept; the block's maximum goes to the image's max word of the
//          block's XCD (atomic max on the bits of a non-negative float), then the XCD's arrival counter is bumped
//   W(k)   wave 0 polls the image's 8 {count, max} pairs (one 8-byte sc1 load per lane) until the counts add up to the grid
//   P2(k)  (p / max_out)^(1/gamma) * 255 -> u8 from the kept p (camera_isp.py:215-218)
// PIPE: P1(k + 1) runs between P1(k)'s post and W(k), so the barrier's latency (~2-3 us after the last arrival) hides
// behind a whole pass of the next image; two images' p are resident.  Images too large for that (up to 6 groups per
// thread, 6.3 MP) run P1 / W / P2 one image at a time; larger ones (a 4096 x 3072 frame is 12 groups per thread: its kept p
// does not fit the registers next to the working set) stay with the two passes.
// The {count, max} words live in a buffer of the library's own (per device, zero at allocation), in two halves: launch n
// uses half n & 1 and clears the other one for launch n + 1 (launches of resident-grid kernels are serialised per device:
// ew::resident_order); a launch that timed out leaves its half to be cleared by the next launch but one.
// ---------------------------------------------------------------------------------------------
template <int B, int E, class F> MI_DEV void ispf_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    ispf_static_for<B + 1, E>(f);
  }
}
constexpr int ISPF_THREADS = 256;
constexpr int ISPF_BPC = 2;               // blocks per CU: 8 waves per CU, up to 256 VGPRs each (16 waves, 128 VGPRs: the BPC = 4 variant)
constexpr int ISPF_MAX_IMAGES = 64;
constexpr int ISPF_SYNC_WORDS = ISPF_MAX_IMAGES * 8 * 16;   // per half: (image, XCD) -> 64 bytes {count, max bits, ...}
struct IspFusedArgs {
  int n_images, iters;                    // groups of 8 pixels per thread and image
  int64_t n_groups;                       // per image: a multiple of 64 (whole waves)
  float gamma_inv, intensity, la, ca;
  const float* state9;
  unsigned* sync;                         // this launch's half
  unsigned* sync_next;                    // the other half: cleared here
  unsigned spin_limit;
  unsigned* fault;                        // the workspace's fault word
  unsigned* mailbox;                      // host-mapped word of the device
  struct IO { void* img; uint8_t* out; } io[ISPF_MAX_IMAGES];
};

// NLDS: the last NLDS of a thread's MAXIT groups keep their p in LDS ([group][16-byte unit][thread]: a wave's accesses are
// contiguous) instead of registers - what lets a 4096 x 3072 f16 frame (12 groups per thread: 7 in 84 registers, 5 in 60 KB
// of LDS per block) stay on the chip between the passes.
template <class TI, bool CA0, int MAXIT, bool PIPE, int BPC, int NLDS = 0>
__global__ __launch_bounds__(ISPF_THREADS, BPC) void isp_reinhard_fused_kernel(const IspFusedArgs a) {
#pragma clang fp contract(fast)
  static_assert(NLDS == 0 || (!PIPE && sizeof(TI) == 2), "LDS-kept groups: one image at a time, f16");
  constexpr int NREG = MAXIT - NLDS;
  __shared__ __attribute__((aligned(16))) uint4 lds_keep[NLDS > 0 ? NLDS : 1][3][NLDS > 0 ? ISPF_THREADS : 1];
  __shared__ float sh_fp[FP_COUNT];
  __shared__ float sh_red[ISPF_THREADS / 64];
  __shared__ float sh_max[2];
  __shared__ __attribute__((aligned(16))) unsigned char io_buf[ISPF_THREADS / 64][64 * 24 * sizeof(TI)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  void* wbuf = io_buf[wave];
  const int64_t stride = (int64_t)gridDim.x * ISPF_THREADS;
  const int64_t tid = (int64_t)blockIdx.x * ISPF_THREADS + threadIdx.x;
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;     // XCC_ID: which XCD's words this block bumps

  // the other half of the sync words, for the next launch (nobody reads it during this one)
  for (int i = blockIdx.x * ISPF_THREADS + threadIdx.x; i < ISPF_SYNC_WORDS; i += gridDim.x * ISPF_THREADS) a.sync_next[i] = 0u;
  if (threadIdx.x == 0) isp_reinhard_scalars(a.state9, sh_fp, a.intensity, a.ca);      // camera_isp.py:186-195
  __syncthreads();
  const float lo = sh_fp[FP_LO], inv = sh_fp[FP_INV];
  ReinhardK rk;
  rk.la = a.la; rk.ca = a.ca; rk.map_key = sh_fp[FP_MAPKEY]; rk.ei = sh_fp[FP_EI];
  rk.mean3[0] = sh_fp[FP_MEAN3]; rk.mean3[1] = sh_fp[FP_MEAN3 + 1]; rk.mean3[2] = sh_fp[FP_MEAN3 + 2];

  constexpr int SLOTS = PIPE ? 2 : 1;
  Raw24<TI> keep[SLOTS][NREG];
  Raw24<TI> raw;                                             // the group in flight (see p1)
  if (tid < a.n_groups) load24_raw<TI>(static_cast<const TI*>(a.io[0].img) + tid * 24, raw);

  // ---- P1: camera_isp.py:198-213 ----
  auto p1 = [&](int k, auto slot_c) {
    constexpr int S = decltype(slot_c)::value;
    TI* img = static_cast<TI*>(a.io[k].img);
    const TI* img_next = static_cast<const TI*>(a.io[k + 1 < a.n_images ? k + 1 : k].img);
    float vmax = -__builtin_inff();
    // `raw` holds this image's first group already (asked for during the previous image's pass: a thread has only two or
    // three groups per image, and a first load that starts with the pass is ~1.5 us of exposed latency per image)
    ispf_static_for<0, MAXIT>([&](auto itc) {
      constexpr int IT = decltype(itc)::value;
      const int64_t g = tid + IT * stride;
      if (IT < a.iters && g < a.n_groups) {                  // (whole waves: n_groups and stride are multiples of 64)
        float v[24], o[24];
        raw_to_float<TI>(raw, v);
        // the next group is asked for while this one is computed: of this image, or the first one of the next image
        if (IT + 1 < a.iters && g + stride < a.n_groups) load24_raw<TI>(img + (g + stride) * 24, raw);
        else if (k + 1 < a.n_images) load24_raw<TI>(img_next + tid * 24, raw);
#pragma unroll
        for (int px = 0; px < 8; ++px) {
          float t[3], q[3];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) t[ch] = isp_norm(v[3 * px + ch], lo, inv);      // camera_isp.py:200: no clamp here
          reinhard_px<CA0>(t, rk, q);
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) o[3 * px + ch] = q[ch];
          vmax = fmaxf(vmax, fmaxf(q[0], fmaxf(q[1], q[2])));                       // of the un-rounded p (camera_isp.py:213)
        }
        // what pass 2 reads back: the ROUNDED p.  f16: packed pairs (v_cvt_pk_f16_f32 = RNE per element, the bits of
        // cast_out<half_t>), made opaque - left to itself the compiler keeps every half in a register of its own
        TI ot[24];
        if constexpr (sizeof(TI) == 2) {
          uint32_t w[12];
#pragma unroll
          for (int j = 0; j < 12; ++j) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w[j]) : "v"(o[2 * j]), "v"(o[2 * j + 1]));
          if constexpr (IT < NREG) {
            __builtin_memcpy(&keep[S][IT], w, sizeof(w));
          } else {
            uint4 u[3];
            __builtin_memcpy(u, w, sizeof(w));
#pragma unroll
            for (int j = 0; j < 3; ++j) lds_keep[IT - NREG][j][threadIdx.x] = u[j];
          }
          __builtin_memcpy(ot, w, sizeof(w));
        } else {
#pragma unroll
          for (int i = 0; i < 24; ++i) ot[i] = cast_out<TI>(o[i]);
          __builtin_memcpy(&keep[S][IT < NREG ? IT : 0], ot, sizeof(ot));
        }
        wave_store24<TI, false>(img + (g - lane) * 24, lane, wbuf, ot);            // camera_isp.py:211
      }
    });
    // the block's maximum -> the image's word of this XCD, then the arrival
    vmax = wave_max(vmax);
    if (lane == 0) sh_red[wave] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = sh_red[0];
#pragma unroll
      for (int w = 1; w < ISPF_THREADS / 64; ++w) m = fmaxf(m, sh_red[w]);
      m = fmaxf(m, 0.f);                                     // max_out = max(1e-6, .): negative maxima never matter; NaN dropped
      unsigned* w2 = a.sync + ((unsigned)k * 8u + xcc) * 16u;
      unsigned old = __hip_atomic_fetch_max(w2 + 1, __builtin_bit_cast(unsigned, m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(old) :: "memory");                 // the maximum is in before the arrival counts
      __hip_atomic_fetch_add(w2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  };
  // ---- W: max_out of image k (camera_isp.py:190,213) ----
  auto wait_max = [&](int k) {
    if (wave == 0) {
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.sync, 0, ISPF_SYNC_WORDS * 4, 0x00020000);
      const uint32_t off = lane < 8 ? ((uint32_t)k * 8u + (uint32_t)lane) * 64u : 0xFFFFFFFFu;
      unsigned spins = 0;
      float m = 0.f;
      for (;;) {
        // sc1: past this CU's L1.  (A 16-byte load whose words are copied out by name: hipcc 7.2 turned the 8-byte form
        // into ONE dword load and used the count for the maximum as well - the .x-for-.y defect of the metering kernel.)
        const u4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
        uint32_t tx = t.x, ty = t.y;
        asm volatile("" : "+v"(tx), "+v"(ty));
        unsigned cnt = lane < 8 ? tx : 0u;
        m = lane < 8 ? __builtin_bit_cast(float, ty) : 0.f;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { cnt += __shfl_xor(cnt, o, 64); m = fmaxf(m, __shfl_xor(m, o, 64)); }
        if (__builtin_amdgcn_readfirstlane(cnt) >= gridDim.x) break;
        if (++spins > a.spin_limit) {
          if (lane == 0) {
            __hip_atomic_store(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.mailbox) __hip_atomic_store(a.mailbox, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      if (lane == 0) sh_max[k & 1] = fmaxf(1e-6f, m);
    }
    __syncthreads();
    return sh_max[k & 1];
  };
  // ---- P2: camera_isp.py:215-218 ----
  auto p2 = [&](int k, auto slot_c, float max_out) {
    constexpr int S = decltype(slot_c)::value;
    uint8_t* out = a.io[k].out;
    const float maxout_inv = 1.0f / max_out;
    ispf_static_for<0, MAXIT>([&](auto itc) {
      constexpr int IT = decltype(itc)::value;
      const int64_t g = tid + IT * stride;
      if (IT < a.iters && g < a.n_groups) {
        float o[24];
        if constexpr (IT < NREG) {
          raw_to_float<TI>(keep[S][IT], o);                  // (through an opaque copy of the packed words)
        } else {
          Raw24<TI> r;
          uint4 u[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) u[j] = lds_keep[IT - NREG][j][threadIdx.x];
          __builtin_memcpy(&r, u, sizeof(u));
          raw_to_float<TI>(r, o);
        }
#pragma unroll
        for (int i = 0; i < 24; ++i) o[i] *= maxout_inv;
        if (a.gamma_inv != 1.f) {
          asm volatile("" ::: "memory");
#pragma unroll
          for (int j = 0; j < 24; ++j) o[j] = hw_pow(o[j], a.gamma_inv);
        }
        uint8_t ot[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) ot[i] = cast_out<uint8_t>(o[i] * 255.f);    // no clamp there; the cast saturates, NaN -> 0
        wave_store24<uint8_t, true>(out + (g - lane) * 24, lane, wbuf, ot);
      }
    });
  };

  if constexpr (PIPE) {
    // P1(0) P1(1) W(0) P2(0) P1(2) W(1) P2(1) ...: slots alternate, indices static
    for (int k = 0; k < a.n_images; k += 2) {
      p1(k, std::integral_constant<int, 0>{});
      if (k >= 1) p2(k - 1, std::integral_constant<int, 1>{}, wait_max(k - 1));
      if (k + 1 < a.n_images) p1(k + 1, std::integral_constant<int, 1>{});
      p2(k, std::integral_constant<int, 0>{}, wait_max(k));
    }
    if ((a.n_images & 1) == 0) p2(a.n_images - 1, std::integral_constant<int, 1>{}, wait_max(a.n_images - 1));
  } else {
    for (int k = 0; k < a.n_images; ++k) {
      p1(k, std::integral_constant<int, 0>{});
      p2(k, std::integral_constant<int, 0>{}, wait_max(k));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ISP Reinhard pass 2 with the u8 result converted to planar YUV 4:2:0 on the way (the step after
// the path for video encoders): per lane 2 rows x 8 pixels.  Exactly rgb_yuv420(u8 image of pass 2): the u8
// values are formed (camera_isp.py:215-218), then fed to the conversion of color/yuv_420.py:39-66 - the u8 RGB
// image itself (3 B/px written, 3 B/px read back) is never stored.  grid = (blocks, n_images).
// ---------------------------------------------------------------------------------------------
template <class TI>
__global__ __launch_bounds__(EW_THREADS) void isp_p2_yuv420_kernel(const ew::PtrList srcs, const ew::PtrList dsts, int H, int W,
                                                                   float gamma_inv, const float* __restrict__ pmax, int nb) {
#pragma clang fp contract(fast)
  __shared__ float sh[EW_THREADS / 64];
  const TI* src = static_cast<const TI*>(srcs.p[blockIdx.y]);
  uint8_t* yuv = static_cast<uint8_t*>(const_cast<void*>(dsts.p[blockIdx.y]));
  // max_out of this image (camera_isp.py:190,213) from the partial maxima of pass 1
  float m = -__builtin_inff();
  for (int i = threadIdx.x; i < nb; i += EW_THREADS) m = fmaxf(m, pmax[(size_t)blockIdx.y * nb + i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  m = sh[0];
#pragma unroll
  for (int w = 1; w < EW_THREADS / 64; ++w) m = fmaxf(m, sh[w]);
  const float maxout_inv = 1.0f / fmaxf(1e-6f, m);

  const int hb = H / 2, wb = W / 2, groups = W / 8;
  uint8_t* yp = yuv;
  uint8_t* plane0 = yuv + (size_t)H * W;
  uint8_t* plane1 = plane0 + (size_t)hb * wb;
  const int64_t n = (int64_t)hb * groups, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int br = (int)(i / groups), g = (int)(i - (int64_t)br * groups);
    float a[2][24];
    load24<TI>(src + ((size_t)(2 * br) * W + 8 * g) * 3, a[0], 8, true);
    load24<TI>(src + ((size_t)(2 * br + 1) * W + 8 * g) * 3, a[1], 8, true);
    float u[2][8], v[2][8];
    uint8_t yo[2][8];
#pragma unroll
    for (int dr = 0; dr < 2; ++dr) {
#pragma unroll
      for (int j = 0; j < 24; ++j) a[dr][j] *= maxout_inv;
      if (gamma_inv != 1.f) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 24; ++j) a[dr][j] = hw_pow(a[dr][j], gamma_inv);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float rgb[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)                 // the u8 pixel of pass 2, back in [0, 1] (yuv_420.py:52)
          rgb[ch] = div_scale<uint8_t>((float)cast_out<uint8_t>(a[dr][3 * k + ch] * 255.f));
        float t[3];
        {
#pragma clang fp contract(off)
          ycrcb_from_rgb(rgb[0], rgb[1], rgb[2], t);
        }
        yo[dr][k] = cast_out<uint8_t>(fminf(1.0f, t[0]) * 255.f);
        u[dr][k] = t[1]; v[dr][k] = t[2];
      }
    }
    uint8_t uo[4], vo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma clang fp contract(off)
      const float su = (((0.f + u[0][2 * c]) + u[0][2 * c + 1]) + u[1][2 * c]) + u[1][2 * c + 1];
      const float sv = (((0.f + v[0][2 * c]) + v[0][2 * c + 1]) + v[1][2 * c]) + v[1][2 * c + 1];
      uo[c] = cast_out<uint8_t>(fminf(1.0f, su / 4.0f) * 255.f);
      vo[c] = cast_out<uint8_t>(fminf(1.0f, sv / 4.0f) * 255.f);
    }
    store_vec<uint8_t, 8>(yp + (size_t)(2 * br) * W + 8 * g, yo[0]);
    store_vec<uint8_t, 8>(yp + (size_t)(2 * br + 1) * W + 8 * g, yo[1]);
    store_vec<uint8_t, 4>(plane1 + (size_t)br * wb + 4 * g, uo);
    store_vec<uint8_t, 4>(plane0 + (size_t)br * wb + 4 * g, vo);
  }
}

// image[::stride, ::stride] as a dense (hs, ws, 3) image: what ISP.update_metering reads (camera_isp.py:168-170), for the
// loaders that do not leave it on the way (mi_isp_load_packed_metered)
template <class T>
__global__ __launch_bounds__(EW_THREADS) void subsample_kernel(const T* __restrict__ img, T* __restrict__ sub, int H, int W, int stride) {
  const int hs = (H + stride - 1) / stride, ws = (W + stride - 1) / stride;
  const int n = hs * ws;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int sr = i / ws, sc = i - sr * ws;
    const T* p = img + ((size_t)(sr * stride) * W + (size_t)sc * stride) * 3;
    T* q = sub + (size_t)i * 3;
    q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
  }
}

// ---------------------------------------------------------------------------------------------
// K8 metering on the stride-subsampled images (camera_isp.py:142-175), two data passes.
// grid = (blocks_per_image, n_images); partial index = blockIdx.y * gridDim.x + blockIdx.x + base
// ---------------------------------------------------------------------------------------------
template <class T, int PHASE>
__global__ __launch_bounds__(EW_THREADS) void metering_kernel(const ew::PtrList imgs, int H, int W,
                                                              int stride, const float* bounds,
                                                              float* partials, int part_stride,
                                                              int part_base) {
#pragma clang fp contract(fast)
  __shared__ float red[4][8];
  const T* img = static_cast<const T*>(imgs.p[blockIdx.y]);
  const int hs = (H + stride - 1) / stride, ws = (W + stride - 1) / stride;
  const int n = hs * ws;
  float vmin = __builtin_inff(), vmax = -__builtin_inff();
  StatsAcc st; st.init();
  float bmin = 0.f, dinv = 1.f;
  if (PHASE == 1) {
    bmin = bounds[0];
    dinv = 1.0f / (bounds[1] - bounds[0] + 1e-6f);   // camera_isp.py:119
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int sr = i / ws, sc = i - sr * ws;
    const T* p = img + ((size_t)(sr * stride) * W + (size_t)sc * stride) * 3;
    const float x0 = (float)p[0], x1 = (float)p[1], x2 = (float)p[2];
    if (PHASE == 0) {
      vmin = fminf(vmin, fminf(x0, fminf(x1, x2)));
      vmax = fmaxf(vmax, fmaxf(x0, fmaxf(x1, x2)));
    } else {
      st.add((x0 - bmin) * dinv, (x1 - bmin) * dinv, (x2 - bmin) * dinv);
    }
  }
  const int block = part_base + blockIdx.y * gridDim.x + blockIdx.x;
  if (PHASE == 0) {
    const float v2[2] = {vmin, vmax};
    const int op[2] = {0, 1};
    block_reduce_store<2>(v2, op, red, partials, part_stride, block);
  } else {
    const float v7[7] = {st.gmin, st.gmax, st.slog, st.sgray, st.s0, st.s1, st.s2};
    const int op[7] = {0, 1, 2, 2, 2, 2, 2};
    block_reduce_store<7>(v7, op, red, partials, part_stride, block);
  }
}

// ---------------------------------------------------------------------------------------------
// The whole of update_metering (camera_isp.py:142-175) in ONE launch: bounds pass, blend of the bounds with the state,
// statistics pass, update of the state.  The four launches it replaces (two data passes over a few MB and two
// one-block finalize kernels, 4.5 - 7 us each) are all latency: 24 us per camera group.  At most one block per CU, all
// resident, meeting at a grid barrier of the whole-frame kernel's kind (isp_mega.h): a block posts {values, tag} with one
// 16-byte write-through store per record, every block polls all records (one per thread) until the tags are those of
// this launch, and folds them in index order - every block derives the same bounds.  The tag is a quiet NaN with a
// payload (no arithmetic of this library produces one, so nothing a workspace may hold from earlier kernels matches it),
// handed out by the host, different for every launch and phase.  The second meeting point is one-sided: only block 0
// waits, folds the sums in fp64 and updates the state with the libm-grade finalize of the multi-launch path.
// A block that does not see its peers within `spin_limit` polls gives up - no hang - and the call FAILS as a whole: the
// block marks its pass-2 record, block 0 (which also fails when its own wait runs out) then leaves state9 exactly as it
// was, sets the workspace's fault word (mi_isp_workspace_check) and stores to the device's host-mapped mailbox
// (mi_isp_metering_faults).  Bounds folded from the records that happened to be there must never be blended into the
// rolling state: a moving average would carry them through every later call.
// ---------------------------------------------------------------------------------------------
constexpr int METER_MAX_BLOCKS = 256;
constexpr int METER_THREADS = 1024;
struct MeterFused {
  ew::PtrList imgs;
  int H, W, stride, bpi, n_images;
  float* fp;                 // FrameParams of the workspace (FP_LO / FP_HI are left as the multi-launch path leaves them)
  float* rec0;               // n_blocks records of 16 bytes: {min, max, -, tag}
  float* rec1;               // n_blocks records of 48 bytes: {gmin, gmax, slog2, tag} {sgray, s0, s1, tag} {s2, -, -, tag}
  float* state9;             // receives the new state ...
  const float* prev9;        // ... blended with this one (may be state9 itself: in place)
  float alpha, n_px;
  uint32_t tag;              // phase 0; phase 1 = tag + 1
  unsigned spin_limit;
  unsigned* fault;           // the workspace's fault word (mi_isp_workspace_check)
  unsigned* mailbox;         // host-mapped word of the device (mi_isp_metering_faults): seen without a synchronisation
};

template <class T>
__global__ __launch_bounds__(METER_THREADS) void metering_fused_kernel(const MeterFused a) {
#pragma clang fp contract(fast)
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  __shared__ float red[METER_THREADS / 64][8];
  __shared__ float sh_b[2];
  __shared__ unsigned sh_bad;              // this block gave up waiting for the bounds (its statistics are garbage)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) sh_bad = 0u;
  // grid = (blocks per image, images): the pointer list is indexed by blockIdx.y itself (a computed index sends the whole
  // argument struct through scratch)
  const int n_blocks = a.bpi * a.n_images, part = blockIdx.x, block = blockIdx.y * a.bpi + part;
  const T* img = static_cast<const T*>(a.imgs.p[blockIdx.y]);
  const int hs = (a.H + a.stride - 1) / a.stride, ws = (a.W + a.stride - 1) / a.stride;
  const int n = hs * ws;
  auto pixel = [&](int i, float (&x)[3]) {
    const int sr = i / ws, sc = i - sr * ws;
    const T* p = img + ((size_t)(sr * a.stride) * a.W + (size_t)sc * a.stride) * 3;
    x[0] = (float)p[0]; x[1] = (float)p[1]; x[2] = (float)p[2];
  };
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(a.rec0, 0, n_blocks * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(a.rec1, 0, n_blocks * 48, 0x00020000);

#ifdef MI_METER_STAMPS
  unsigned long long* stamps_ = reinterpret_cast<unsigned long long*>(a.rec1 + 3 * 1024 * 4) + (size_t)block * 8;
#define MI_METER_STAMP(i) do { if (threadIdx.x == 0) stamps_[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MI_METER_STAMP(i) do { } while (0)
#endif
  MI_METER_STAMP(0);
  // ---- pass 1: bounds of the samples (camera_isp.py:152-153) ----
  float vmin = __builtin_inff(), vmax = -__builtin_inff();
  // (the gather is bound by latency: four independent samples in flight per thread)
  const int step = a.bpi * METER_THREADS;
  for (int i0 = part * METER_THREADS + threadIdx.x; i0 < n; i0 += 4 * step) {
    float x[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) pixel(i0 + u * step < n ? i0 + u * step : i0, x[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {                        // (a repeated sample changes neither bound)
      vmin = fminf(vmin, fminf(x[u][0], fminf(x[u][1], x[u][2])));
      vmax = fmaxf(vmax, fmaxf(x[u][0], fmaxf(x[u][1], x[u][2])));
    }
  }
  vmin = wave_min(vmin); vmax = wave_max(vmax);
  if (lane == 0) { red[wave][0] = vmin; red[wave][1] = vmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float lo = red[0][0], hi = red[0][1];
    for (int w = 1; w < METER_THREADS / 64; ++w) { lo = fminf(lo, red[w][0]); hi = fmaxf(hi, red[w][1]); }
    const u4 mine = {__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi), 0u, a.tag};
    __builtin_amdgcn_raw_buffer_store_b128(mine, r0, (uint32_t)block * 16u, 0, 16);     // sc1: write-through
  }
  MI_METER_STAMP(1);
  // ---- every block: all records of pass 1 (wave 0 polls, four records per lane; the other waves wait at the barrier) ----
  if (wave == 0) {
    uint32_t rx[4] = {0u, 0u, 0u, 0u}, ry[4] = {0u, 0u, 0u, 0u};   // (copied out where they are loaded: hipcc 7.2 read .x for .y
    bool have[4];                                                  // of a vector carried round the loop)
#pragma unroll
    for (int j = 0; j < 4; ++j) have[j] = j * 64 + lane >= n_blocks;
    unsigned spins = 0;
    for (;;) {
      u4 t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        t[j] = __builtin_amdgcn_raw_buffer_load_b128(r0, have[j] ? 0xFFFFFFFFu : (uint32_t)(j * 64 + lane) * 16u, 0, 16);
      bool all = true;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!have[j] && t[j].w == a.tag) { rx[j] = t[j].x; ry[j] = t[j].y; have[j] = true; }
        all = all && have[j];
      }
      if (__builtin_amdgcn_ballot_w64(!all) == 0) break;
      if (++spins > a.spin_limit) {
        if (lane == 0) sh_bad = 1u;
        break;
      }
    }
    float lo = __builtin_inff(), hi = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j * 64 + lane < n_blocks && have[j]) {
        lo = fminf(lo, __builtin_bit_cast(float, rx[j])); hi = fmaxf(hi, __builtin_bit_cast(float, ry[j]));
      }
    lo = wave_min(lo); hi = wave_max(hi);
    if (lane == 0) {
      // camera_isp.py:156-157: b = lerp(alpha, new, prev) = new + alpha * (prev - new)  (ew::FIN_ISP_BOUNDS)
      sh_b[0] = lo + a.alpha * (a.prev9[0] - lo);
      sh_b[1] = hi + a.alpha * (a.prev9[1] - hi);
      if (block == 0) { a.fp[FP_LO] = sh_b[0]; a.fp[FP_HI] = sh_b[1]; }
    }
  }
  __syncthreads();
  MI_METER_STAMP(2);
  // ---- pass 2: statistics of the samples normalised by the blended bounds (camera_isp.py:117-128) ----
  const float bmin = sh_b[0], dinv = 1.0f / (sh_b[1] - sh_b[0] + 1e-6f);   // camera_isp.py:119
  StatsAcc st; st.init();
  for (int i0 = part * METER_THREADS + threadIdx.x; i0 < n; i0 += 4 * step) {
    float x[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) pixel(i0 + u * step < n ? i0 + u * step : i0, x[u]);
    // (all four samples loaded before the first is consumed: left alone, the compiler sinks each load into its condition
    // and the thread waits for four round trips in a row - pass 2 then took 10 - 17 us against 4 - 6 of pass 1)
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(x[u][0]), "+v"(x[u][1]), "+v"(x[u][2]));
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * step < n) st.add((x[u][0] - bmin) * dinv, (x[u][1] - bmin) * dinv, (x[u][2] - bmin) * dinv);
  }
  MI_METER_STAMP(6);
  {
    const float v7[7] = {wave_min(st.gmin), wave_max(st.gmax), wave_sum(st.slog), wave_sum(st.sgray), wave_sum(st.s0),
                         wave_sum(st.s1), wave_sum(st.s2)};
    __syncthreads();
    MI_METER_STAMP(7);
    if (lane == 0)
      for (int k = 0; k < 7; ++k) red[wave][k] = v7[k];
    __syncthreads();
    if (threadIdx.x < 3) {
      float t[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        float r = red[0][k];
#pragma unroll
        for (int w = 1; w < METER_THREADS / 64; ++w) r = k == 0 ? fminf(r, red[w][k]) : (k == 1 ? fmaxf(r, red[w][k]) : r + red[w][k]);
        t[k] = r;
      }
      // chunk c of the block's record.  Selected, not indexed: a local array indexed by the thread id is "promoted" to LDS
      // slots addressed by the flat thread id, for which the kernel reads the block size from the dispatch packet - in
      // host memory, 13 - 27 us away (measured: the whole kernel took 34 us, 20 of them here).
      const int c = threadIdx.x;
      const float m0 = c == 0 ? t[0] : (c == 1 ? t[3] : t[6]);
      // (chunk 2's second word: 1 = this block never saw the bounds - what it posts here is not to be used)
      const float m1 = c == 0 ? t[1] : (c == 1 ? t[4] : __builtin_bit_cast(float, sh_bad));
      const float m2 = c == 0 ? t[2] : (c == 1 ? t[5] : 0.f);
      const u4 mine = {__builtin_bit_cast(uint32_t, m0), __builtin_bit_cast(uint32_t, m1), __builtin_bit_cast(uint32_t, m2), a.tag + 1u};
      __builtin_amdgcn_raw_buffer_store_b128(mine, r1, (uint32_t)block * 48u + 16u * c, 0, 16);
    }
  }
  MI_METER_STAMP(3);
  if (block != 0) return;
  // ---- block 0, wave 0: all records of pass 2 (four per lane), sums in fp64, the state (ew::FIN_ISP_STATS) ----
  if (wave != 0) return;
  {
    uint32_t w7[4][7];
    bool have[4];
    bool failed = sh_bad != 0u;                              // (block 0's own wait for the bounds)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      have[j] = j * 64 + lane >= n_blocks;
#pragma unroll
      for (int k = 0; k < 7; ++k) w7[j][k] = 0u;
    }
    unsigned spins = 0;
    for (;;) {
      bool all = true;
      u4 t0[4], t1[4], t2[4];                              // all twelve loads of a round in flight together
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t off = have[j] ? 0xFFFFFFFFu : (uint32_t)(j * 64 + lane) * 48u;
        t0[j] = __builtin_amdgcn_raw_buffer_load_b128(r1, off, 0, 16);
        t1[j] = __builtin_amdgcn_raw_buffer_load_b128(r1, off + (have[j] ? 0u : 16u), 0, 16);
        t2[j] = __builtin_amdgcn_raw_buffer_load_b128(r1, off + (have[j] ? 0u : 32u), 0, 16);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!have[j] && t0[j].w == a.tag + 1u && t1[j].w == a.tag + 1u && t2[j].w == a.tag + 1u) {
          w7[j][0] = t0[j].x; w7[j][1] = t0[j].y; w7[j][2] = t0[j].z; w7[j][3] = t1[j].x; w7[j][4] = t1[j].y; w7[j][5] = t1[j].z;
          w7[j][6] = t2[j].x;
          failed = failed || t2[j].y != 0u;
          have[j] = true;
        }
        all = all && have[j];
      }
      if (__builtin_amdgcn_ballot_w64(!all) == 0) break;
      if (++spins > a.spin_limit) { failed = true; break; }
    }
    if (__builtin_amdgcn_ballot_w64(failed) != 0) {          // the state stays what it was; the failure is reported twice
      if (lane == 0) {
        __hip_atomic_store(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.mailbox) __hip_atomic_store(a.mailbox, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (a.state9 != a.prev9 && lane < 9) a.state9[lane] = a.prev9[lane];     // "unchanged" for a state that is written elsewhere
      return;
    }
    float gmin = __builtin_inff(), gmax = -__builtin_inff();
    double sum[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 4; ++j)                          // records in index order within the lane, lanes by the DPP tree: fixed
      if (j * 64 + lane < n_blocks && have[j]) {
        gmin = fminf(gmin, __builtin_bit_cast(float, w7[j][0])); gmax = fmaxf(gmax, __builtin_bit_cast(float, w7[j][1]));
#pragma unroll
        for (int k = 0; k < 5; ++k) sum[k] += (double)__builtin_bit_cast(float, w7[j][2 + k]);
      }
    MI_METER_STAMP(4);
    gmin = wave_min(gmin); gmax = wave_max(gmax);
    double tot[7];
#pragma unroll
    for (int k = 0; k < 5; ++k) tot[2 + k] = wave_sum(sum[k]);
    if (lane == 0) {
      tot[0] = gmin; tot[1] = gmax;
      ew::FinArgs fa = {};
      fa.fp = a.fp; fa.state9 = a.state9; fa.state9_in = a.prev9; fa.bounds_in = sh_b; fa.n_px = a.n_px; fa.alpha = a.alpha;
      ew::finalize_scalars(ew::FIN_ISP_STATS, fa, tot);
    }
    MI_METER_STAMP(5);
  }
}

// ---------------------------------------------------------------------------------------------
// finalize: one block folds the per-block partials (sums in fp64) and thread 0 derives the
// scalars of the next pass with the accurate libm-grade functions.
// ---------------------------------------------------------------------------------------------
constexpr int FIN_THREADS = 1024;
__global__ __launch_bounds__(FIN_THREADS) void finalize_kernel(int mode, const ew::FinArgs a) {
  __shared__ double sh[7][FIN_THREADS / 64];
  __shared__ double tot[7];
  const int nrows = (mode == ew::FIN_STATS || mode == ew::FIN_ISP_STATS || mode == ew::FIN_ISP_SUMS) ? 7 : 2;
  // row 0: min, row 1: max, rows 2..6: sums (fp64).  All rows of an index are loaded together so the
  // (at most 4) iterations carry independent loads instead of one dependent latency per row.
  double acc[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) acc[k] = k == 0 ? (double)__builtin_inff() : (k == 1 ? -(double)__builtin_inff() : 0.0);
  for (int i = threadIdx.x; i < a.nblocks; i += FIN_THREADS) {
    float v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = k < nrows ? a.partials[(size_t)k * a.stride + i] : 0.f;
    acc[0] = fmin(acc[0], (double)v[0]);
    acc[1] = fmax(acc[1], (double)v[1]);
#pragma unroll
    for (int k = 2; k < 7; ++k) acc[k] += (double)v[k];
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    if (k >= nrows) break;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double other = __shfl_xor(acc[k], o, 64);
      acc[k] = k == 0 ? fmin(acc[k], other) : (k == 1 ? fmax(acc[k], other) : acc[k] + other);
    }
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < nrows) {
    const int k = threadIdx.x;
    double r = sh[k][0];
    for (int w = 1; w < FIN_THREADS / 64; ++w)
      r = k == 0 ? fmin(r, sh[k][w]) : (k == 1 ? fmax(r, sh[k][w]) : r + sh[k][w]);
    tot[k] = r;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;

  ew::finalize_scalars(mode, a, tot);
}

// max_out of every image of a batched ISP Reinhard pass 1: block i folds image i's partial maxima
__global__ __launch_bounds__(EW_THREADS) void maxout_batch_kernel(const float* __restrict__ pmax, int nb,
                                                                  float* __restrict__ maxouts) {
  __shared__ float sh[EW_THREADS / 64];
  float m = -__builtin_inff();
  for (int i = threadIdx.x; i < nb; i += EW_THREADS) m = fmaxf(m, pmax[(size_t)blockIdx.x * nb + i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < EW_THREADS / 64; ++w) m = fmaxf(m, sh[w]);
    maxouts[blockIdx.x] = fmaxf(1e-6f, m);                               // camera_isp.py:190,213
  }
}

__global__ void isp_reinhard_prep_kernel(const float* state9, float* fp, float intensity, float ca) {
  isp_reinhard_scalars(state9, fp, intensity, ca);
}

}  // namespace

// =================================================================================================
// internal launch interface
// =================================================================================================
namespace ew {

int finalize(int mode, const FinArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(FIN_THREADS), 0, s, mode, a);
  MI_LAUNCH_CHECK();
  return 0;
}

int isp_reinhard_prep(const float* state9, float* fp, float intensity, float ca, hipStream_t s) {
  hipLaunchKernelGGL(isp_reinhard_prep_kernel, dim3(1), dim3(1), 0, s, state9, fp, intensity, ca);
  MI_LAUNCH_CHECK();
  return 0;
}

template <class TI, class TO>
static int launch_pass_t(int mode, const PassArgs& a, int nblocks, hipStream_t s, int ny = 1) {
#define MI_PASS(M)                                                                              \
  case M:                                                                                       \
    hipLaunchKernelGGL((rgb_pass_kernel<TI, TO, M>), dim3(nblocks, ny), dim3(PASS_THREADS), 0, s, a); \
    break;
  switch (mode) {
    MI_PASS(PM_MINMAX) MI_PASS(PM_STATS) MI_PASS(PM_RH_MINMAX) MI_PASS(PM_RH_STORE)
    MI_PASS(PM_LINEAR_STORE) MI_PASS(PM_ISP_RH_P1) MI_PASS(PM_ISP_RH_P2) MI_PASS(PM_ISP_RH_P2R)
    default: mi_set_error("bad pass mode %d", mode); return 1;
  }
#undef MI_PASS
  MI_LAUNCH_CHECK();
  return 0;
}

// number of blocks an RGB pass over n_px pixels uses (== number of partials it leaves)
// reduce_only: the statistics / bounds passes of the stateless chain run one block per CU (measured on a 4K
// frame, blocks -> us for pass 1 / pass 2: 256 -> 16.0 / 21.2, 512 -> 16.7 / 22.6, 1024 -> 18.4 / 27.0: every
// block pays the prologue that folds the previous pass's partials); the storing passes two per CU
static int pass_blocks(int64_t n_px, int cap, bool reduce_only = false) {
  int64_t groups = (n_px + 7) / 8;
  int64_t b = (groups + PASS_THREADS - 1) / PASS_THREADS;
  if (b < 1) b = 1;
#ifdef MI_ISP_MEASURE
  static const int env_limit = getenv("MI_ISP_PASS_BLOCKS") ? atoi(getenv("MI_ISP_PASS_BLOCKS")) : 0;   // measurement aid
#else
  constexpr int env_limit = 0;
#endif
  const int limit = env_limit > 0 ? env_limit : (reduce_only ? 256 : 512);
  if (b > limit) b = limit;
  if (b > PASS_MAX_BLOCKS) b = PASS_MAX_BLOCKS;
  if (b > cap) b = cap;
  return (int)b;
}

static int launch_pass(int mode, int in_dtype, int out_dtype, const PassArgs& a, int nblocks, hipStream_t s, int ny = 1) {
  // instantiate the (in, out) pairs the API can produce: reductions ignore TO
  const bool reduce_only = mode == PM_MINMAX || mode == PM_STATS || mode == PM_RH_MINMAX || mode == PM_ISP_RH_P1;
  if (reduce_only) out_dtype = MI_U8;
  switch (in_dtype) {
#define MI_IN(DT, TI)                                                                    \
  case DT:                                                                               \
    switch (out_dtype) {                                                                 \
      case MI_U8: return launch_pass_t<TI, uint8_t>(mode, a, nblocks, s, ny);                \
      case MI_U16: return launch_pass_t<TI, uint16_t>(mode, a, nblocks, s, ny);              \
      case MI_F16: return launch_pass_t<TI, half_t>(mode, a, nblocks, s, ny);                \
      default: return launch_pass_t<TI, float>(mode, a, nblocks, s, ny);                     \
    }
    MI_IN(MI_U8, uint8_t)
    MI_IN(MI_U16, uint16_t)
    MI_IN(MI_F16, half_t)
    default:
      switch (out_dtype) {
        case MI_U8: return launch_pass_t<float, uint8_t>(mode, a, nblocks, s, ny);
        case MI_U16: return launch_pass_t<float, uint16_t>(mode, a, nblocks, s, ny);
        case MI_F16: return launch_pass_t<float, half_t>(mode, a, nblocks, s, ny);
        default: return launch_pass_t<float, float>(mode, a, nblocks, s, ny);
      }
#undef MI_IN
  }
}

}  // namespace ew

// forward: defined with the C ABI below
static bool vec_ok(const void* p, int dtype);

namespace ew {
int subsample(const void* img, void* sub, int H, int W, int stride, int dtype, hipStream_t s) {
  const int hs = (H + stride - 1) / stride, ws = (W + stride - 1) / stride;
  int blocks = (hs * ws + EW_THREADS - 1) / EW_THREADS;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  return dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((subsample_kernel<T>), dim3(blocks), dim3(EW_THREADS), 0, s, static_cast<const T*>(img), static_cast<T*>(sub), H, W, stride);
    MI_LAUNCH_CHECK();
    return 0;
  });
}

int tail_blocks(int H, int W) { return pass_blocks((int64_t)H * W, mi_partial_cap(H, W)); }
// The three data passes after the bounds pass of tonemap.py:146-154, chained by pulled finalizes
// (PassArgs::pull_mode): pass 1 folds the bounds partials `bounds` of whoever produced the image,
// pass 2 the statistics partials of pass 1, pass 3 the bounds partials of pass 2.  Workspace rows:
// [0, 2 cap) the caller's bounds partials, [2 cap, 6 cap) pass 1's 7 rows, [6 cap, 7 cap) pass 2's 2 rows
// (row stride TAIL_STRIDE).  which: -1 = all three, 1..3 = only that pass (measurement aid; the
// partials of an earlier full run must still be in the workspace).
int tonemap_reinhard_tail(const void* src, void* dst, int H, int W, int in_dtype, int out_dtype, float gamma,
                          float intensity, float la, float ca, float* ws, int which, const PullSrc& bounds,
                          hipStream_t s) {
  constexpr int TAIL_STRIDE = PASS_MAX_BLOCKS;    // the block limit of pass_blocks
  float* fp = ws;
  float* partials = fp + FP_COUNT;
  const int cap = mi_partial_cap(H, W);           // >= 4096: 7 rows fit in 4 cap, 2 rows in cap
  float* stats_part = partials + 2 * (size_t)cap;
  float* bounds2_part = partials + 6 * (size_t)cap;
  PassArgs a = {};
  a.src = src; a.dst = dst; a.fp = fp; a.fp_w = fp; a.n_px = (int64_t)H * W;
  a.vec_in = vec_ok(src, in_dtype); a.vec_out = vec_ok(dst, out_dtype);
  a.gamma_inv = 1.0f / gamma; a.la = la; a.ca = ca;
  a.out_scale = mi_scale_factor(out_dtype); a.transform = MI_T_NONE; a.H = H; a.W = W;
  a.pull_npx = (float)a.n_px; a.pull_intensity = intensity;
  a.no_nan = bounds.bounds_post > 0;               // the fused pipeline's image (clamped by the tile kernel)
  const int nb = pass_blocks(a.n_px, cap), nbr = pass_blocks(a.n_px, cap, true);
  // measurement aid: MI_ISP_PULL_DEBUG=0 runs single passes (which > 0) on the scalars a full run left
  // in FrameParams, without the pulled finalize
#ifdef MI_ISP_MEASURE
  static const char* dbg = getenv("MI_ISP_PULL_DEBUG");
#else
  const char* dbg = nullptr;
#endif
  const int no_pull = (dbg && which > 0 && dbg[0] == '0') ? -1 : 0;
  const int dbg_bits = (dbg && which > 0 && dbg[0] > '0') ? (dbg[0] - '0') << 8 : 0;
  if (which < 0 || which == 1) {                                                      // tonemap.py:147-149
    PassArgs p1 = a;
    p1.partials = stats_part; p1.part_stride = TAIL_STRIDE;
    p1.pull_mode = no_pull ? -1 : FIN_BOUNDS | dbg_bits; p1.pull_partials = bounds.partials; p1.pull_stride = bounds.stride;
    p1.pull_n = (dbg_bits & 0x200) ? 0 : bounds.n; p1.pull_bounds_post = bounds.bounds_post;
    if (int rc = launch_pass(PM_STATS, in_dtype, out_dtype, p1, nbr, s)) return rc;
  }
  if (which < 0 || which == 2) {                                                      // :150,153
    PassArgs p2 = a;
    p2.partials = bounds2_part; p2.part_stride = TAIL_STRIDE;
    p2.pull_mode = no_pull ? -1 : FIN_STATS | dbg_bits; p2.pull_partials = stats_part; p2.pull_stride = TAIL_STRIDE; p2.pull_n = (dbg_bits & 0x200) ? 0 : nbr;
    if (int rc = launch_pass(PM_RH_MINMAX, in_dtype, out_dtype, p2, nbr, s)) return rc;
  }
  if (which < 0 || which == 3) {                                                      // :154
    PassArgs p3 = a;
    p3.pull_mode = no_pull ? -1 : FIN_BOUNDS2 | dbg_bits; p3.pull_partials = bounds2_part; p3.pull_stride = TAIL_STRIDE; p3.pull_n = (dbg_bits & 0x200) ? 0 : nbr;
    if (int rc = launch_pass(PM_RH_STORE, in_dtype, out_dtype, p3, nb, s)) return rc;
  }
  return 0;
}
}  // namespace ew

// =================================================================================================
// C ABI
// =================================================================================================
using namespace ew;

extern "C" int mi_isp_decode12(const uint8_t* enc, void* out, int64_t n_px, int out_dtype, int scaled,
                               int ids_format, void* stream) {
  MI_REQUIRE(n_px >= 0 && n_px % 2 == 0, "decode12: pixel count must be even, got %lld", (long long)n_px);
  MI_REQUIRE(mi_valid_dtype(out_dtype), "decode12: bad dtype %d", out_dtype);
  if (n_px == 0) return 0;
  MI_REQUIRE(enc && out, "decode12: null pointer");
  const float k = (float)((double)mi_scale_factor(out_dtype) / 4095.0);
  const int fast = mi_aligned(enc, 4) && mi_aligned(out, 16);
  const int64_t n_pairs = n_px / 2;
  hipStream_t s = (hipStream_t)stream;
  return dispatch_dtype(out_dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((decode12_kernel<T>), dim3(grid_for((n_pairs + 3) / 4)), dim3(EW_THREADS), 0, s, enc,
                       static_cast<T*>(out), n_pairs, scaled, ids_format, k, fast);
    MI_LAUNCH_CHECK();
    return 0;
  });
}

extern "C" int mi_isp_decode16(const uint8_t* enc, void* out, int64_t n_px, int out_dtype, int scaled,
                               void* stream) {
  MI_REQUIRE(n_px >= 0, "decode16: negative count");
  MI_REQUIRE(mi_valid_dtype(out_dtype), "decode16: bad dtype %d", out_dtype);
  if (n_px == 0) return 0;
  MI_REQUIRE(enc && out, "decode16: null pointer");
  const float k = (float)((double)mi_scale_factor(out_dtype) / 65535.0);
  const int fast = mi_aligned(enc, 16) && mi_aligned(out, 16);
  hipStream_t s = (hipStream_t)stream;
  return dispatch_dtype(out_dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((decode16_kernel<T>), dim3(grid_for((n_px + 7) / 8)), dim3(EW_THREADS), 0, s, enc,
                       static_cast<T*>(out), n_px, scaled, k, fast);
    MI_LAUNCH_CHECK();
    return 0;
  });
}

extern "C" int mi_isp_encode12(const void* values, uint8_t* enc, int64_t n_px, int in_dtype, int scaled,
                               int ids_format, void* stream) {
  MI_REQUIRE(n_px >= 0 && n_px % 2 == 0, "encode12: pixel count must be even, got %lld", (long long)n_px);
  MI_REQUIRE(mi_valid_dtype(in_dtype), "encode12: bad dtype %d", in_dtype);
  if (n_px == 0) return 0;
  MI_REQUIRE(values && enc, "encode12: null pointer");
  const float k = (float)(4095.0 / (double)mi_scale_factor(in_dtype));
  hipStream_t s = (hipStream_t)stream;
  return dispatch_dtype(in_dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((encode12_kernel<T>), dim3(grid_for(n_px / 2)), dim3(EW_THREADS), 0, s,
                       static_cast<const T*>(values), enc, n_px / 2, scaled, ids_format, k);
    MI_LAUNCH_CHECK();
    return 0;
  });
}

extern "C" int mi_isp_load_convert(const void* src, void* dst, int64_t n, int mode, int out_dtype, void* stream) {
  MI_REQUIRE(src && dst, "load_convert: null pointer");
  MI_REQUIRE(mode >= MI_LOAD_16U && mode <= MI_LOAD_16F, "load_convert: bad mode %d", mode);
  MI_REQUIRE(out_dtype == MI_F16 || out_dtype == MI_F32, "load_convert: output must be f16/f32");
  if (n <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  return dispatch_dtype(out_dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((load_convert_kernel<T>), dim3(grid_for(n)), dim3(EW_THREADS), 0, s, src,
                       static_cast<T*>(dst), n, mode);
    MI_LAUNCH_CHECK();
    return 0;
  });
}

extern "C" int mi_isp_mosaic(const void* rgb, void* cfa, int H, int W, int dtype, int pattern, void* stream) {
  MI_REQUIRE(rgb && cfa, "mosaic: null pointer");
  MI_REQUIRE(H > 0 && W > 0, "mosaic: bad shape %dx%d", H, W);
  MI_REQUIRE(mi_valid_dtype(dtype), "mosaic: bad dtype %d", dtype);
  MI_REQUIRE(pattern >= 0 && pattern <= 3, "mosaic: bad pattern %d", pattern);
  // bayer.py:85-90 pixel_orders as (r0c0, r0c1, r1c0, r1c1), 2 bits each
  static const int orders[4][4] = {{0, 1, 1, 2}, {1, 0, 2, 1}, {1, 2, 0, 1}, {2, 1, 1, 0}};
  int order4 = 0;
  for (int i = 0; i < 4; ++i) order4 |= orders[pattern][i] << (2 * i);
  hipStream_t s = (hipStream_t)stream;
  // the reference loops over whole 2x2 quads only (bayer.py:106): odd trailing row/col untouched
  const int He = H & ~1, We = W & ~1;
  MI_REQUIRE(He == H && We == W, "mosaic: image must be even size, got %dx%d", H, W);
  return dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((mosaic_kernel<T>), dim3(grid_for((int64_t)H * W)), dim3(EW_THREADS), 0, s,
                       static_cast<const T*>(rgb), static_cast<T*>(cfa), H, W, order4);
    MI_LAUNCH_CHECK();
    return 0;
  });
}

template <class F> static int dispatch_dtype2(int in_dtype, int out_dtype, F&& f) {
  return dispatch_dtype(in_dtype, [&](auto ti_tag) {
    return dispatch_dtype(out_dtype, [&](auto to_tag) { return f(ti_tag, to_tag); });
  });
}

extern "C" int mi_isp_rgb_to_yuv420(const void* rgb, void* yuv, int H, int W, int in_dtype, int out_dtype,
                                    void* stream) {
  MI_REQUIRE(rgb && yuv, "rgb_to_yuv420: null pointer");
  MI_REQUIRE(H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "rgb_to_yuv420: image must be even size, got %dx%d", H, W);
  MI_REQUIRE(mi_valid_dtype(in_dtype) && mi_valid_dtype(out_dtype), "rgb_to_yuv420: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  return dispatch_dtype2(in_dtype, out_dtype, [&](auto ti_tag, auto to_tag) {
    using TI = decltype(ti_tag);
    using TO = decltype(to_tag);
    // vector path: whole 8-pixel groups, and every row of every plane 16-byte aligned
    const bool vec = W % 16 == 0 && ((uintptr_t)rgb & 15) == 0 && ((uintptr_t)yuv & 15) == 0 &&
                     ((size_t)(H / 2) * (W / 2) * sizeof(TO)) % 16 == 0;
    if (vec)
      hipLaunchKernelGGL((rgb_yuv420_vec_kernel<TI, TO>), dim3(grid_for((int64_t)(H / 2) * (W / 8))), dim3(EW_THREADS), 0,
                         s, static_cast<const TI*>(rgb), static_cast<TO*>(yuv), H, W);
    else
      hipLaunchKernelGGL((rgb_yuv420_kernel<TI, TO>), dim3(grid_for((int64_t)(H / 2) * (W / 2))), dim3(EW_THREADS), 0, s,
                         static_cast<const TI*>(rgb), static_cast<TO*>(yuv), H, W);
    MI_LAUNCH_CHECK();
    return 0;
  });
}

extern "C" int mi_isp_yuv420_to_rgb(const void* yuv, void* rgb, int H, int W, int in_dtype, int out_dtype,
                                    void* stream) {
  MI_REQUIRE(rgb && yuv, "yuv420_to_rgb: null pointer");
  MI_REQUIRE(H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "yuv420_to_rgb: image must be even size, got %dx%d", H, W);
  MI_REQUIRE(mi_valid_dtype(in_dtype) && mi_valid_dtype(out_dtype), "yuv420_to_rgb: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  return dispatch_dtype2(in_dtype, out_dtype, [&](auto ti_tag, auto to_tag) {
    using TI = decltype(ti_tag);
    using TO = decltype(to_tag);
    hipLaunchKernelGGL((yuv420_rgb_kernel<TI, TO>), dim3(grid_for((int64_t)H * W)), dim3(EW_THREADS), 0, s,
                       static_cast<const TI*>(yuv), static_cast<TO*>(rgb), H, W);
    MI_LAUNCH_CHECK();
    return 0;
  });
}

extern "C" int mi_isp_resize_bilinear(const void* src, void* dst, int Hs, int Ws, int Hd, int Wd, float s0,
                                      float s1, int in_dtype, int out_dtype, void* stream) {
  MI_REQUIRE(src && dst, "resize: null pointer");
  MI_REQUIRE(Hs > 0 && Ws > 0 && Hd >= 0 && Wd >= 0, "resize: bad shape");
  MI_REQUIRE(s0 > 0.f && s1 > 0.f, "resize: scale must be positive");
  MI_REQUIRE(mi_valid_dtype(in_dtype) && mi_valid_dtype(out_dtype), "resize: bad dtype");
  if ((int64_t)Hd * Wd == 0) return 0;
  const float intensity = (float)((double)mi_scale_factor(out_dtype) / (double)mi_scale_factor(in_dtype));
  hipStream_t s = (hipStream_t)stream;
  const int g = grid_for((int64_t)Hd * Wd);
  return dispatch_dtype(in_dtype, [&](auto ti) {
    using TI = decltype(ti);
    return dispatch_dtype(out_dtype, [&](auto to) {
      using TO = decltype(to);
      hipLaunchKernelGGL((resize_kernel<TI, TO>), dim3(g), dim3(EW_THREADS), 0, s, static_cast<const TI*>(src),
                         static_cast<TO*>(dst), Hs, Ws, Hd, Wd, s0, s1, intensity);
      MI_LAUNCH_CHECK();
      return 0;
    });
  });
}

extern "C" int mi_isp_transform(const void* src, void* dst, int Hs, int Ws, int dtype, int transform, void* stream) {
  MI_REQUIRE(src && dst, "transform: null pointer");
  MI_REQUIRE(Hs > 0 && Ws > 0, "transform: bad shape");
  MI_REQUIRE(mi_valid_dtype(dtype), "transform: bad dtype");
  MI_REQUIRE(transform >= MI_T_NONE && transform <= MI_T_TRANSVERSE, "transform: bad transform %d", transform);
  MI_REQUIRE(transform != MI_T_TRANSVERSE || Hs == Ws,
             "transform: transverse is only defined for square images (the reference reads out of bounds)");
  const bool swap = transform == MI_T_ROTATE_90 || transform == MI_T_ROTATE_270 || transform == MI_T_TRANSPOSE;
  const int Hd = swap ? Ws : Hs, Wd = swap ? Hs : Ws;
  hipStream_t s = (hipStream_t)stream;
  return dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((transform_kernel<T>), dim3(grid_for((int64_t)Hd * Wd)), dim3(EW_THREADS), 0, s,
                       static_cast<const T*>(src), static_cast<T*>(dst), Hs, Ws, Hd, Wd, transform);
    MI_LAUNCH_CHECK();
    return 0;
  });
}

// ---- metering --------------------------------------------------------------------------------
static int metering_pass(int phase, const void* const* images, int n_images, int H, int W, int stride,
                         int dtype, const float* bounds, float* partials, int cap, int* nblocks_out,
                         hipStream_t s) {
  const int hs = (H + stride - 1) / stride, ws = (W + stride - 1) / stride;
  // the strided gather is latency-bound (one 64-byte line per pixel): few pixels per thread, as many blocks per
  // image as the partial rows allow (16 blocks per image measured 44 us for six 4K images, 192: see DESIGN.md 7)
  int bpi = (hs * ws + EW_THREADS * 4 - 1) / (EW_THREADS * 4);
  const int room = cap / (n_images > 0 ? n_images : 1);
  if (bpi > room) bpi = room;
  if (bpi > 256) bpi = 256;
  if (bpi < 1) bpi = 1;
  MI_REQUIRE((int64_t)bpi * n_images <= cap, "metering: too many images (%d) for the workspace", n_images);
  int base = 0;
  for (int i0 = 0; i0 < n_images; i0 += 64) {
    PtrList pl;
    const int n = n_images - i0 < 64 ? n_images - i0 : 64;
    for (int i = 0; i < 64; ++i) pl.p[i] = i < n ? images[i0 + i] : nullptr;
    const int rc = dispatch_dtype(dtype, [&](auto tag) {
      using T = decltype(tag);
      if (phase == 0)
        hipLaunchKernelGGL((metering_kernel<T, 0>), dim3(bpi, n), dim3(EW_THREADS), 0, s, pl, H, W, stride, bounds,
                           partials, cap, base);
      else
        hipLaunchKernelGGL((metering_kernel<T, 1>), dim3(bpi, n), dim3(EW_THREADS), 0, s, pl, H, W, stride, bounds,
                           partials, cap, base);
      MI_LAUNCH_CHECK();
      return 0;
    });
    if (rc) return rc;
    base += bpi * n;
  }
  *nblocks_out = base;
  return 0;
}

static int metering_check(const void* const* images, int n_images, int H, int W, int stride, int dtype,
                          const void* ws) {
  MI_REQUIRE(images && ws, "metering: null pointer");
  MI_REQUIRE(n_images > 0, "metering: need at least one image");
  for (int i = 0; i < n_images; ++i) MI_REQUIRE(images[i], "metering: image %d is null", i);
  MI_REQUIRE(H > 0 && W > 0 && stride > 0, "metering: bad shape/stride");
  MI_REQUIRE(mi_valid_dtype(dtype), "metering: bad dtype");
  return 0;
}

extern "C" int mi_isp_metering_bounds(const void* const* images, int n_images, int H, int W, int stride, int dtype,
                                      float* out2, void* ws, void* stream) {
  if (int rc = metering_check(images, n_images, H, W, stride, dtype, ws)) return rc;
  MI_REQUIRE(out2, "metering_bounds: null output");
  hipStream_t s = (hipStream_t)stream;
  float* fp = static_cast<float*>(ws);
  float* partials = fp + FP_COUNT;
  const int cap = mi_partial_cap(H, W);
  int nb = 0;
  if (int rc = metering_pass(0, images, n_images, H, W, stride, dtype, nullptr, partials, cap, &nb, s)) return rc;
  FinArgs fa = {};
  fa.partials = partials; fa.stride = cap; fa.nblocks = nb; fa.fp = fp; fa.out = out2;
  return finalize(FIN_RAW_BOUNDS, fa, s);
}

extern "C" int mi_isp_metering_sums(const void* const* images, int n_images, int H, int W, int stride, int dtype,
                                    const float* bounds2, float* out8, void* ws, void* stream) {
  if (int rc = metering_check(images, n_images, H, W, stride, dtype, ws)) return rc;
  MI_REQUIRE(bounds2 && out8, "metering_sums: null pointer");
  hipStream_t s = (hipStream_t)stream;
  float* fp = static_cast<float*>(ws);
  float* partials = fp + FP_COUNT;
  const int cap = mi_partial_cap(H, W);
  int nb = 0;
  if (int rc = metering_pass(1, images, n_images, H, W, stride, dtype, bounds2, partials, cap, &nb, s)) return rc;
  const int hs = (H + stride - 1) / stride, wss = (W + stride - 1) / stride;
  FinArgs fa = {};
  fa.partials = partials; fa.stride = cap; fa.nblocks = nb; fa.fp = fp; fa.out = out8;
  fa.n_px = (float)((int64_t)n_images * hs * wss);
  return finalize(FIN_ISP_SUMS, fa, s);
}

// ---- the sharded batch (one process per GPU): combine the ranks' partials after an all-gather ---------------------------
// Round 1: every rank contributes its raw [min, max]; round 2: its [log_min, log_max, sum_log, sum_gray, sum_r, sum_g,
// sum_b, n].  Each round is ONE all-gather plus one of these single-thread kernels on the compute stream; the arithmetic
// is the single-GPU finalize's (FIN_ISP_BOUNDS / FIN_ISP_STATS), so a sharded batch ends with the state an unsharded one
// would have (up to the summation order of the partial sums).
namespace {
__global__ void metering_combine_bounds_kernel(const float* __restrict__ gathered, int n_ranks, const float* state9, float alpha,
                                               float* bounds2) {
  float lo = __builtin_inff(), hi = -__builtin_inff();
  for (int r = 0; r < n_ranks; ++r) { lo = fminf(lo, gathered[2 * r]); hi = fmaxf(hi, gathered[2 * r + 1]); }
  bounds2[0] = lo + alpha * (state9[0] - lo);          // camera_isp.py:156-157
  bounds2[1] = hi + alpha * (state9[1] - hi);
}
__global__ void metering_combine_sums_kernel(const float* __restrict__ gathered, int n_ranks, const float* bounds2, float* state9,
                                             float alpha) {
  float lmin = __builtin_inff(), lmax = -__builtin_inff();
  double sum[5] = {0, 0, 0, 0, 0}, n = 0;
  for (int r = 0; r < n_ranks; ++r) {
    const float* g = gathered + 8 * r;
    lmin = fminf(lmin, g[0]); lmax = fmaxf(lmax, g[1]);
    for (int k = 0; k < 5; ++k) sum[k] += (double)g[2 + k];
    n += (double)g[7];
  }
  const float nn = (float)n;                            // camera_isp.py:131-134,164-166
  const float v[9] = {bounds2[0], bounds2[1], lmin, lmax, (float)sum[0] / nn, (float)sum[1] / nn,
                      (float)sum[2] / nn, (float)sum[3] / nn, (float)sum[4] / nn};
  for (int i = 0; i < 9; ++i) state9[i] = v[i] + alpha * (state9[i] - v[i]);
}
}  // namespace

extern "C" int mi_isp_metering_combine_bounds(const float* gathered, int n_ranks, const float* state9, float alpha,
                                              float* bounds2_out, void* stream) {
  MI_REQUIRE(gathered && state9 && bounds2_out && n_ranks >= 1, "metering_combine_bounds: bad arguments");
  hipLaunchKernelGGL(metering_combine_bounds_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, gathered, n_ranks, state9, alpha,
                     bounds2_out);
  MI_LAUNCH_CHECK();
  return 0;
}

extern "C" int mi_isp_metering_combine_sums(const float* gathered, int n_ranks, const float* bounds2, float* state9, float alpha,
                                            void* stream) {
  MI_REQUIRE(gathered && bounds2 && state9 && n_ranks >= 1, "metering_combine_sums: bad arguments");
  hipLaunchKernelGGL(metering_combine_sums_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, gathered, n_ranks, bounds2, state9,
                     alpha);
  MI_LAUNCH_CHECK();
  return 0;
}

// One launch (metering_fused_kernel) when the group's blocks fit one per CU; MI_ISP_METERING_LAUNCHES=4 in the environment
// forces the four-launch path (measurement, and the reference the fused path is tested against).
static std::atomic<uint32_t> g_meter_launches{0};
// Its launches take part in the one order of the library's resident grids (ew::resident_order, isp_elementwise.h).
static std::atomic<unsigned> g_meter_poll_limit{0};          // 0 = default; tests: mi_isp_metering_set_poll_limit
static int metering_fused(const void* const* images, int n_images, int H, int W, int stride, int dtype, const float* prev9,
                          float* state9, float alpha, float* fp, float* partials, int cap, hipStream_t s, bool* done) {
  *done = false;
  const char* env = getenv("MI_ISP_METERING_LAUNCHES");       // (read per call: a test switches it)
  if ((env && atoi(env) == 4) || n_images > 64 || n_images > METER_MAX_BLOCKS) return 0;
  int dev = 0, n_cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 0;
  if (dev < 0 || dev >= 16) return 0;
  hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap_status);
  if (cap_status != hipStreamCaptureStatusNone) return 0;     // (a capture cannot be put in order: the four launches)
  const int max_blocks = n_cus < METER_MAX_BLOCKS ? n_cus : METER_MAX_BLOCKS;
  if (n_images > max_blocks || cap < 4 * METER_MAX_BLOCKS) return 0;
  const int hs = (H + stride - 1) / stride, wss = (W + stride - 1) / stride;
  int bpi = (hs * wss + METER_THREADS * 2 - 1) / (METER_THREADS * 2);
  if (bpi > max_blocks / n_images) bpi = max_blocks / n_images;
  if (bpi < 1) bpi = 1;
  MeterFused a = {};
  for (int i = 0; i < 64; ++i) a.imgs.p[i] = i < n_images ? images[i] : nullptr;
  a.H = H; a.W = W; a.stride = stride; a.bpi = bpi; a.n_images = n_images;
  a.fp = fp; a.rec0 = partials; a.rec1 = partials + cap;       // partial rows 0 and 1..3 (cap >= 1024 floats each)
  a.state9 = state9; a.prev9 = prev9; a.alpha = alpha; a.n_px = (float)((int64_t)n_images * hs * wss);
  // a quiet NaN with a payload, two per launch (0x7FC00001 ...): see the kernel's head
  const uint32_t k = g_meter_launches.fetch_add(1, std::memory_order_relaxed);
  a.tag = 0x7FC00001u + 2u * (k % 0x1FFFFFu);
  const unsigned limit = g_meter_poll_limit.load(std::memory_order_relaxed);
  a.spin_limit = limit ? limit - 1u : 2000000u;                // ~1 s of polling (a test's limit of 1: no second round)
  a.fault = reinterpret_cast<unsigned*>(fp) + 62;              // FP_ERROR (isp_mega.h): what mi_isp_workspace_check reads
  ew::ResidentOrder& ord = ew::resident_order();
  std::lock_guard<std::mutex> lock(ord.mu);
  if (int rc = ew::resident_enter_locked(dev, s)) return rc;
  a.mailbox = ord.mailbox_dev[dev] + ew::MAILBOX_METERING;
  const int rc = dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((metering_fused_kernel<T>), dim3(bpi, n_images), dim3(METER_THREADS), 0, s, a);
    MI_LAUNCH_CHECK();
    return 0;
  });
  if (rc) return rc;
  if (int rc2 = ew::resident_leave_locked(dev, s)) return rc2;
  *done = true;
  return 0;
}

extern "C" int mi_isp_metering_set_poll_limit(unsigned polls) {
  g_meter_poll_limit.store(polls, std::memory_order_relaxed);
  return 0;
}

// The metering mailbox of the current device: non-zero when a one-launch update_metering gave up waiting for its
// blocks since the word was last cleared (its state9 was left untouched).  A plain host read, no synchronisation.
extern "C" int mi_isp_metering_faults(int clear) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  ew::ResidentOrder& ord = ew::resident_order();
  std::lock_guard<std::mutex> lock(ord.mu);
  if (!ord.mailbox_host[dev]) return 0;
  volatile unsigned* mb = ord.mailbox_host[dev] + ew::MAILBOX_METERING;
  const unsigned v = *mb;
  if (clear) *mb = 0;
  return (int)v;
}

namespace ew {
ResidentOrder& resident_order() {
  static ResidentOrder o;
  return o;
}
int resident_mailbox_locked(int dev) {
  ResidentOrder& o = resident_order();
  if (!o.mailbox_host[dev]) {
    void* h = nullptr;
    MI_HIP(hipHostMalloc(&h, 64, hipHostMallocMapped));
    memset(h, 0, 64);
    void* d = nullptr;
    MI_HIP(hipHostGetDevicePointer(&d, h, 0));
    o.mailbox_host[dev] = static_cast<unsigned*>(h);
    o.mailbox_dev[dev] = static_cast<unsigned*>(d);
  }
  return 0;
}
int resident_enter_locked(int dev, hipStream_t s) {
  ResidentOrder& o = resident_order();
  if (int rc = resident_mailbox_locked(dev)) return rc;
  if (o.has_last[dev] && o.last[dev] != s) MI_HIP(hipStreamWaitEvent(s, o.done[dev], 0));
  return 0;
}
int resident_leave_locked(int dev, hipStream_t s) {
  ResidentOrder& o = resident_order();
  if (!o.done[dev]) MI_HIP(hipEventCreateWithFlags(&o.done[dev], hipEventDisableTiming));
  MI_HIP(hipEventRecord(o.done[dev], s));
  o.last[dev] = s; o.has_last[dev] = true;
  return 0;
}
}  // namespace ew

// prev9 -> state9 (camera_isp.py:172-173: `metering = prev.clone()`, then the kernel updates the clone - here without the
// copy: the previous state is only read, the new one only written; prev9 == state9 is the in-place form)
extern "C" int mi_isp_metering_to(const void* const* images, int n_images, int H, int W, int stride, int dtype,
                                  const float* prev9, float* state9, float alpha, void* ws, void* stream) {
  if (int rc = metering_check(images, n_images, H, W, stride, dtype, ws)) return rc;
  MI_REQUIRE(state9 && prev9, "metering: null state");
  hipStream_t s = (hipStream_t)stream;
  float* fp = static_cast<float*>(ws);
  float* partials = fp + FP_COUNT;
  const int cap = mi_partial_cap(H, W);
  {
    bool done = false;
    if (int rc = metering_fused(images, n_images, H, W, stride, dtype, prev9, state9, alpha, fp, partials, cap, s, &done)) return rc;
    if (done) return 0;
  }
  int nb = 0;
  if (int rc = metering_pass(0, images, n_images, H, W, stride, dtype, nullptr, partials, cap, &nb, s)) return rc;
  FinArgs fa = {};
  fa.partials = partials; fa.stride = cap; fa.nblocks = nb; fa.fp = fp; fa.state9 = state9; fa.state9_in = prev9; fa.alpha = alpha;
  if (int rc = finalize(FIN_ISP_BOUNDS, fa, s)) return rc;
  if (int rc = metering_pass(1, images, n_images, H, W, stride, dtype, fp + FP_LO, partials, cap, &nb, s)) return rc;
  const int hs = (H + stride - 1) / stride, wss = (W + stride - 1) / stride;
  fa.nblocks = nb;
  fa.n_px = (float)((int64_t)n_images * hs * wss);
  return finalize(FIN_ISP_STATS, fa, s);
}

extern "C" int mi_isp_metering(const void* const* images, int n_images, int H, int W, int stride, int dtype,
                               float* state9, float alpha, void* ws, void* stream) {
  return mi_isp_metering_to(images, n_images, H, W, stride, dtype, state9, state9, alpha, ws, stream);
}

// ---- ISP tonemaps ----------------------------------------------------------------------------
static bool vec_ok(const void* p, int dtype) { return mi_aligned(p, dtype == MI_U8 ? 8 : 16); }  // 24-element groups

extern "C" int mi_isp_reinhard(void* image, uint8_t* out, int H, int W, int dtype, const float* state9, float gamma,
                               float intensity, float light_adapt, float color_adapt, int transform, void* ws,
                               void* stream) {
  MI_REQUIRE(image && out && state9 && ws, "reinhard: null pointer");
  MI_REQUIRE(H > 0 && W > 0, "reinhard: bad shape");
  MI_REQUIRE(dtype == MI_F16 || dtype == MI_F32, "reinhard: image must be f16 or f32");
  MI_REQUIRE(gamma > 0.f, "reinhard: gamma must be positive");
  MI_REQUIRE(transform >= MI_T_NONE && transform <= MI_T_TRANSVERSE, "reinhard: bad transform");
  MI_REQUIRE(transform != MI_T_TRANSVERSE || H == W, "reinhard: transverse needs a square image");
  hipStream_t s = (hipStream_t)stream;
  float* fp = static_cast<float*>(ws);
  float* partials = fp + FP_COUNT;
  const int cap = mi_partial_cap(H, W);
  if (int rc = isp_reinhard_prep(state9, fp, intensity, color_adapt, s)) return rc;
  PassArgs a = {};
  a.src = image; a.inplace = image; a.dst = out; a.fp = fp; a.partials = partials; a.part_stride = cap;
  a.n_px = (int64_t)H * W; a.vec_in = vec_ok(image, dtype); a.vec_out = vec_ok(out, MI_U8);
  a.gamma_inv = (float)(1.0 / (double)gamma); a.la = light_adapt; a.ca = color_adapt; a.out_scale = 255.f;
  a.transform = transform; a.H = H; a.W = W;
  const int nb = pass_blocks(a.n_px, cap);
  if (int rc = launch_pass(PM_ISP_RH_P1, dtype, MI_U8, a, nb, s)) return rc;
  FinArgs fa = {};
  fa.partials = partials; fa.stride = cap; fa.nblocks = nb; fa.fp = fp;
  if (int rc = finalize(FIN_MAXOUT, fa, s)) return rc;
  return launch_pass(PM_ISP_RH_P2, dtype, MI_U8, a, nb, s);
}

// mi_isp_reinhard_batch through isp_reinhard_fused_kernel when the group fits: no orientation transform, aligned whole
// groups, every image's p resident (<= MAXIT groups per thread).  *done = false: the caller takes the two-pass path.
static std::atomic<unsigned> g_ispf_poll_limit{0};
static struct { unsigned* buf[16] = {}; unsigned launches[16] = {}; int bpc[16] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}; } g_ispf;
template <class TI, bool CA0, int MAXIT, bool PIPE, int BPC, int NLDS = 0>
static int ispf_launch(const IspFusedArgs& a, int nblocks, int slot, hipStream_t s, bool* ok) {
  if (g_ispf.bpc[slot] < 0) {
    int per_cu = 0;
    MI_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, isp_reinhard_fused_kernel<TI, CA0, MAXIT, PIPE, BPC, NLDS>, ISPF_THREADS, 0));
    g_ispf.bpc[slot] = per_cu;
  }
  *ok = g_ispf.bpc[slot] >= BPC;                             // every block resident, or not at all
  if (!*ok) return 0;
  hipLaunchKernelGGL((isp_reinhard_fused_kernel<TI, CA0, MAXIT, PIPE, BPC, NLDS>), dim3(nblocks), dim3(ISPF_THREADS), 0, s, a);
  MI_LAUNCH_CHECK();
  return 0;
}
static int isp_reinhard_fused(void* const* images, uint8_t* const* outs, int m, int H, int W, int dtype, const float* state9,
                              float gamma, float intensity, float la, float ca, int transform, float* fp, hipStream_t s,
                              bool* done) {
  *done = false;
  // MEASURED (round 4, 6 x 1440 x 1920 f16, gamma 0.6): bit-identical to the two passes and SLOWER - 0.267 - 0.272 ms per
  // 6-camera step against 0.257 (scripts/time_isp.py).  The two passes move 350 MB at 5.5 TB/s: they are at the memory's
  // practical rate, and the 100 MB this kernel saves do not pay for what it loses - 11 transcendentals per pixel in
  // dependent chains on 8 - 16 waves per CU instead of 32, two or three groups per thread and image (nothing to pipeline
  // within an image), and a block-wide hand-over per image.  So it is OFF unless MI_ISP_REINHARD_LAUNCHES=1 asks for it
  // (read per call; tests/ compare the two paths bit for bit).
  const char* env = getenv("MI_ISP_REINHARD_LAUNCHES");
  const int forced = env ? atoi(env) : 0;
  if (forced == 2) return 0;
  if (transform != MI_T_NONE || m < 1 || m > ISPF_MAX_IMAGES) return 0;
  const int64_t n_px = (int64_t)H * W;
  if (n_px % 512 != 0) return 0;                               // whole waves of whole groups only
  for (int i = 0; i < m; ++i)
    if (!vec_ok(images[i], dtype) || !mi_aligned(outs[i], 16)) return 0;
  int dev = 0, n_cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus < 1) return 0;
  hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap_status);
  if (cap_status != hipStreamCaptureStatusNone) return 0;      // (a capture cannot be put in order)
  const int64_t n_groups = n_px / 8;
  const bool f16 = dtype == MI_F16;
  // (Four blocks per CU - 16 waves, <= 128 VGPRs, two kept groups per image and thread - measured no faster: 0.272 against
  // 0.267 ms per step, and it spilled 9 - 13 registers once the next image's first group was prefetched.  Taken out.)
  int nblocks = n_cus * ISPF_BPC;
  if ((int64_t)nblocks * ISPF_THREADS > n_groups) nblocks = (int)(n_groups / ISPF_THREADS) > 0 ? (int)(n_groups / ISPF_THREADS) : 1;
  const int iters = (int)((n_groups + (int64_t)nblocks * ISPF_THREADS - 1) / ((int64_t)nblocks * ISPF_THREADS));
  const bool pipe = iters <= 3 && m > 1;
  // LARGE images (7 - 12 groups per thread, f16: a 4096 x 3072 frame has 12): 7 groups in registers, 5 in LDS, one image at a
  // time.  MEASURED (six 4K cameras): bit-identical and much SLOWER - 0.53 - 0.54 ms per step against 0.44 - 0.45.  A frame's p is
  // 295 KB per CU; what is left of the register file next to it holds ONE 3 KB group in flight per wave, 8 waves per CU: 6 MB
  // in flight on the whole chip, where 6.5 TB/s x 2 us of latency want 13.  The two passes run 32 waves per CU.  Opt-in too.
  const bool large = f16 && !pipe && iters > 6 && iters <= 12;
  if (forced != 1) return 0;
  if (!pipe && !large && iters > (f16 ? 6 : 4)) return 0;     // (what the registers hold next to the working set)
  IspFusedArgs a = {};
  a.n_images = m; a.iters = iters; a.n_groups = n_groups;
  a.gamma_inv = (float)(1.0 / (double)gamma); a.intensity = intensity; a.la = la; a.ca = ca; a.state9 = state9;
  const unsigned limit = g_ispf_poll_limit.load(std::memory_order_relaxed);
  a.spin_limit = limit ? limit - 1u : 2000000u;
  a.fault = reinterpret_cast<unsigned*>(fp) + 62;              // FP_ERROR
  for (int i = 0; i < m; ++i) { a.io[i].img = images[i]; a.io[i].out = outs[i]; }
  ew::ResidentOrder& ord = ew::resident_order();
  std::lock_guard<std::mutex> lock(ord.mu);
  if (!g_ispf.buf[dev]) {
    void* d = nullptr;
    MI_HIP(hipMalloc(&d, (size_t)2 * ISPF_SYNC_WORDS * 4));
    MI_HIP(hipMemset(d, 0, (size_t)2 * ISPF_SYNC_WORDS * 4));
    g_ispf.buf[dev] = static_cast<unsigned*>(d);
  }
  if (int rc = ew::resident_enter_locked(dev, s)) return rc;
  const unsigned half = g_ispf.launches[dev] & 1u;
  a.sync = g_ispf.buf[dev] + half * ISPF_SYNC_WORDS;
  a.sync_next = g_ispf.buf[dev] + (half ^ 1u) * ISPF_SYNC_WORDS;
  a.mailbox = ord.mailbox_dev[dev] + ew::MAILBOX_ISP_TONEMAP;
  bool ok = false;
  int rc = 0;
  const bool ca0 = ca == 0.f;
  const int slot = large ? 8 + (ca0 ? 0 : 1) : (f16 ? 0 : 4) + (ca0 ? 0 : 2) + (pipe ? 0 : 1);
#define MI_ISPF(TI, CA0, MAXIT, PIPE, BPC) rc = ispf_launch<TI, CA0, MAXIT, PIPE, BPC>(a, nblocks, slot, s, &ok)
  if (large) {
    if (ca0) rc = ispf_launch<half_t, true, 12, false, 2, 5>(a, nblocks, slot, s, &ok);
    else rc = ispf_launch<half_t, false, 12, false, 2, 5>(a, nblocks, slot, s, &ok);
  } else if (f16) {
    if (ca0) { if (pipe) MI_ISPF(half_t, true, 3, true, 2); else MI_ISPF(half_t, true, 6, false, 2); }
    else     { if (pipe) MI_ISPF(half_t, false, 3, true, 2); else MI_ISPF(half_t, false, 6, false, 2); }
  } else {
    if (ca0) { if (pipe) MI_ISPF(float, true, 3, true, 2); else MI_ISPF(float, true, 4, false, 2); }
    else     { if (pipe) MI_ISPF(float, false, 3, true, 2); else MI_ISPF(float, false, 4, false, 2); }
  }
#undef MI_ISPF
  if (rc || !ok) return rc;
  ++g_ispf.launches[dev];
  if (int rc2 = ew::resident_leave_locked(dev, s)) return rc2;
  *done = true;
  return 0;
}

extern "C" int mi_isp_reinhard_set_poll_limit(unsigned polls) {
  g_ispf_poll_limit.store(polls, std::memory_order_relaxed);
  return 0;
}
// the fused ISP tonemap's mailbox word of the current device (see mi_isp_metering_faults)
extern "C" int mi_isp_reinhard_faults(int clear) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  ew::ResidentOrder& ord = ew::resident_order();
  std::lock_guard<std::mutex> lock(ord.mu);
  if (!ord.mailbox_host[dev]) return 0;
  volatile unsigned* mb = ord.mailbox_host[dev] + ew::MAILBOX_ISP_TONEMAP;
  const unsigned v = *mb;
  if (clear) *mb = 0;
  return (int)v;
}

// N images of one tonemap_reinhard call (camera_isp.py:399-403) with 4 launches in total instead of
// 4 per image: prep, pass 1 over all images (grid.y = image), per-image max_out, pass 2 over all.
static int reinhard_batch_impl(void* const* images, uint8_t* const* outs, int n, int H, int W, int dtype,
                               const float* state9, float gamma, float intensity, float light_adapt,
                               float color_adapt, int transform, void* ws, void* stream, bool write_back) {
  MI_REQUIRE(images && outs && state9 && ws, "reinhard_batch: null pointer");
  MI_REQUIRE(n >= 0, "reinhard_batch: negative image count");
  MI_REQUIRE(H > 0 && W > 0, "reinhard_batch: bad shape");
  MI_REQUIRE(dtype == MI_F16 || dtype == MI_F32, "reinhard_batch: image must be f16 or f32");
  MI_REQUIRE(gamma > 0.f, "reinhard_batch: gamma must be positive");
  MI_REQUIRE(transform >= MI_T_NONE && transform <= MI_T_TRANSVERSE, "reinhard_batch: bad transform");
  MI_REQUIRE(transform != MI_T_TRANSVERSE || H == W, "reinhard_batch: transverse needs a square image");
  hipStream_t s = (hipStream_t)stream;
  float* fp = static_cast<float*>(ws);
  float* partials = fp + FP_COUNT;
  const int cap = mi_partial_cap(H, W);
  if (n == 0) return 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int m = n - i0 < 64 ? n - i0 : 64;
    {
      for (int i = 0; i < m; ++i) MI_REQUIRE(images[i0 + i] && outs[i0 + i], "reinhard_batch: image %d is null", i0 + i);
      bool done = false;                                       // one persistent launch when the group's p fits the chip
      if (write_back)
        if (int rc = isp_reinhard_fused(images + i0, outs + i0, m, H, W, dtype, state9, gamma, intensity, light_adapt,
                                        color_adapt, transform, fp, s, &done))
          return rc;
      if (done) continue;
    }
    PassArgs a = {};
    a.fp = fp; a.partials = partials; a.part_stride = cap; a.n_px = (int64_t)H * W;
    a.vec_in = 1; a.vec_out = 1;
    a.isp_state9 = state9; a.pull_intensity = intensity;     // scalars derived in pass 1's prologue
    for (int i = 0; i < m; ++i) {
      MI_REQUIRE(images[i0 + i] && outs[i0 + i], "reinhard_batch: image %d is null", i0 + i);
      a.srcs.p[i] = images[i0 + i]; a.dsts.p[i] = outs[i0 + i];
      a.vec_in = a.vec_in && vec_ok(images[i0 + i], dtype);
      a.vec_out = a.vec_out && vec_ok(outs[i0 + i], MI_U8);
    }
    a.batched = 1;
    a.gamma_inv = (float)(1.0 / (double)gamma); a.la = light_adapt; a.ca = color_adapt; a.out_scale = 255.f;
    a.transform = transform; a.H = H; a.W = W;
    int nb = pass_blocks(a.n_px, cap / m > 0 ? cap / m : 1);
    float* maxouts = partials + (size_t)2 * cap;           // partial row 2: unused by the 2-row reductions
    a.maxouts = maxouts;
    // Zig-zag over the images: the blocks of a batched launch are dispatched image by image, and what a pass touched last
    // is what the 256 MB Infinity Cache still holds.  The caller produced the images in list order (load_packed12 per
    // camera), so pass 1 starts with the LAST image; it leaves p of the first image last, so pass 2 runs forward again.
    // At 4K (6 x 75.5 MB) about half of each pass's reads then never reach HBM (profiles/r03_isp_order.txt).
    int order1 = 1, order2 = 0;
#ifdef MI_ISP_MEASURE
    if (const char* o = getenv("MI_ISP_ORDER")) { order1 = o[0] == 'r'; order2 = o[0] && o[1] == 'r'; }
#endif
    auto set_lists = [&](int reversed) {
      for (int i = 0; i < m; ++i) {
        const int j = reversed ? m - 1 - i : i;
        a.srcs.p[i] = images[i0 + j]; a.dsts.p[i] = outs[i0 + j];
      }
    };
    set_lists(order1);
    a.no_writeback = write_back ? 0 : 1;
    if (int rc = launch_pass(PM_ISP_RH_P1, dtype, MI_U8, a, nb, s, m)) return rc;
    a.pull_maxout_n = nb;                                    // max_out per image folded in pass 2's prologue
    set_lists(order2);
    a.part_flip = order1 != order2;                          // pass 1 left image j's maxima in the row of ITS list position
    // without the write-back pass 2 derives p again from the untouched image (and rounds it as the write-back would have)
    if (int rc = launch_pass(write_back ? PM_ISP_RH_P2 : PM_ISP_RH_P2R, dtype, MI_U8, a, nb, s, m)) return rc;
  }
  return 0;
}

extern "C" int mi_isp_reinhard_batch(void* const* images, uint8_t* const* outs, int n, int H, int W, int dtype,
                                     const float* state9, float gamma, float intensity, float light_adapt,
                                     float color_adapt, int transform, void* ws, void* stream) {
  return reinhard_batch_impl(images, outs, n, H, W, dtype, state9, gamma, intensity, light_adapt, color_adapt, transform, ws,
                             stream, true);
}

// Extension (not the reference's semantics): the same u8 outputs WITHOUT overwriting the images with p (camera_isp.py:211) -
// for callers that drop or re-use the images as they were.  Pass 1 only reduces, pass 2 recomputes p: a third of the two
// passes' bytes (the write of p and its re-read) is not moved.
extern "C" int mi_isp_reinhard_batch_keep(const void* const* images, uint8_t* const* outs, int n, int H, int W, int dtype,
                                          const float* state9, float gamma, float intensity, float light_adapt,
                                          float color_adapt, int transform, void* ws, void* stream) {
  return reinhard_batch_impl(const_cast<void* const*>(images), outs, n, H, W, dtype, state9, gamma, intensity, light_adapt,
                             color_adapt, transform, ws, stream, false);
}

// tonemap_reinhard of a list of images straight to planar YUV 4:2:0 (u8): pass 1 as in mi_isp_reinhard_batch (the images
// are overwritten with p, camera_isp.py:211), pass 2 fused with the conversion.  No orientation transform.
extern "C" int mi_isp_reinhard_batch_yuv420(void* const* images, uint8_t* const* yuv_outs, int n, int H, int W, int dtype,
                                            const float* state9, float gamma, float intensity, float light_adapt,
                                            float color_adapt, void* ws, void* stream) {
  MI_REQUIRE(images && yuv_outs && state9 && ws, "reinhard_batch_yuv420: null pointer");
  MI_REQUIRE(n >= 0, "reinhard_batch_yuv420: negative image count");
  MI_REQUIRE(H > 0 && W > 0 && H % 2 == 0 && W % 16 == 0, "reinhard_batch_yuv420: H must be even and W a multiple of 16, got %dx%d", H, W);
  MI_REQUIRE(dtype == MI_F16 || dtype == MI_F32, "reinhard_batch_yuv420: image must be f16 or f32");
  MI_REQUIRE(gamma > 0.f, "reinhard_batch_yuv420: gamma must be positive");
  hipStream_t s = (hipStream_t)stream;
  float* fp = static_cast<float*>(ws);
  float* partials = fp + FP_COUNT;
  const int cap = mi_partial_cap(H, W);
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int m = n - i0 < 64 ? n - i0 : 64;
    PassArgs a = {};
    a.fp = fp; a.partials = partials; a.part_stride = cap; a.n_px = (int64_t)H * W;
    a.vec_in = 1; a.vec_out = 1;
    a.isp_state9 = state9; a.pull_intensity = intensity;
    ew::PtrList outs = {};
    for (int i = 0; i < m; ++i) {
      MI_REQUIRE(images[i0 + i] && yuv_outs[i0 + i], "reinhard_batch_yuv420: image %d is null", i0 + i);
      MI_REQUIRE(vec_ok(images[i0 + i], dtype) && mi_aligned(yuv_outs[i0 + i], 16), "reinhard_batch_yuv420: buffers must be 16-byte aligned");
      a.srcs.p[i] = images[i0 + i]; a.dsts.p[i] = yuv_outs[i0 + i];
      outs.p[i] = yuv_outs[i0 + i];
    }
    a.batched = 1;
    a.gamma_inv = (float)(1.0 / (double)gamma); a.la = light_adapt; a.ca = color_adapt; a.out_scale = 255.f;
    a.transform = MI_T_NONE; a.H = H; a.W = W;
    const int nb = pass_blocks(a.n_px, cap / m > 0 ? cap / m : 1);
    if (int rc = launch_pass(PM_ISP_RH_P1, dtype, MI_U8, a, nb, s, m)) return rc;
    const int64_t lanes = (int64_t)(H / 2) * (W / 8);
    int blocks = (int)((lanes + EW_THREADS - 1) / EW_THREADS);
    if (blocks > 2048) blocks = 2048;
    if (dtype == MI_F16)
      hipLaunchKernelGGL((isp_p2_yuv420_kernel<half_t>), dim3(blocks, m), dim3(EW_THREADS), 0, s, a.srcs, outs, H, W, a.gamma_inv,
                         partials + cap, nb);
    else
      hipLaunchKernelGGL((isp_p2_yuv420_kernel<float>), dim3(blocks, m), dim3(EW_THREADS), 0, s, a.srcs, outs, H, W, a.gamma_inv,
                         partials + cap, nb);
    MI_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int mi_isp_linear_batch(const void* const* images, uint8_t* const* outs, int n, int H, int W, int dtype,
                                   const float* state9, float gamma, int transform, void* ws, void* stream) {
  MI_REQUIRE(images && outs && state9 && ws, "linear_batch: null pointer");
  MI_REQUIRE(n >= 0 && H > 0 && W > 0, "linear_batch: bad shape");
  MI_REQUIRE(dtype == MI_F16 || dtype == MI_F32, "linear_batch: image must be f16 or f32");
  MI_REQUIRE(gamma > 0.f, "linear_batch: gamma must be positive");
  MI_REQUIRE(transform >= MI_T_NONE && transform <= MI_T_TRANSVERSE, "linear_batch: bad transform");
  MI_REQUIRE(transform != MI_T_TRANSVERSE || H == W, "linear_batch: transverse needs a square image");
  hipStream_t s = (hipStream_t)stream;
  float* fp = static_cast<float*>(ws);
  if (n == 0) return 0;
  if (int rc = isp_reinhard_prep(state9, fp, 0.f, 0.f, s)) return rc;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int m = n - i0 < 64 ? n - i0 : 64;
    PassArgs a = {};
    a.fp = fp; a.n_px = (int64_t)H * W; a.vec_in = 1; a.vec_out = 1;
    for (int i = 0; i < m; ++i) {
      MI_REQUIRE(images[i0 + i] && outs[i0 + i], "linear_batch: image %d is null", i0 + i);
      a.srcs.p[i] = images[i0 + i]; a.dsts.p[i] = outs[i0 + i];
      a.vec_in = a.vec_in && vec_ok(images[i0 + i], dtype);
      a.vec_out = a.vec_out && vec_ok(outs[i0 + i], MI_U8);
    }
    a.batched = 1;
    a.gamma_inv = (float)(1.0 / (double)gamma); a.out_scale = 255.f;
    a.transform = transform; a.H = H; a.W = W;
    if (int rc = launch_pass(PM_LINEAR_STORE, dtype, MI_U8, a, pass_blocks(a.n_px, 1 << 30), s, m)) return rc;
  }
  return 0;
}

extern "C" int mi_isp_linear(const void* image, uint8_t* out, int H, int W, int dtype, const float* state9,
                             float gamma, int transform, void* ws, void* stream) {
  MI_REQUIRE(image && out && state9 && ws, "linear: null pointer");
  MI_REQUIRE(H > 0 && W > 0, "linear: bad shape");
  MI_REQUIRE(dtype == MI_F16 || dtype == MI_F32, "linear: image must be f16 or f32");
  MI_REQUIRE(gamma > 0.f, "linear: gamma must be positive");
  MI_REQUIRE(transform >= MI_T_NONE && transform <= MI_T_TRANSVERSE, "linear: bad transform");
  MI_REQUIRE(transform != MI_T_TRANSVERSE || H == W, "linear: transverse needs a square image");
  hipStream_t s = (hipStream_t)stream;
  float* fp = static_cast<float*>(ws);
  // FP_LO / FP_INV from the metering bounds state9[0..1] (camera_isp.py:226-227, tonemap.py:13)
  if (int rc = isp_reinhard_prep(state9, fp, 0.f, 0.f, s)) return rc;
  PassArgs a = {};
  a.src = image; a.dst = out; a.fp = fp; a.n_px = (int64_t)H * W;
  a.vec_in = vec_ok(image, dtype); a.vec_out = vec_ok(out, MI_U8);
  a.gamma_inv = (float)(1.0 / (double)gamma); a.out_scale = 255.f;
  a.transform = transform; a.H = H; a.W = W;
  return launch_pass(PM_LINEAR_STORE, dtype, MI_U8, a, pass_blocks(a.n_px, 1 << 30), s);
}

// ---- stateless tonemaps (tonemap.py) --------------------------------------------------------------
static int tonemap_check(const void* src, const void* dst, int H, int W, int in_dtype, int out_dtype, float gamma,
                         const void* ws) {
  MI_REQUIRE(src && dst && ws, "tonemap: null pointer");
  MI_REQUIRE(H > 0 && W > 0, "tonemap: bad shape");
  MI_REQUIRE(mi_valid_dtype(in_dtype) && mi_valid_dtype(out_dtype), "tonemap: bad dtype");
  MI_REQUIRE(gamma > 0.f, "tonemap: gamma must be positive");
  return 0;
}

extern "C" int mi_isp_tonemap_linear(const void* src, void* dst, int H, int W, int in_dtype, int out_dtype,
                                     float gamma, void* ws, void* stream) {
  if (int rc = tonemap_check(src, dst, H, W, in_dtype, out_dtype, gamma, ws)) return rc;
  hipStream_t s = (hipStream_t)stream;
  float* fp = static_cast<float*>(ws);
  float* partials = fp + FP_COUNT;
  const int cap = mi_partial_cap(H, W);
  PassArgs a = {};
  a.src = src; a.dst = dst; a.fp = fp; a.partials = partials; a.part_stride = cap; a.n_px = (int64_t)H * W;
  a.vec_in = vec_ok(src, in_dtype); a.vec_out = vec_ok(dst, out_dtype);
  a.gamma_inv = 1.0f / gamma;                      // tonemap.py:16: 1/gamma evaluated in f32
  a.out_scale = mi_scale_factor(out_dtype); a.transform = MI_T_NONE; a.H = H; a.W = W;
  const int nb = pass_blocks(a.n_px, cap);
  if (int rc = launch_pass(PM_MINMAX, in_dtype, out_dtype, a, nb, s)) return rc;       // tonemap.py:23
  a.pull_mode = FIN_BOUNDS; a.pull_partials = partials; a.pull_stride = cap; a.pull_n = nb; a.fp_w = fp;
  return launch_pass(PM_LINEAR_STORE, in_dtype, out_dtype, a, nb, s);
}

extern "C" int mi_isp_tonemap_reinhard(const void* src, void* dst, int H, int W, int in_dtype, int out_dtype,
                                       float gamma, float intensity, float light_adapt, float color_adapt,
                                       void* ws, void* stream) {
  if (int rc = tonemap_check(src, dst, H, W, in_dtype, out_dtype, gamma, ws)) return rc;
  hipStream_t s = (hipStream_t)stream;
  float* fp = static_cast<float*>(ws);
  float* partials = fp + FP_COUNT;
  const int cap = mi_partial_cap(H, W);
  PassArgs a = {};
  a.src = src; a.dst = dst; a.fp = fp; a.partials = partials; a.part_stride = cap; a.n_px = (int64_t)H * W;
  a.vec_in = vec_ok(src, in_dtype); a.vec_out = vec_ok(dst, out_dtype);
  a.gamma_inv = 1.0f / gamma; a.la = light_adapt; a.ca = color_adapt;
  a.out_scale = mi_scale_factor(out_dtype); a.transform = MI_T_NONE; a.H = H; a.W = W;
  const int nb = pass_blocks(a.n_px, cap);
  if (int rc = launch_pass(PM_MINMAX, in_dtype, out_dtype, a, nb, s)) return rc;      // tonemap.py:146
  const PullSrc bounds = {partials, cap, nb, 0};
  return tonemap_reinhard_tail(src, dst, H, W, in_dtype, out_dtype, gamma, intensity, light_adapt, color_adapt, fp,
                               -1, bounds, s);
}
