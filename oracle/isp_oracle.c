/*
 * isp_oracle.c -- plain-C (OpenMP) restatement of the taichi_image camera-ISP hot path.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ (cross-check of the NumPy oracle, full-size parity)
 * and by the cpu_baseline leg of bench.py.  Never linked into or called by the product.
 *
 * PARITY STATUS: parity unpinned by the reference (it cannot run here and ships no golden
 * vectors); pinned by the same hand-derived KATs / properties as oracle/isp_oracle.py, against
 * which tests/test_c_oracle.py compares it (bit-exact for unpack and demosaic).
 *
 * Every function cites the reference lines it follows (relative to /root/reference/taichi_image/).
 * All arithmetic is float32 in the reference's operation order; compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- float16 rounding (ti.cast(f32 -> f16), round-to-nearest-even), software ---------------- */
static inline uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7FFFFFFFu;
  if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u : 0)); /* inf / nan */
  if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                   /* overflow -> inf */
  if (x < 0x33000001u) return (uint16_t)sign;                                                /* underflow -> 0 */
  int e = (int)(x >> 23) - 127;
  uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
  int shift = e < -14 ? 13 + (-14 - e) : 13;     /* subnormal halves lose more bits */
  uint32_t half_m = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (half_m & 1))) half_m++;
  uint32_t out = e < -14 ? half_m : (((uint32_t)(e + 15) << 10) + (half_m - 0x400u));
  return (uint16_t)(sign | out);   /* a mantissa carry correctly bumps the exponent */
}
static inline float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1F, m = h & 0x3FF, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else {
      int s = 0;
      while (!(m & 0x400)) { m <<= 1; s++; }
      x = sign | ((uint32_t)(127 - 15 - s + 1) << 23) | ((m & 0x3FF) << 13);
    }
  } else if (e == 31) x = sign | 0x7F800000u | (m << 13);
  else x = sign | ((e + 112) << 23) | (m << 13);
  float f;
  memcpy(&f, &x, 4);
  return f;
}
static inline float round_f16(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }

int orc_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* limit the OpenMP team (the GPU box gives one GPU a share of 16 host cores) */
void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

uint16_t orc_f32_to_f16_bits(float f) { return f32_to_f16_bits(f); }

/* ---- packed.py:24-31 (standard), :37-44 (IDS) ------------------------------------------------ */
static inline void decode_pair(const uint8_t* b, int ids, uint32_t* p0, uint32_t* p1) {
  if (!ids) {
    *p0 = ((uint32_t)(b[1] & 0xF) << 8) | b[0];
    *p1 = ((uint32_t)b[2] << 4) | (b[1] >> 4);
  } else {
    *p0 = ((uint32_t)b[0] << 4) | (b[2] & 0xF);
    *p1 = ((uint32_t)b[1] << 4) | (b[2] >> 4);
  }
}

/* decode12 scaled to the work dtype (packed.py:98-100 with scale 1), result widened to f32 */
void orc_decode12_scaled(const uint8_t* enc, int64_t n_pairs, int ids, int work_f16, float* out) {
  const float k = (float)(1.0 / 4095.0);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n_pairs; ++j) {
    uint32_t p0, p1;
    decode_pair(enc + 3 * j, ids, &p0, &p1);
    float a = (float)p0 * k, b = (float)p1 * k;
    if (work_f16) { a = round_f16(a); b = round_f16(b); }
    out[2 * j] = a;
    out[2 * j + 1] = b;
  }
}

/* unscaled u16 decode (packed.py:102-104) */
void orc_decode12_u16(const uint8_t* enc, int64_t n_pairs, int ids, uint16_t* out) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n_pairs; ++j) {
    uint32_t p0, p1;
    decode_pair(enc + 3 * j, ids, &p0, &p1);
    out[2 * j] = (uint16_t)p0;
    out[2 * j + 1] = (uint16_t)p1;
  }
}

/* ---- bayer.py:15-55 ---------------------------------------------------------------------------- */
static const int TAP_DR[13] = {-2, -1, -1, -1, 0, 0, 0, 0, 0, 1, 1, 1, 2};
static const int TAP_DC[13] = {0, -1, 0, 1, -2, -1, 0, 1, 2, -1, 0, 1, 0};
static int KW[4][13][3];
static int kw_ready = 0;

static void wedge(const int a0, const int b0, const int b1, const int c0, const int c1, const int c2, int out[13]) {
  /* kernel.py:3-12 'symmetrical' of the three partial rows, on the diamond of bayer.py:15-27 */
  const int v[13] = {a0, b0, b1, b0, c0, c1, c2, c1, c0, b0, b1, b0, a0};
  memcpy(out, v, sizeof(v));
}
static void build_kernels(void) {
  if (kw_ready) return;
  int g_rb[13], r_g1[13], r_g2[13], rb_br[13], ident[13];
  wedge(-2, 0, 4, -2, 4, 8, g_rb);      /* bayer.py:37 */
  wedge(-2, -2, 8, 1, 0, 10, r_g1);     /* :38 */
  wedge(1, -2, 0, -2, 8, 10, r_g2);     /* :39 */
  wedge(-3, 4, 0, -3, 0, 12, rb_br);    /* :40 */
  wedge(0, 0, 0, 0, 0, 16, ident);      /* :41 */
  const int* site[4][3] = {{ident, g_rb, rb_br}, {r_g1, ident, r_g2}, {r_g2, ident, r_g1}, {rb_br, g_rb, ident}};
  for (int k = 0; k < 4; ++k)
    for (int t = 0; t < 13; ++t)
      for (int c = 0; c < 3; ++c) KW[k][t][c] = site[k][c][t];
  kw_ready = 1;
}
void orc_bayer_kernels(int32_t out[4 * 13 * 3]) {
  build_kernels();
  for (int k = 0; k < 4; ++k)
    for (int t = 0; t < 13; ++t)
      for (int c = 0; c < 3; ++c) out[(k * 13 + t) * 3 + c] = KW[k][t][c];
}

/* bayer.py:92-97: kernels at (even,even), (odd,even), (even,odd), (odd,odd); BayerPattern values */
static const int KPAT[4][4] = {{0, 1, 2, 3} /*RGGB*/, {2, 3, 0, 1} /*GRBG*/, {1, 0, 3, 2} /*GBRG*/, {3, 2, 1, 0} /*BGGR*/};

/* filter_at (bayer.py:138-155): cfa holds f32 values; returns the clamped [0,1] pixel, optionally
 * rounded to f16 (the work dtype the reference stores the RGB image in, scale 1). */
void orc_demosaic(const float* cfa, int H, int W, int pattern, float in_scale, const float* ccm9, int round_out_f16,
                  float* rgb) {
  build_kernels();
#pragma omp parallel for schedule(static)
  for (int r = 0; r < H; ++r) {
    for (int c = 0; c < W; ++c) {
      const int (*w)[3] = KW[KPAT[pattern][(r & 1) + 2 * (c & 1)]];
      float acc[3] = {0.f, 0.f, 0.f}, t[3] = {0.f, 0.f, 0.f};
      for (int k = 0; k < 13; ++k) {
        const int rr = r + TAP_DR[k], cc = c + TAP_DC[k];
        if (rr >= 0 && rr < H && cc >= 0 && cc < W) {
          const float x = cfa[(size_t)rr * W + cc];
          for (int ch = 0; ch < 3; ++ch) {
            acc[ch] = acc[ch] + x * (float)w[k][ch];
            t[ch] = t[ch] + (float)w[k][ch];
          }
        }
      }
      float v[3];
      for (int ch = 0; ch < 3; ++ch) v[ch] = acc[ch] / (in_scale * t[ch]);
      if (ccm9) {
        const float a = v[0], b = v[1], d = v[2];
        for (int ch = 0; ch < 3; ++ch) v[ch] = (ccm9[3 * ch] * a + ccm9[3 * ch + 1] * b) + ccm9[3 * ch + 2] * d;
      }
      for (int ch = 0; ch < 3; ++ch) {
        float x = fminf(fmaxf(v[ch], 0.f), 1.f);
        rgb[((size_t)r * W + c) * 3 + ch] = round_out_f16 ? round_f16(x) : x;
      }
    }
  }
}

/* ---- tonemap.py:135-168 (stateless Reinhard) on a scale-1 f32 image ------------------------------
 * out_kind: 0 = u8, 2 = f16 (bits in uint16), 3 = f32.  stats8 (optional) receives
 * lo, hi, Bmin, Bmax, lmean, gmean, lo2, hi2. */
static inline float gray3(const float* p) { return (p[0] * 0.299f + p[1] * 0.587f) + p[2] * 0.114f; }

void orc_tonemap_reinhard(const float* src, int H, int W, float gamma, float intensity, float la, float ca,
                          int out_kind, void* out, float* stats8) {
  const size_t n = (size_t)H * W;
  float lo = INFINITY, hi = -INFINITY;
#pragma omp parallel for reduction(min : lo) reduction(max : hi) schedule(static)
  for (size_t i = 0; i < n * 3; ++i) {
    lo = fminf(lo, src[i]);
    hi = fmaxf(hi, src[i]);
  }
  const float inv = 1.0f / (hi - lo);                                        /* tonemap.py:13 */
  float* temp = (float*)malloc(n * 3 * sizeof(float));
  double slog = 0, sgray = 0, s0 = 0, s1 = 0, s2 = 0;
  float lmin = INFINITY, lmax = -INFINITY;
#pragma omp parallel for reduction(+ : slog, sgray, s0, s1, s2) reduction(min : lmin) reduction(max : lmax) schedule(static)
  for (size_t i = 0; i < n; ++i) {
    float* t = temp + 3 * i;
    for (int ch = 0; ch < 3; ++ch) t[ch] = fminf(fmaxf((src[3 * i + ch] - lo) * inv, 0.f), 1.f);   /* :147 */
    const float g = gray3(t);
    const float lg = logf(fmaxf(g, 1e-4f));                                 /* :89-90 */
    lmin = fminf(lmin, lg);
    lmax = fmaxf(lmax, lg);
    slog += lg; sgray += g; s0 += t[0]; s1 += t[1]; s2 += t[2];
  }
  const float nn = (float)n;
  const float Bmin = lmin, Bmax = -lmax;                                     /* :102 sign quirk */
  const float lmean = (float)slog / nn, gmean = (float)sgray / nn;
  const float rm[3] = {(float)s0 / nn, (float)s1 / nn, (float)s2 / nn};
  const float key = (Bmax - lmean) / (Bmax - Bmin);                          /* :116 */
  const float map_key = 0.3f + 0.7f * powf(key, 1.4f);
  float mean3[3];
  for (int c = 0; c < 3; ++c) mean3[c] = gmean + ca * (rm[c] - gmean);       /* :119 */
  const float ei = expf(-intensity);
  float lo2 = INFINITY, hi2 = -INFINITY;
#pragma omp parallel for reduction(min : lo2) reduction(max : hi2) schedule(static)
  for (size_t i = 0; i < n; ++i) {
    float* t = temp + 3 * i;
    const float g = gray3(t);
    float q[3];
    for (int c = 0; c < 3; ++c) {
      const float ac = g + ca * (t[c] - g);                                  /* :125 */
      const float am = mean3[c] + la * (ac - mean3[c]);                      /* :128 */
      const float ad = powf(ei * am, map_key);                               /* :129 */
      q[c] = t[c] * (1.0f / (ad + t[c]));                                    /* :131 */
    }
    for (int c = 0; c < 3; ++c) {
      t[c] = q[c];
      if (!isnan(q[c])) { lo2 = fminf(lo2, q[c]); hi2 = fmaxf(hi2, q[c]); }
    }
  }
  const float inv2 = 1.0f / (hi2 - lo2), ginv = 1.0f / gamma;
  const float scale = out_kind == 0 ? 255.f : 1.f;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n * 3; ++i) {
    float x = powf((temp[i] - lo2) * inv2, ginv);                            /* :16 */
    x = fminf(fmaxf(x, 0.f), 1.f);                                           /* NaN -> 0 */
    if (isnan(x)) x = 0.f;
    x *= scale;
    if (out_kind == 0) ((uint8_t*)out)[i] = (uint8_t)x;
    else if (out_kind == 2) ((uint16_t*)out)[i] = f32_to_f16_bits(x);
    else ((float*)out)[i] = x;
  }
  free(temp);
  if (stats8) {
    const float s[8] = {lo, hi, Bmin, Bmax, lmean, gmean, lo2, hi2};
    memcpy(stats8, s, sizeof(s));
  }
}

/* test/pipeline.py:26-32 (BASELINE config 2): decode12(scaled, work) -> bayer_to_rgb -> tonemap_reinhard */
void orc_pipeline12_reinhard(const uint8_t* packed, int H, int W, int ids, int pattern, int work_f16, float gamma,
                             float intensity, float la, float ca, int out_kind, void* out) {
  const size_t n = (size_t)H * W;
  float* cfa = (float*)malloc(n * sizeof(float));
  float* rgb = (float*)malloc(n * 3 * sizeof(float));
  orc_decode12_scaled(packed, (int64_t)(n / 2), ids, work_f16, cfa);
  orc_demosaic(cfa, H, W, pattern, 1.0f, NULL, work_f16, rgb);
  orc_tonemap_reinhard(rgb, H, W, gamma, intensity, la, ca, out_kind, out, NULL);
  free(cfa);
  free(rgb);
}

/* ==== camera_isp.py: the stateful ISP tonemap (metering state + Reinhard / linear to u8) ==========
 * Images are f32 arrays holding work-dtype values (f16 values widened exactly, or f32);
 * `work_f16` says which dtype the reference image has (it decides the rounding of the in-place
 * write-back of camera_isp.py:211). */

/* util.py:83-84 */
static inline float lerpf(float t, float a, float b) { return a + t * (b - a); }

/* camera_isp.py:142-175: metering_images over the stride-subsample of n images.
 * Sums are evaluated in float64 and rounded once (the reference's atomic-add order is not
 * deterministic); everything else is float32 in the reference's operation order. */
void orc_metering_images(const float* const* images, int n, int H, int W, int stride, float alpha, const float* prev9,
                         float* out9) {
  float lo = INFINITY, hi = -INFINITY;
  for (int k = 0; k < n; ++k) {
    const float* im = images[k];
#pragma omp parallel for reduction(min : lo) reduction(max : hi) schedule(static)
    for (int r = 0; r < H; r += stride)
      for (int c = 0; c < W; c += stride)
        for (int ch = 0; ch < 3; ++ch) {                                       /* :151-154 */
          const float v = im[((size_t)r * W + c) * 3 + ch];
          lo = fminf(lo, v);
          hi = fmaxf(hi, v);
        }
  }
  const float bmin = lerpf(alpha, lo, prev9[0]), bmax = lerpf(alpha, hi, prev9[1]);   /* :156-157 */
  double slog = 0, sgray = 0, s0 = 0, s1 = 0, s2 = 0;
  float lmin = INFINITY, lmax = -INFINITY;
  long cnt = 0;
  const float den = bmax - bmin + 1e-6f;                                       /* :118 */
  for (int k = 0; k < n; ++k) {
    const float* im = images[k];
#pragma omp parallel for reduction(+ : slog, sgray, s0, s1, s2, cnt) reduction(min : lmin) reduction(max : lmax) schedule(static)
    for (int r = 0; r < H; r += stride)
      for (int c = 0; c < W; c += stride) {
        const float* p = im + ((size_t)r * W + c) * 3;
        const float t[3] = {(p[0] - bmin) / den, (p[1] - bmin) / den, (p[2] - bmin) / den};
        const float g = gray3(t);                                              /* :119 */
        const float lg = logf(fmaxf(g, 1e-4f));                               /* :120 */
        if (!isnan(lg)) { lmin = fminf(lmin, lg); lmax = fmaxf(lmax, lg); }   /* :122-123 */
        slog += lg; sgray += g; s0 += t[0]; s1 += t[1]; s2 += t[2];           /* :125-127 */
        cnt += 1;
      }
  }
  const float nn = (float)cnt;                                                 /* :164 n = N * H' * W' */
  const float v[9] = {bmin, bmax, lmin, lmax, (float)slog / nn, (float)sgray / nn,
                      (float)s0 / nn, (float)s1 / nn, (float)s2 / nn};         /* :131-134 */
  for (int i = 0; i < 9; ++i) out9[i] = lerpf(alpha, v[i], prev9[i]);          /* :165-166 */
}

/* camera_isp.py:177-218: reinhard_kernel.  image (f32 holding work-dtype values) is overwritten
 * with p rounded to the work dtype (:211); out_u8 receives cast(255 * (p/max_out)^(1/gamma)). */
void orc_reinhard_isp(float* image, int H, int W, int work_f16, const float* m9, float gamma, float intensity,
                      float la, float ca, uint8_t* out_u8, float* max_out_ret) {
  const size_t n = (size_t)H * W;
  const float bmin = m9[0], bmax = m9[1], lmin = m9[2], lmax = m9[3], lmean = m9[4], mean = m9[5];
  const float key = (lmax - lmean) / (lmax - lmin);                            /* :192 */
  const float map_key = 0.3f + 0.7f * powf(key, 1.4f);                         /* :193 */
  float mean3[3];
  for (int c = 0; c < 3; ++c) mean3[c] = mean + ca * (m9[6 + c] - mean);       /* :195 */
  const float ei = expf(-intensity);
  const float range = bmax - bmin;
  float max_p = -INFINITY;
#pragma omp parallel for reduction(max : max_p) schedule(static)
  for (size_t i = 0; i < n; ++i) {
    float* px = image + 3 * i;
    const float t[3] = {(px[0] - bmin) / range, (px[1] - bmin) / range, (px[2] - bmin) / range};   /* :200 */
    const float g = gray3(t);
    for (int c = 0; c < 3; ++c) {
      const float ac = g + ca * (t[c] - g);                                    /* :204 */
      const float am = mean3[c] + la * (ac - mean3[c]);                        /* :207 */
      const float ad = powf(ei * am, map_key);                                 /* :208 */
      const float p = t[c] * (1.0f / (ad + t[c]));                             /* :210 */
      if (!isnan(p)) max_p = fmaxf(max_p, p);                                  /* :213 */
      px[c] = work_f16 ? round_f16(p) : p;                                     /* :211 */
    }
  }
  const float max_out = fmaxf(1e-6f, max_p);                                   /* :190 */
  const float ginv = 1.0f / gamma;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n * 3; ++i) {
    float q = 255.f * powf(image[i] / max_out, ginv);                          /* :217-218 */
    if (isnan(q)) q = 0.f;
    q = fminf(fmaxf(q, 0.f), 255.f);
    out_u8[i] = (uint8_t)q;
  }
  if (max_out_ret) *max_out_ret = max_out;
}

/* camera_isp.py:220-227 -> tonemap.py:12-17 with the metering bounds, u8 out */
void orc_linear_isp(const float* image, int H, int W, const float* m9, float gamma, uint8_t* out_u8) {
  const size_t n = (size_t)H * W * 3;
  const float lo = m9[0], inv = 1.0f / (m9[1] - m9[0]), ginv = 1.0f / gamma;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; ++i) {
    float x = powf((image[i] - lo) * inv, ginv);
    x = fminf(fmaxf(x, 0.f), 1.f);
    if (isnan(x)) x = 0.f;
    out_u8[i] = (uint8_t)(x * 255.f);
  }
}
