// C ABI of the tile-based entry points (demosaic, fused load, fused config-2 pipeline) plus the
// library-wide plumbing (version, error string, workspace size).
#include <stdarg.h>
#include <stdlib.h>

#include "isp_elementwise.h"
#include "isp_tile.h"
#include "isp_resize_tile.h"
#include "isp_stream.h"
#include "isp_mega.h"
#include "isp_mega_cam.h"
#include "isp_stream_resize.h"
#include <mutex>
#include <atomic>

static thread_local char g_err[512] = "";

void mi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mi_isp_version(void) { return 1000; }  // 0.1.0 -> major*1e4 + minor*1e3 ... (monotone)
extern "C" const char* mi_isp_last_error(void) { return g_err; }

extern "C" int mi_isp_bayer_weights(int32_t out[4 * 13 * 3]) {
  MI_REQUIRE(out, "bayer_weights: null pointer");
  // host copy of the table the kernels are compiled with (isp_tile.h: tile::KW)
  static const int8_t kw[4][13][3] = {
      {{0, -2, -3}, {0, 0, 4}, {0, 4, 0}, {0, 0, 4}, {0, -2, -3}, {0, 4, 0}, {16, 8, 12},
       {0, 4, 0}, {0, -2, -3}, {0, 0, 4}, {0, 4, 0}, {0, 0, 4}, {0, -2, -3}},
      {{-2, 0, 1}, {-2, 0, -2}, {8, 0, 0}, {-2, 0, -2}, {1, 0, -2}, {0, 0, 8}, {10, 16, 10},
       {0, 0, 8}, {1, 0, -2}, {-2, 0, -2}, {8, 0, 0}, {-2, 0, -2}, {-2, 0, 1}},
      {{1, 0, -2}, {-2, 0, -2}, {0, 0, 8}, {-2, 0, -2}, {-2, 0, 1}, {8, 0, 0}, {10, 16, 10},
       {8, 0, 0}, {-2, 0, 1}, {-2, 0, -2}, {0, 0, 8}, {-2, 0, -2}, {1, 0, -2}},
      {{-3, -2, 0}, {4, 0, 0}, {0, 4, 0}, {4, 0, 0}, {-3, -2, 0}, {0, 4, 0}, {12, 8, 16},
       {0, 4, 0}, {-3, -2, 0}, {4, 0, 0}, {0, 4, 0}, {4, 0, 0}, {-3, -2, 0}}};
  for (int k = 0; k < 4; ++k)
    for (int t = 0; t < 13; ++t)
      for (int c = 0; c < 3; ++c) out[(k * 13 + t) * 3 + c] = kw[k][t][c];
  return 0;
}

extern "C" size_t mi_isp_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
#ifdef MI_STREAM_STAMPS
  return (size_t)(FP_COUNT + ((size_t)strm::PART_ROWS + 16) * (size_t)mi_partial_cap(H, W)) * sizeof(float);
#endif
  return (size_t)(FP_COUNT + (size_t)strm::PART_ROWS * (size_t)mi_partial_cap(H, W)) * sizeof(float);
}

// ---------------------------------------------------------------------------------------------
static int fill_common(tile::Params& p, int H, int W, int pattern, const float* ccm9, const char* who) {
  MI_REQUIRE(H > 0 && W > 0, "%s: bad shape %dx%d", who, H, W);
  MI_REQUIRE(H % 2 == 0 && W % 2 == 0, "%s: image must be even size, got %dx%d", who, H, W);  // bayer.py:206
  MI_REQUIRE(pattern >= MI_RGGB && pattern <= MI_BGGR, "%s: bad pattern %d", who, pattern);
  p.H = H; p.W = W;
  p.has_ccm = ccm9 != nullptr;
  for (int i = 0; i < 9; ++i) p.ccm[i] = ccm9 ? ccm9[i] : (i % 4 == 0 ? 1.f : 0.f);
  p.gamma_inv = 1.f; p.la = 1.f; p.ca = 0.f;
  tile::set_weights(p);
  return 0;
}

static int vec_store_ok(const void* dst, int W, int out_dtype) {
  // a strip row is 24 contiguous elements at element offset (r*W + c)*3, c % 8 == 0
  return W % 8 == 0 && mi_aligned(dst, out_dtype == MI_U8 ? 8 : 16);
}

extern "C" int mi_isp_demosaic(const void* cfa, void* rgb, int H, int W, int in_dtype, int out_dtype, int pattern,
                               const float* ccm9, void* stream) {
  MI_REQUIRE(cfa && rgb, "demosaic: null pointer");
  MI_REQUIRE(mi_valid_dtype(in_dtype) && mi_valid_dtype(out_dtype), "demosaic: bad dtype");
  tile::Params p = {};
  if (int rc = fill_common(p, H, W, pattern, ccm9, "demosaic")) return rc;
  p.src = cfa; p.dst = rgb;
  p.src_kind = in_dtype;                       // SRC_CFA_* share the MI_* numbering
  p.out_dtype = out_dtype;
  p.vec_store = vec_store_ok(rgb, W, out_dtype);
  p.in_scale = mi_scale_factor(in_dtype);
  p.out_scale = mi_scale_factor(out_dtype);
  // work type: f16 holds u8 and f16 inputs exactly; u16 / f32 need f32
  const int work = (in_dtype == MI_U8 || in_dtype == MI_F16) ? MI_F16 : MI_F32;
  return tile::launch(p, work, pattern, tile::EPI_STORE, (hipStream_t)stream);
}

static int packed_params(tile::Params& p, const uint8_t* packed, int H, int W, int bits, int ids_format,
                         int work_dtype, const char* who) {
  MI_REQUIRE(packed, "%s: null packed pointer", who);
  MI_REQUIRE(bits == 12 || bits == 16, "%s: bits must be 12 or 16, got %d", who, bits);
  MI_REQUIRE(work_dtype == MI_F16 || work_dtype == MI_F32, "%s: work dtype must be f16 or f32", who);
  p.src = packed;
  p.src_kind = bits == 16 ? tile::SRC_PACKED16 : (ids_format ? tile::SRC_PACKED12_IDS : tile::SRC_PACKED12);
  p.src_fast = bits == 16 ? (W % 8 == 0 && mi_aligned(packed, 16)) : (W % 8 == 0 && mi_aligned(packed, 4));
  p.in_scale = 1.f;                            // the decoded CFA is f16/f32 in [0, 1]
  p.k_decode = (float)(1.0 / (bits == 16 ? 65535.0 : 4095.0));   // packed.py:99,140 with scale 1.0
  return 0;
}

#ifdef MI_STREAM_STAMPS
// measurement build only: a home for the in-kernel stamps of the kernels that take no workspace (16 words per wave)
static float* stamp_buffer() {
  static float* buf = nullptr;
  if (!buf && hipMalloc(&buf, 4096 * 16 * 4) == hipSuccess) (void)hipMemset(buf, 0, 4096 * 16 * 4);
  return buf;
}
extern "C" int mi_isp_debug_read_stamps(void* host_out, int n_waves) {
  MI_REQUIRE(host_out && n_waves > 0 && n_waves <= 4096, "debug_read_stamps: bad arguments");
  MI_HIP(hipDeviceSynchronize());
  MI_HIP(hipMemcpy(host_out, stamp_buffer(), (size_t)n_waves * 16 * 4, hipMemcpyDeviceToHost));
  MI_HIP(hipMemset(stamp_buffer(), 0, 4096 * 16 * 4));
  return 0;
}
#endif

// The streaming kernels (isp_stream.h) take the standard 12-bit layout with aligned rows and whole 8-pixel units and
// store through 16-byte units; everything else stays with the tile kernels.  MI_ISP_MEASURE builds can switch them
// off (MI_ISP_NO_STREAM=1) to time the tile path.
static bool use_stream(const tile::Params& p, int work_dtype, const void* out, int out_dtype) {
#ifdef MI_ISP_MEASURE
  static const bool off = getenv("MI_ISP_NO_STREAM") != nullptr;
  if (off) return false;
#endif
  return strm::supported(p, work_dtype) && (!out || vec_store_ok(out, p.W, out_dtype)) &&
         (int64_t)p.H * p.W * 3 * (int64_t)mi_dtype_size(out_dtype) < (int64_t)strm::INVALID_OFF;
}

static int load_packed_impl(const uint8_t* packed, void* rgb, int H, int W, int bits, int ids_format,
                            int pattern, const float* ccm9, int work_dtype, int Hd, int Wd, float scale,
                            void* sub, int sub_stride, void* stream) {
  MI_REQUIRE(rgb, "load_packed: null output");
  tile::Params p = {};
  if (int rc = fill_common(p, H, W, pattern, ccm9, "load_packed")) return rc;
  if (int rc = packed_params(p, packed, H, W, bits, ids_format, work_dtype, "load_packed")) return rc;
  p.dst = rgb; p.out_dtype = work_dtype; p.out_scale = 1.f;
  if (scale > 0.f) {
    // unpack -> demosaic -> bilinear fused (isp_resize_tile.h); the caller checks the scale first
    MI_REQUIRE(Hd > 0 && Wd > 0 && H >= 2 && W >= 2, "load_packed: bad output shape %dx%d", Hd, Wd);
    if (use_stream(p, work_dtype, nullptr, work_dtype) && rstrm::supported(p, work_dtype, rgb, Hd, Wd, scale, scale)) {
      rstrm::RSArgs ra = {};
      ra.t = p; ra.Hd = Hd; ra.Wd = Wd; ra.s0 = scale; ra.s1 = scale;
      rstrm::geometry(H, W, ra);
#ifdef MI_STREAM_STAMPS
      ra.t.partials = stamp_buffer(); ra.t.part_stride = 0;      // this entry point has no workspace: a buffer of the build
#endif
      if (int rc = rstrm::launch(ra, pattern, (hipStream_t)stream)) return rc;
      if (sub) return ew::subsample(rgb, sub, Hd, Wd, sub_stride, work_dtype, (hipStream_t)stream);
      return 0;
    }
    MI_REQUIRE(rtile::scales_fit(scale, scale), "load_packed: scale %g is outside the fused kernel's range "
               "(mi_isp_load_packed_scale_supported); resize separately", (double)scale);
    rtile::RParams rp = {};
    rp.t = p; rp.Hd = Hd; rp.Wd = Wd; rp.s0 = scale; rp.s1 = scale;
    if (int rc = rtile::launch(rp, work_dtype, pattern, (hipStream_t)stream)) return rc;
    if (sub) return ew::subsample(rgb, sub, Hd, Wd, sub_stride, work_dtype, (hipStream_t)stream);
    return 0;
  }
  MI_REQUIRE(Hd == H && Wd == W, "load_packed: output shape must equal the frame when scale <= 0");
  p.vec_store = vec_store_ok(rgb, W, work_dtype);
  if (use_stream(p, work_dtype, rgb, work_dtype)) {
    strm::SArgs a = {};
    a.t = p;
    strm::geometry(H, W, a);
    if (sub && sub_stride == 8) { a.sub = sub; a.sub_w = (W + 7) / 8; }   // the metering subsample on the way
    if (int rc = strm::launch(a, work_dtype, pattern, strm::S_STORE, (hipStream_t)stream)) return rc;
    if (sub && sub_stride != 8) return ew::subsample(rgb, sub, H, W, sub_stride, work_dtype, (hipStream_t)stream);
    return 0;
  }
  if (int rc = tile::launch(p, work_dtype, pattern, tile::EPI_STORE, (hipStream_t)stream)) return rc;
  if (sub) return ew::subsample(rgb, sub, H, W, sub_stride, work_dtype, (hipStream_t)stream);
  return 0;
}

extern "C" int mi_isp_load_packed(const uint8_t* packed, void* rgb, int H, int W, int bits, int ids_format,
                                  int pattern, const float* ccm9, int work_dtype, int Hd, int Wd, float scale,
                                  void* stream) {
  return load_packed_impl(packed, rgb, H, W, bits, ids_format, pattern, ccm9, work_dtype, Hd, Wd, scale, nullptr, 0, stream);
}

// The cameras of a group in ONE launch per 8 (grid.y = camera): dispatch, decode table, first loads and drain are paid per
// launch instead of per camera (the load kernels take 23 - 30 us each, ~4 us of that is launch overhead: config 3, six
// cameras, 43.0 -> ~39.5 us per frame).  Same arithmetic, same bits as n calls of mi_isp_load_packed[_metered].
extern "C" int mi_isp_load_packed_batch(const uint8_t* const* packed, void* const* rgb, void* const* subs, int n, int H, int W,
                                        int bits, int ids_format, int pattern, const float* ccm9, int work_dtype, int Hd,
                                        int Wd, float scale, int sub_stride, void* stream) {
  MI_REQUIRE(packed && rgb, "load_packed_batch: null pointer");
  MI_REQUIRE(n >= 0, "load_packed_batch: negative frame count");
  for (int i = 0; i < n; ++i) MI_REQUIRE(packed[i] && rgb[i] && (!subs || subs[i]), "load_packed_batch: frame %d has a null buffer", i);
  if (n == 0) return 0;
  MI_REQUIRE(!subs || sub_stride >= 1, "load_packed_batch: bad subsample stride");
  // does every frame take the same streaming kernel?  (alignment is per buffer)
  tile::Params p0 = {};
  bool same = true, resize = scale > 0.f;
  for (int i = 0; i < n && same; ++i) {
    tile::Params p = {};
    if (int rc = fill_common(p, H, W, pattern, ccm9, "load_packed_batch")) return rc;
    if (int rc = packed_params(p, packed[i], H, W, bits, ids_format, work_dtype, "load_packed_batch")) return rc;
    p.dst = rgb[i]; p.out_dtype = work_dtype; p.out_scale = 1.f;
    if (resize) {
      same = Hd > 0 && Wd > 0 && H >= 2 && W >= 2 && use_stream(p, work_dtype, nullptr, work_dtype) &&
             rstrm::supported(p, work_dtype, rgb[i], Hd, Wd, scale, scale);
    } else {
      p.vec_store = vec_store_ok(rgb[i], W, work_dtype);
      same = Hd == H && Wd == W && use_stream(p, work_dtype, rgb[i], work_dtype) && (!subs || sub_stride == 8);
    }
    if (i == 0) p0 = p;
  }
  if (!same) {                                               // some frame needs another kernel: one by one
    for (int i = 0; i < n; ++i)
      if (int rc = load_packed_impl(packed[i], rgb[i], H, W, bits, ids_format, pattern, ccm9, work_dtype, Hd, Wd, scale,
                                    subs ? subs[i] : nullptr, sub_stride, stream))
        return rc;
    return 0;
  }
  for (int i0 = 0; i0 < n; i0 += strm::LOAD_BATCH) {
    const int m = n - i0 < strm::LOAD_BATCH ? n - i0 : strm::LOAD_BATCH;
    if (resize) {
      rstrm::RSArgs ra = {};
      ra.t = p0; ra.Hd = Hd; ra.Wd = Wd; ra.s0 = scale; ra.s1 = scale;
      rstrm::geometry(H, W, ra);
      ra.n_batch = m;
      for (int i = 0; i < m; ++i) { ra.srcs[i] = packed[i0 + i]; ra.dsts[i] = rgb[i0 + i]; }
#ifdef MI_STREAM_STAMPS
      ra.t.partials = stamp_buffer(); ra.t.part_stride = 0;
#endif
      if (int rc = rstrm::launch(ra, pattern, (hipStream_t)stream)) return rc;
      if (subs)
        for (int i = 0; i < m; ++i)
          if (int rc = ew::subsample(rgb[i0 + i], subs[i0 + i], Hd, Wd, sub_stride, work_dtype, (hipStream_t)stream)) return rc;
    } else {
      strm::SArgs a = {};
      a.t = p0;
      strm::geometry(H, W, a);
      a.n_batch = m; a.sub_w = (W + 7) / 8;
      for (int i = 0; i < m; ++i) { a.srcs[i] = packed[i0 + i]; a.dsts[i] = rgb[i0 + i]; a.subs[i] = subs ? subs[i0 + i] : nullptr; }
      if (int rc = strm::launch(a, work_dtype, pattern, strm::S_STORE, (hipStream_t)stream)) return rc;
    }
  }
  return 0;
}

extern "C" int mi_isp_load_packed_metered_is_fused(int H, int W, int bits, int ids_format, int work_dtype, int sub_stride) {
  if (bits != 12 || ids_format || sub_stride != 8 || H <= 0 || W <= 0) return 0;
  tile::Params p = {};
  p.H = H; p.W = W; p.src_kind = tile::SRC_PACKED12; p.src_fast = ((int64_t)W * 3 / 2) % 4 == 0; p.in_scale = 1.f;
  return strm::supported(p, work_dtype) && (int64_t)H * W * 3 * (int64_t)mi_dtype_size(work_dtype) < (int64_t)strm::INVALID_OFF ? 1 : 0;
}

extern "C" int mi_isp_load_packed_metered(const uint8_t* packed, void* rgb, int H, int W, int bits, int ids_format,
                                          int pattern, const float* ccm9, int work_dtype, int Hd, int Wd, float scale,
                                          void* sub, int sub_stride, void* stream) {
  MI_REQUIRE(sub && sub_stride >= 1, "load_packed_metered: need a subsample buffer and a positive stride");
  return load_packed_impl(packed, rgb, H, W, bits, ids_format, pattern, ccm9, work_dtype, Hd, Wd, scale, sub, sub_stride, stream);
}

extern "C" int mi_isp_load_packed_scale_supported(float scale) { return rtile::scales_fit(scale, scale) ? 1 : 0; }

// ---- measurement aid: HIP events around each data pass, on the stream it runs on ---------------------
#include <vector>
static struct {
  std::mutex mu;                  // the ABI is callable from several threads (one per stream)
  bool on = false;
  std::vector<hipEvent_t> ev;     // 8 per sampled frame: (start, stop) x 4 passes
  size_t used = 0;
  int every = 1;                  // every n-th frame is sampled (events between launches cost gaps)
  long frames_seen = 0;
  std::vector<int> npass;         // passes recorded per sampled frame (4; 1 for a whole-frame launch)
} g_prof;

extern "C" int mi_isp_profile_enable(int max_frames, int every) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.used = 0;
  g_prof.on = max_frames > 0;
  g_prof.every = every > 0 ? every : 1;
  g_prof.frames_seen = 0;
  g_prof.npass.assign(max_frames > 0 ? max_frames : 0, 4);
  for (int i = 0; i < 8 * max_frames; ++i) {
    hipEvent_t e;
    MI_HIP(hipEventCreate(&e));
    g_prof.ev.push_back(e);
  }
  return 0;
}

extern "C" int mi_isp_profile_collect(float avg_us[4], int* count) {
  MI_REQUIRE(avg_us && count, "profile_collect: null pointer");
  std::lock_guard<std::mutex> lock(g_prof.mu);
  double sum[4] = {0, 0, 0, 0};
  int n = 0;
  for (size_t f = 0; f + 8 <= g_prof.used; f += 8, ++n) {
    for (int k = 0; k < g_prof.npass[f / 8]; ++k) {
      float ms = 0.f;
      MI_HIP(hipEventSynchronize(g_prof.ev[f + 2 * k + 1]));
      MI_HIP(hipEventElapsedTime(&ms, g_prof.ev[f + 2 * k], g_prof.ev[f + 2 * k + 1]));
      sum[k] += ms * 1e3;
    }
  }
  for (int k = 0; k < 4; ++k) avg_us[k] = n ? (float)(sum[k] / n) : 0.f;
  *count = n;
  g_prof.used = 0;
  return 0;
}

// RAII-less helper: events of pass k of the frame whose slots start at `base` (or nothing)
struct PassTimer {
  size_t base; bool on; hipStream_t s;
  int begin(int k) const { if (on) MI_HIP(hipEventRecord(g_prof.ev[base + 2 * k], s)); return 0; }
  int end(int k) const { if (on) MI_HIP(hipEventRecord(g_prof.ev[base + 2 * k + 1], s)); return 0; }
};
static PassTimer pass_timer(hipStream_t s, int npass = 4) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (g_prof.on) (void)hipStreamIsCapturing(s, &cap);
  if (cap != hipStreamCaptureStatusNone) return PassTimer{0, false, s};   // events inside a graph cannot be timed
  const bool sampled = g_prof.on && (g_prof.frames_seen++ % g_prof.every) == 0;
  PassTimer t = {g_prof.used, sampled && g_prof.used + 8 <= g_prof.ev.size(), s};
  if (t.on) { g_prof.npass[g_prof.used / 8] = npass; g_prof.used += 8; }
  return t;
}

// One frame of the fused config-2 chain, recompute variant (output dtype != work dtype): four tile passes over
// the packed frame + three finalize launches.
static int pipeline_frame(tile::Params p, int pattern, int work_dtype, float intensity, float* ws, hipStream_t s) {
  float* fp = ws;
  float* partials = ws + FP_COUNT;
  const int cap = mi_partial_cap(p.H, p.W);
  const int nb = tile::num_tiles(p.H, p.W);
  p.fp = fp; p.partials = partials; p.part_stride = cap;
  ew::FinArgs fa = {};
  fa.partials = partials; fa.stride = cap; fa.nblocks = nb; fa.fp = fp;
  fa.n_px = (float)((int64_t)p.H * p.W); fa.intensity = intensity; fa.la = p.la; fa.ca = p.ca;
  fa.bounds_post = work_dtype == MI_F16 ? 2 : 1;
  const PassTimer tm = pass_timer(s);
  static const int epis[4] = {tile::EPI_MINMAX, tile::EPI_STATS, tile::EPI_RH_MINMAX, tile::EPI_RH_STORE};
  static const int fins[3] = {ew::FIN_BOUNDS, ew::FIN_STATS, ew::FIN_BOUNDS2};   // tonemap.py:146, :147-149, :150-153, :154
  for (int k = 0; k < 4; ++k) {
    if (int rc = tm.begin(k)) return rc;
    if (int rc = tile::launch(p, work_dtype, pattern, epis[k], s)) return rc;
    if (int rc = tm.end(k)) return rc;
    if (k < 3)
      if (int rc = ew::finalize(fins[k], fa, s)) return rc;
  }
  return 0;
}

// The "cached" variant of the same chain, used when the output has the work dtype (f16 -> f16,
// f32 -> f32): the first pass writes the demosaiced work-dtype image INTO THE OUTPUT BUFFER while
// reducing its bounds; the three tonemap passes then run elementwise on that image, the last one in
// place.  One demosaic instead of four (the path is vector-issue-bound) at the price
// of re-reading the 6 B/px image three times (it stays resident in the 256 MB Infinity Cache).
// which: -1 = the whole chain; 0..3 = only that data pass (measurement aid).
// `image`: where the work-dtype image lives between the passes (H * W * 3 work-dtype elements) - the output
// buffer itself when the output has the work dtype, a caller-provided buffer otherwise; `out` / `out_dtype`:
// the final destination of pass 3.
static int pipeline_frame_cached(tile::Params p, int pattern, int work_dtype, float gamma, float intensity,
                                 float* ws, int which, hipStream_t s, void* image, void* out, int out_dtype) {
  float* fp = ws;
  float* partials = ws + FP_COUNT;
  const int cap = mi_partial_cap(p.H, p.W);
  p.fp = fp; p.partials = partials; p.part_stride = cap;
  p.dst = image; p.vec_store = vec_store_ok(image, p.W, work_dtype);
  p.out_dtype = work_dtype; p.out_scale = 1.f;
  const PassTimer tm = which < 0 ? pass_timer(s) : PassTimer{0, false, s};
  if (int rc = tm.begin(0)) return rc;
  int n_bounds = tile::num_tiles(p.H, p.W);
  const bool stream0 = use_stream(p, work_dtype, image, work_dtype);
  if (stream0) {
    strm::SArgs a = {};
    a.t = p;
    strm::geometry(p.H, p.W, a);
    n_bounds = a.n_blocks;
    if (which < 0 || which == 0)
      if (int rc = strm::launch(a, work_dtype, pattern, strm::S_STORE_BOUNDS, s)) return rc;
  } else if (which < 0 || which == 0) {
    if (int rc = tile::launch(p, work_dtype, pattern, tile::EPI_STORE_MINMAX, s)) return rc;   // bayer.py + tonemap.py:146
  }
  if (int rc = tm.end(0)) return rc;
  if (which == 0) return 0;
  const ew::PullSrc bounds = {partials, cap, n_bounds, work_dtype == MI_F16 ? 2 : 1};
  if (which > 0)
    return ew::tonemap_reinhard_tail(image, out, p.H, p.W, work_dtype, out_dtype, gamma, intensity, p.la, p.ca, ws,
                                     which, bounds, s);
  for (int k = 1; k <= 3; ++k) {
    if (int rc = tm.begin(k)) return rc;
    if (int rc = ew::tonemap_reinhard_tail(image, out, p.H, p.W, work_dtype, out_dtype, gamma, intensity, p.la,
                                           p.ca, ws, k, bounds, s))
      return rc;
    if (int rc = tm.end(k)) return rc;
  }
  return 0;
}

// ---- the whole-frame kernel (isp_mega.h) --------------------------------------------------------------------------
// Two whole-frame grids must never share the chip: each needs every one of its blocks resident for its grid barriers,
// and two half-resident grids would wait for each other (the kernel's bounded poll would turn that into an error flag,
// not a hang - but the frames would be lost).  Launches on ONE stream are ordered by the stream.  Across streams the
// library orders them itself: under one lock per process it makes the new stream wait for an event the library
// recorded right behind the previous whole-frame launch, launches, and records that event again - the lock is held
// over all three, so two threads cannot interleave (round 2 released it before the launch: a second thread could
// record "done" ahead of the first thread's kernel).  The previous caller's stream handle is only ever compared, never
// used.  Launches captured by the CALLER into a graph of his own are not ordered by the library (nothing can be waited
// for at capture time): he must keep them off parallel branches, and replays of such a graph must not run beside
// direct launches on other streams.  Other PROCESSES on the same GPU are invisible to all of this: a foreign kernel that
// holds CUs makes the barrier time out - which is what the fault word, the mailbox and the multi-pass fallback are for.
// Round 4: the order, its lock, the event and the mailbox page are those of ALL resident-grid kernels of the library
// (ew::resident_order, isp_elementwise.h) - the one-launch metering kernel takes part in the same order.
static struct {
  int n_cus[16] = {};
  int per_cu[4] = {-1, -1, -1, -1};          // per CFA pattern: the allocator's outcome differs per instantiation
  unsigned poll_limit = 0;                   // 0 = default
  int sabotage_block = -1;                   // tests: this block never posts at barrier 0 of a launch's first frame
  std::atomic<uint32_t> launches{0};         // the host's part of a launch's tag (isp_mega.h)
} g_mega;
static inline std::mutex& mega_mu() { return ew::resident_order().mu; }

static bool mega_fits(const tile::Params& p, int work_dtype, const void* out, int out_dtype, int pattern, strm::SArgs& a) {
  if (work_dtype != MI_F16 || mi_dtype_size(out_dtype) > 2) return false;
  if (!use_stream(p, work_dtype, out, out_dtype) || !p.vec_store) return false;
  if (pattern < 0 || pattern > 3) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  std::lock_guard<std::mutex> lock(mega_mu());
  if (g_mega.per_cu[pattern] < 0) g_mega.per_cu[pattern] = mega::blocks_per_cu(pattern);
  if (g_mega.n_cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    g_mega.n_cus[dev] = n;
  }
  // fewer than two resident blocks per CU (a spilling or fatter instantiation) would deadlock the barrier: refuse
  return g_mega.per_cu[pattern] >= 2 && mega::geometry(p.H, p.W, g_mega.n_cus[dev], a);
}

// One launch for frames [0, n): same geometry and parameters, frame i reads srcs[i], writes dsts[i] and owns the workspace
// ws + i * ws_floats.  Ordered against the previous whole-frame launch of this process on this device (see above).
static int mega_launch_frames(tile::Params p, strm::SArgs a, int pattern, float intensity, const uint8_t* const* srcs,
                              void* const* dsts, float* ws, size_t ws_floats, int n, hipStream_t s) {
  p.src = nullptr; p.dst = nullptr; p.fp = nullptr; p.partials = nullptr;
  p.part_stride = mi_partial_cap(p.H, p.W);
  a.t = p;
  a.n_px = (float)((int64_t)p.H * p.W); a.intensity = intensity; a.fp_w = nullptr; a.bounds_post = 2;
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  MI_REQUIRE(dev >= 0 && dev < 16, "whole-frame kernel: device index %d out of range", dev);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  std::lock_guard<std::mutex> lock(mega_mu());
  ew::ResidentOrder& ord = ew::resident_order();
  if (int rc = ew::resident_mailbox_locked(dev)) return rc;
  mega::MBatch mb = {};
  mb.m.s = a;
  mb.m.spin_limit = g_mega.poll_limit ? g_mega.poll_limit : 100000;   // ~100 ms of polling before a wave gives up
  mb.m.l2_first = 1;
  mb.m.poll_sleep = 0;                                       // extra 512-cycle naps between two polls (swept: 0 is best)
  mb.m.mailbox = ord.mailbox_dev[dev] + ew::MAILBOX_WHOLE_FRAME;
  mb.m.sabotage_block = g_mega.sabotage_block;
#ifdef MI_ISP_MEASURE
  if (getenv("MI_ISP_POLL_SLEEP")) mb.m.poll_sleep = (unsigned)atoi(getenv("MI_ISP_POLL_SLEEP"));
  if (getenv("MI_ISP_L2_FIRST")) mb.m.l2_first = (unsigned)atoi(getenv("MI_ISP_L2_FIRST"));
#endif
  const bool direct = cap == hipStreamCaptureStatusNone;
  if (direct) { if (int rc = ew::resident_enter_locked(dev, s)) return rc; }
  const PassTimer tm0 = direct ? pass_timer(s, 1) : PassTimer{0, false, s};   // measurement aid: a launch as "pass 0"
  for (int i0 = 0; i0 < n; i0 += mega::MAX_BATCH) {
    mb.n_frames = n - i0 < mega::MAX_BATCH ? n - i0 : mega::MAX_BATCH;
    // the host's part of the launch's tag: a block of an EARLIER launch that comes to life late (a foreign kernel held its
    // CU) must not post records a later launch takes for its own (the workspace's own count covers graph replays, whose
    // arguments are frozen)
    mb.m.launch_id = g_mega.launches.fetch_add(1, std::memory_order_relaxed) + 1u;
    for (int i = 0; i < mb.n_frames; ++i) {
      mb.io[i].src = srcs[i0 + i];
      mb.io[i].dst = dsts[i0 + i];
      mb.io[i].ws = ws + (size_t)(i0 + i) * ws_floats;
    }
    // (events around the first launch of the call only: a call of more than 64 frames is several launches)
    const PassTimer tm = i0 == 0 ? tm0 : PassTimer{0, false, s};
    if (int rc = tm.begin(0)) return rc;
    if (int rc = mega::launch(mb, pattern, s)) return rc;
    if (int rc = tm.end(0)) return rc;
  }
  if (direct) { if (int rc = ew::resident_leave_locked(dev, s)) return rc; }
  return 0;
}

extern "C" size_t mi_isp_workspace_error_offset(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  return (size_t)mega::FP_ERROR * sizeof(float);
}

extern "C" int mi_isp_whole_frame_set_poll_limit(unsigned polls) {
  std::lock_guard<std::mutex> lock(mega_mu());
  g_mega.poll_limit = polls;
  return 0;
}

// Test hook: block `block` of every later whole-frame launch does not post its record at barrier 0 of the launch's first
// frame (-1: off) - the one fault a test can provoke that looks like a block which is not resident: everybody else waits
// the FULL poll budget for it.
extern "C" int mi_isp_whole_frame_set_sabotage(int block) {
  std::lock_guard<std::mutex> lock(mega_mu());
  g_mega.sabotage_block = block;
  return 0;
}

extern "C" int mi_isp_whole_frame_faults(int clear) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  std::lock_guard<std::mutex> lock(mega_mu());
  ew::ResidentOrder& ord = ew::resident_order();
  if (!ord.mailbox_host[dev]) return 0;
  volatile unsigned* mb = ord.mailbox_host[dev] + ew::MAILBOX_WHOLE_FRAME;
  const unsigned v = *mb;
  if (clear) *mb = 0;
  return (int)v;
}

extern "C" int mi_isp_workspace_check(void* ws_dev, int n_frames, int H, int W, int* failed_host, int* n_failed,
                                      void* stream) {
  MI_REQUIRE(ws_dev && n_failed, "workspace_check: null pointer");
  MI_REQUIRE(n_frames >= 0, "workspace_check: negative frame count");
  const size_t ws_bytes = mi_isp_workspace_bytes(H, W);
  MI_REQUIRE(ws_bytes > 0, "workspace_check: bad frame size");
  MI_HIP(hipStreamSynchronize((hipStream_t)stream));
  int bad = 0;
  for (int i = 0; i < n_frames; ++i) {
    unsigned* w = reinterpret_cast<unsigned*>(static_cast<char*>(ws_dev) + (size_t)i * ws_bytes) + mega::FP_ERROR;
    unsigned v = 0;
    MI_HIP(hipMemcpy(&v, w, sizeof(v), hipMemcpyDeviceToHost));
    if (failed_host) failed_host[i] = v != 0;
    if (v != 0) {
      ++bad;
      v = 0;
      MI_HIP(hipMemcpy(w, &v, sizeof(v), hipMemcpyHostToDevice));   // the word is sticky in the kernel: cleared here
      // and the barrier records of a failed frame are wiped: whatever a block that came to life late has left there can
      // then never carry the tag of a later launch (a graph replay repeats the host's part of the tag)
      float* partials = reinterpret_cast<float*>(static_cast<char*>(ws_dev) + (size_t)i * ws_bytes) + FP_COUNT;
      const size_t stride = (size_t)mi_partial_cap(H, W);
      MI_HIP(hipMemset(partials + (size_t)mega::MROW_BAR0 * stride, 0,
                       (size_t)(mega::MROW_END - mega::MROW_BAR0) * stride * sizeof(float)));
    }
  }
  *n_failed = bad;
  return 0;
}

// The same chain on the streaming kernels: every pass re-derives the demosaiced image from the packed frame
// (18.9 MB, served by L2 / Infinity Cache after the first pass) instead of writing and re-reading a 6 B/px
// intermediate; HBM sees the packed frame in and the output out.  Pass A (S_BOUNDS) also accumulates the statistics
// of tonemap.py:147-149 under the assumption that the bounds are exactly (0, 1) - true for every frame with a clipped
// pixel at both ends - and pass B (S_STATS) returns at once when that held, so the usual frame costs three data
// passes; any other frame gets the statistics from pass B.  The finalize steps are pulled into the consumers.
// which: -1 = whole chain, 0..3 = one pass (measurement aid; the partials of a full run must be in the workspace).
static int pipeline_frame_stream(tile::Params p, int pattern, int work_dtype, float intensity, float* ws, int which,
                                 hipStream_t s) {
  strm::SArgs a = {};
  p.fp = ws; p.partials = ws + FP_COUNT; p.part_stride = mi_partial_cap(p.H, p.W);
  a.t = p;
  strm::geometry(p.H, p.W, a);
  a.n_px = (float)((int64_t)p.H * p.W); a.intensity = intensity; a.fp_w = ws;
  a.bounds_post = work_dtype == MI_F16 ? 2 : 1;
  static const int epis[4] = {strm::S_BOUNDS, strm::S_STATS, strm::S_RH_MINMAX, strm::S_RH_STORE};
  const PassTimer tm = which < 0 ? pass_timer(s) : PassTimer{0, false, s};
  for (int k = 0; k < 4; ++k) {
    if (which >= 0 && which != k) continue;
    if (int rc = tm.begin(k)) return rc;
    if (int rc = strm::launch(a, work_dtype, pattern, epis[k], s)) return rc;
    if (int rc = tm.end(k)) return rc;
  }
  return 0;
}

// measurement aid: MI_ISP_PIPELINE=cached times the store-and-re-read chain on the same build
static bool force_cached() {
#ifdef MI_ISP_MEASURE
  static const bool on = getenv("MI_ISP_PIPELINE") && !strcmp(getenv("MI_ISP_PIPELINE"), "cached");
  return on;
#else
  return false;
#endif
}

static bool no_cached_pipeline() {
#ifdef MI_ISP_MEASURE
  static const bool off = getenv("MI_ISP_NO_CACHED_PIPELINE") != nullptr;   // measurement aid: the recompute tile chain
  return off;
#else
  return false;
#endif
}

static bool use_cached(const tile::Params& p, int work_dtype, int out_dtype) {
  return !no_cached_pipeline() && work_dtype == out_dtype && p.vec_store;
}

static int pipeline_params(tile::Params& p, int H, int W, int ids_format, int pattern, const float* ccm9,
                           int work_dtype, int out_dtype, float gamma, float la, float ca) {
  if (int rc = fill_common(p, H, W, pattern, ccm9, "pipeline12_reinhard")) return rc;
  MI_REQUIRE(mi_valid_dtype(out_dtype), "pipeline12_reinhard: bad output dtype");
  MI_REQUIRE(gamma > 0.f, "pipeline12_reinhard: gamma must be positive");
  MI_REQUIRE(work_dtype == MI_F16 || work_dtype == MI_F32, "pipeline12_reinhard: work dtype must be f16/f32");
  (void)ids_format;
  p.out_dtype = out_dtype;
  p.out_scale = mi_scale_factor(out_dtype);
  p.gamma_inv = 1.0f / gamma; p.la = la; p.ca = ca;
  return 0;
}

// One frame: the cached variant when there is a place for the work-dtype image (the output itself when it has the
// work dtype, else `work_image`), the recompute variant otherwise.
// whole_frame: 0 = the multi-pass chain, 1 = the single-launch whole-frame kernel (an error when the frame does not fit)
static int pipeline12_frame(const uint8_t* packed, void* out, void* work_image, int H, int W, int ids_format,
                            int pattern, const float* ccm9, int work_dtype, int out_dtype, float gamma,
                            float intensity, float light_adapt, float color_adapt, float* ws, hipStream_t s,
                            const char* who, int whole_frame = 0) {
  tile::Params p = {};
  if (int rc = pipeline_params(p, H, W, ids_format, pattern, ccm9, work_dtype, out_dtype, gamma, light_adapt,
                               color_adapt))
    return rc;
  if (int rc = packed_params(p, packed, H, W, 12, ids_format, work_dtype, who)) return rc;
  p.dst = out;
  p.vec_store = vec_store_ok(out, W, out_dtype);
  if (whole_frame) {
    strm::SArgs ma = {};
    MI_REQUIRE(mega_fits(p, work_dtype, out, out_dtype, pattern, ma),
               "%s: the whole-frame kernel takes f16 work dtype, u8 / u16 / f16 outputs, the standard 12-bit layout with "
               "W %% 8 == 0 and 16-byte aligned buffers, and at most 2 x CUs x 4 waves of 512 x 12 pixels (4096 x 3072 on "
               "MI355X); use mi_isp_pipeline12_reinhard for this frame", who);
    const uint8_t* srcs[1] = {packed};
    void* dsts[1] = {out};
    return mega_launch_frames(p, ma, pattern, intensity, srcs, dsts, ws, 0, 1, s);
  }
  if (use_stream(p, work_dtype, out, out_dtype) && p.vec_store && !force_cached())
    return pipeline_frame_stream(p, pattern, work_dtype, intensity, ws, -1, s);
  void* image = work_dtype == out_dtype ? out : work_image;
  if (!no_cached_pipeline() && image && vec_store_ok(image, W, work_dtype))
    return pipeline_frame_cached(p, pattern, work_dtype, gamma, intensity, ws, -1, s, image, out, out_dtype);
  p.dst = out;
  p.vec_store = vec_store_ok(out, W, out_dtype);
  return pipeline_frame(p, pattern, work_dtype, intensity, ws, s);
}

extern "C" int mi_isp_pipeline12_reinhard(const uint8_t* packed, void* out, void* work_image, int H, int W,
                                          int ids_format, int pattern, const float* ccm9, int work_dtype,
                                          int out_dtype, float gamma, float intensity, float light_adapt,
                                          float color_adapt, void* ws, void* stream) {
  MI_REQUIRE(out && ws, "pipeline12_reinhard: null pointer");
  return pipeline12_frame(packed, out, work_image, H, W, ids_format, pattern, ccm9, work_dtype, out_dtype, gamma,
                          intensity, light_adapt, color_adapt, static_cast<float*>(ws), (hipStream_t)stream,
                          "pipeline12_reinhard");
}

extern "C" int mi_isp_pipeline12_reinhard_whole_frame(const uint8_t* packed, void* out, int H, int W, int ids_format,
                                                      int pattern, const float* ccm9, int out_dtype, float gamma,
                                                      float intensity, float light_adapt, float color_adapt, void* ws,
                                                      void* stream) {
  MI_REQUIRE(out && ws, "pipeline12_reinhard_whole_frame: null pointer");
  return pipeline12_frame(packed, out, nullptr, H, W, ids_format, pattern, ccm9, MI_F16, out_dtype, gamma, intensity,
                          light_adapt, color_adapt, static_cast<float*>(ws), (hipStream_t)stream,
                          "pipeline12_reinhard_whole_frame", 1);
}

extern "C" int mi_isp_pipeline12_whole_frame_fits(int H, int W, int out_dtype) {
  tile::Params p = {};
  p.H = H; p.W = W; p.src_kind = tile::SRC_PACKED12; p.src_fast = 1; p.in_scale = 1.f; p.vec_store = 1;
  strm::SArgs a = {};
  // asked without a pattern: all four instantiations must be launchable
  if (!(H > 0 && W > 0 && H % 2 == 0 && mi_valid_dtype(out_dtype))) return 0;
  for (int pat = 0; pat < 4; ++pat)
    if (!mega_fits(p, MI_F16, nullptr, out_dtype, pat, a)) return 0;
  return 1;
}

// n_frames frames through ONE launch of the whole-frame kernel per 64 frames (isp_mega.h: the grid stays resident and
// walks through the frames), in order, on `stream`.
extern "C" int mi_isp_pipeline12_reinhard_whole_frame_batch(const uint8_t* const* packed, void* const* out, int n_frames,
                                                            int H, int W, int ids_format, int pattern, const float* ccm9,
                                                            int out_dtype, float gamma, float intensity, float light_adapt,
                                                            float color_adapt, void* ws, void* stream) {
  const char* who = "pipeline12_reinhard_whole_frame_batch";
  MI_REQUIRE(packed && out && ws, "%s: null pointer", who);
  MI_REQUIRE(n_frames >= 1, "%s: need at least one frame", who);
  tile::Params p = {};
  strm::SArgs ma = {};
  for (int i = 0; i < n_frames; ++i) {
    MI_REQUIRE(packed[i] && out[i], "%s: frame %d has a null buffer", who, i);
    tile::Params pi = {};
    if (int rc = pipeline_params(pi, H, W, ids_format, pattern, ccm9, MI_F16, out_dtype, gamma, light_adapt, color_adapt)) return rc;
    if (int rc = packed_params(pi, packed[i], H, W, 12, ids_format, MI_F16, who)) return rc;
    pi.dst = out[i];
    pi.vec_store = vec_store_ok(out[i], W, out_dtype);
    MI_REQUIRE(mega_fits(pi, MI_F16, out[i], out_dtype, pattern, ma),
               "%s: frame %d does not fit the whole-frame kernel (see mi_isp_pipeline12_reinhard_whole_frame)", who, i);
    if (i == 0) p = pi;
  }
  return mega_launch_frames(p, ma, pattern, intensity, packed, out, static_cast<float*>(ws),
                            mi_isp_workspace_bytes(H, W) / sizeof(float), n_frames, (hipStream_t)stream);
}

extern "C" int mi_isp_pipeline12_reinhard_batch(const uint8_t* const* packed, void* const* out,
                                                void* const* work_images, int n_frames, int H, int W,
                                                int ids_format, int pattern, const float* ccm9, int work_dtype,
                                                int out_dtype, float gamma, float intensity, float light_adapt,
                                                float color_adapt, void* ws, void* const* streams, int n_streams) {
  MI_REQUIRE(packed && out && ws, "pipeline12_reinhard_batch: null pointer");
  MI_REQUIRE(n_frames >= 0, "pipeline12_reinhard_batch: negative frame count");
  MI_REQUIRE(n_streams >= 1 && streams, "pipeline12_reinhard_batch: need at least one stream");
  const size_t ws_floats = mi_isp_workspace_bytes(H, W) / sizeof(float);
  for (int i = 0; i < n_frames; ++i) {
    MI_REQUIRE(out[i], "pipeline12_reinhard_batch: output %d is null", i);
    float* wsi = static_cast<float*>(ws) + (size_t)i * ws_floats;
    if (int rc = pipeline12_frame(packed[i], out[i], work_images ? work_images[i] : nullptr, H, W, ids_format, pattern,
                                  ccm9, work_dtype, out_dtype, gamma, intensity, light_adapt, color_adapt, wsi,
                                  (hipStream_t)streams[i % n_streams], "pipeline12_reinhard_batch"))
      return rc;
  }
  return 0;
}

// ---- one camera group, packed bytes -> u8 outputs, in one call -----------------------------------------------------------
// ISP.load_packed12 / load_packed16 per camera (camera_isp.py:333-347, resize fused when scale > 0), the rolling
// metering over the group (:376-385, :142-175), then ISP.tonemap_reinhard or tonemap_linear (:394-413) with the
// orientation transform folded into the u8 store: what a frame group costs a C caller is this one call on its stream.
extern "C" int mi_isp_metering(const void* const* images, int n_images, int H, int W, int stride, int dtype, float* state9,
                               float alpha, void* ws, void* stream);
extern "C" int mi_isp_metering_to(const void* const* images, int n_images, int H, int W, int stride, int dtype,
                                  const float* prev9, float* state9, float alpha, void* ws, void* stream);
extern "C" int mi_isp_reinhard_batch(void* const* images, uint8_t* const* outs, int n, int H, int W, int dtype,
                                     const float* state9, float gamma, float intensity, float light_adapt,
                                     float color_adapt, int transform, void* ws, void* stream);
extern "C" int mi_isp_linear_batch(const void* const* images, uint8_t* const* outs, int n, int H, int W, int dtype,
                                   const float* state9, float gamma, int transform, void* ws, void* stream);

extern "C" int mi_isp_camera_frame_batch(const uint8_t* const* packed, void* const* images, uint8_t* const* outs, int n,
                                         int H, int W, int bits, int ids_format, int pattern, const float* ccm9,
                                         int work_dtype, int Hd, int Wd, float scale, int metering_stride,
                                         float* state9, float alpha, int tonemap, float gamma, float intensity,
                                         float light_adapt, float color_adapt, int transform, void* ws, void* stream) {
  MI_REQUIRE(packed && images && outs && state9 && ws, "camera_frame_batch: null pointer");
  MI_REQUIRE(n >= 1, "camera_frame_batch: need at least one camera");
  MI_REQUIRE(tonemap == 0 || tonemap == 1, "camera_frame_batch: tonemap must be 0 (reinhard) or 1 (linear)");
  MI_REQUIRE(metering_stride >= 1, "camera_frame_batch: bad metering stride");
  for (int i = 0; i < n; ++i) MI_REQUIRE(packed[i] && images[i] && outs[i], "camera_frame_batch: camera %d has a null buffer", i);
  if (int rc = mi_isp_load_packed_batch(packed, images, nullptr, n, H, W, bits, ids_format, pattern, ccm9, work_dtype, Hd, Wd,
                                        scale, 0, stream))
    return rc;
  if (int rc = mi_isp_metering(const_cast<const void* const*>(images), n, Hd, Wd, metering_stride, work_dtype, state9,
                               alpha, ws, stream))
    return rc;
  if (tonemap == 0)
    return mi_isp_reinhard_batch(images, outs, n, Hd, Wd, work_dtype, state9, gamma, intensity, light_adapt, color_adapt,
                                 transform, ws, stream);
  return mi_isp_linear_batch(const_cast<const void* const*>(images), outs, n, Hd, Wd, work_dtype, state9, gamma, transform,
                             ws, stream);
}

// ---- one camera group at full resolution: subsample, metering, ONE persistent launch (isp_mega_cam.h) -------------------
// What the reference's bench does per step (bench/camera_isp.py:19-28: load_packed12 per camera, tonemap_reinhard over the
// list, the loaded images dropped): the stride-8 subsample of every camera straight from its packed frame
// (strm::sub_kernel), the rolling metering over the subsamples (mi_isp_metering: camera_isp.py:376-385 -> :142-175), then
// mega::camera_kernel walks through the cameras - demosaic, Reinhard, max_out at a grid barrier, u8 out - with the image
// resident on the chip.  images == NULL: p is not stored (the bench drops it); else images[i] receives what the reference
// leaves in the loaded image (camera_isp.py:211).
static int g_cam_per_cu[4] = {-1, -1, -1, -1};
static unsigned g_cam_poll_limit = 0;

static bool camera_group_fits(int H, int W, int pattern, strm::SArgs& a) {
  if (pattern < 0 || pattern > 3 || H <= 0 || W <= 0) return false;
  tile::Params p = {};
  p.H = H; p.W = W; p.src_kind = tile::SRC_PACKED12; p.src_fast = ((int64_t)W * 3 / 2) % 4 == 0; p.in_scale = 1.f; p.vec_store = 1;
  if (!strm::supported(p, MI_F16) || (int64_t)H * W * 6 >= (int64_t)strm::INVALID_OFF) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  std::lock_guard<std::mutex> lock(mega_mu());
  if (g_cam_per_cu[pattern] < 0) g_cam_per_cu[pattern] = mega::cam_blocks_per_cu(pattern);
  if (g_mega.n_cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    g_mega.n_cus[dev] = n;
  }
  return g_cam_per_cu[pattern] >= 2 && mega::geometry(H, W, g_mega.n_cus[dev], a);
}

extern "C" int mi_isp_camera_group_fits(int H, int W, int pattern, int work_dtype, int metering_stride) {
  strm::SArgs a = {};
  return work_dtype == MI_F16 && metering_stride == 8 && camera_group_fits(H, W, pattern, a) ? 1 : 0;
}

extern "C" size_t mi_isp_camera_group_scratch_bytes(int n, int H, int W) {
  if (n <= 0 || H <= 0 || W <= 0) return 0;
  const size_t per = (size_t)((H + 7) / 8) * (size_t)((W + 7) / 8) * 3 * sizeof(half_t);
  return (size_t)n * ((per + 255) / 256 * 256);
}

extern "C" int mi_isp_camera_group_set_poll_limit(unsigned polls) {
  std::lock_guard<std::mutex> lock(mega_mu());
  g_cam_poll_limit = polls;
  return 0;
}

extern "C" int mi_isp_camera_group_faults(int clear) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  std::lock_guard<std::mutex> lock(mega_mu());
  ew::ResidentOrder& ord = ew::resident_order();
  if (!ord.mailbox_host[dev]) return 0;
  volatile unsigned* mb = ord.mailbox_host[dev] + ew::MAILBOX_CAMERA_GROUP;
  const unsigned v = *mb;
  if (clear) *mb = 0;
  return (int)v;
}

// the frames' common parameters, checked per camera
static int camera_group_params(tile::Params& p, const uint8_t* const* packed, void* const* images, uint8_t* const* outs, int n,
                               int H, int W, int pattern, const float* ccm9, const char* who) {
  MI_REQUIRE(packed, "%s: null pointer", who);
  MI_REQUIRE(n >= 1 && n <= mega::MAX_BATCH, "%s: 1 .. %d cameras per call", who, mega::MAX_BATCH);
  for (int i = 0; i < n; ++i) {
    MI_REQUIRE(packed[i] && (!outs || outs[i]) && (!images || images[i]), "%s: camera %d has a null buffer", who, i);
    MI_REQUIRE((!outs || ((uintptr_t)outs[i] & 7) == 0) && (!images || ((uintptr_t)images[i] & 15) == 0),
               "%s: camera %d: outputs must be 8-byte, images 16-byte aligned", who, i);
    tile::Params pi = {};
    if (int rc = fill_common(pi, H, W, pattern, ccm9, who)) return rc;
    if (int rc = packed_params(pi, packed[i], H, W, 12, 0, MI_F16, who)) return rc;
    MI_REQUIRE(strm::supported(pi, MI_F16), "%s: camera %d: the packed frame does not take the streaming kernels "
               "(standard 12-bit layout, W %% 8 == 0, even H, 4-byte aligned rows)", who, i);
    if (i == 0) p = pi;
  }
  p.src = nullptr; p.dst = nullptr; p.fp = nullptr; p.partials = nullptr;
  return 0;
}

// step 1: image[::8, ::8] of every camera's (never materialised) image, (ceil(H / 8), ceil(W / 8), 3) f16 each, in scratch
extern "C" int mi_isp_camera_group_subsample(const uint8_t* const* packed, int n, int H, int W, int pattern, const float* ccm9,
                                             void* scratch, void* stream) {
  const char* who = "camera_group_subsample";
  MI_REQUIRE(scratch, "%s: null pointer", who);
  tile::Params p = {};
  if (int rc = camera_group_params(p, packed, nullptr, nullptr, n, H, W, pattern, ccm9, who)) return rc;
  const size_t sub_bytes = mi_isp_camera_group_scratch_bytes(1, H, W);
  for (int i0 = 0; i0 < n; i0 += strm::LOAD_BATCH) {
    strm::SubArgs sa = {};
    sa.t = p;
    strm::sub_geometry(H, W, sa);
    sa.n_batch = n - i0 < strm::LOAD_BATCH ? n - i0 : strm::LOAD_BATCH;
    for (int i = 0; i < sa.n_batch; ++i) { sa.srcs[i] = packed[i0 + i]; sa.subs[i] = static_cast<char*>(scratch) + (size_t)(i0 + i) * sub_bytes; }
    if (int rc = strm::launch_sub(sa, MI_F16, pattern, (hipStream_t)stream)) return rc;
  }
  return 0;
}

// step 3: the cameras through one resident launch, with the Reinhard scalars of state9 (read on the device)
extern "C" int mi_isp_camera_group_tonemap(const uint8_t* const* packed, void* const* images, uint8_t* const* outs, int n, int H,
                                           int W, int pattern, const float* ccm9, const float* state9, float gamma,
                                           float intensity, float light_adapt, float color_adapt, void* ws, void* stream) {
  const char* who = "camera_group_tonemap";
  MI_REQUIRE(outs && state9 && ws, "%s: null pointer", who);
  MI_REQUIRE(gamma > 0.f, "%s: gamma must be positive", who);
  hipStream_t s = (hipStream_t)stream;
  tile::Params p = {};
  if (int rc = camera_group_params(p, packed, images, outs, n, H, W, pattern, ccm9, who)) return rc;
  strm::SArgs ma = {};
  MI_REQUIRE(camera_group_fits(H, W, pattern, ma),
             "%s: the frame does not fit the resident grid (mi_isp_camera_group_fits); use mi_isp_camera_frame_batch", who);
  p.out_dtype = MI_U8; p.out_scale = 255.f; p.gamma_inv = 1.0f / gamma; p.la = light_adapt; p.ca = color_adapt;
  p.part_stride = mi_partial_cap(H, W);
  ma.t = p;
  ma.n_px = (float)((int64_t)H * W); ma.intensity = intensity; ma.fp_w = nullptr; ma.bounds_post = 0;
  const size_t ws_floats = mi_isp_workspace_bytes(H, W) / sizeof(float);
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  MI_REQUIRE(dev >= 0 && dev < 16, "%s: device index %d out of range", who, dev);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  std::lock_guard<std::mutex> lock(mega_mu());
  ew::ResidentOrder& ord = ew::resident_order();
  if (int rc = ew::resident_mailbox_locked(dev)) return rc;
  mega::CBatch cb = {};
  cb.m.s = ma;
  cb.m.spin_limit = g_cam_poll_limit ? g_cam_poll_limit : 100000;
  cb.m.l2_first = 1;
  cb.m.poll_sleep = 0;
  cb.m.mailbox = ord.mailbox_dev[dev] + ew::MAILBOX_CAMERA_GROUP;
  cb.m.sabotage_block = -1;
  cb.m.launch_id = g_mega.launches.fetch_add(1, std::memory_order_relaxed) + 1u;
  cb.state9 = state9;
  cb.gamma_inv = 1.0f / gamma;
  cb.n_frames = n;
  for (int i = 0; i < n; ++i) {
    cb.io[i].src = packed[i];
    cb.io[i].p_out = images ? images[i] : nullptr;
    cb.io[i].out = outs[i];
    cb.io[i].ws = static_cast<float*>(ws) + (size_t)i * ws_floats;
  }
  const bool direct = cap == hipStreamCaptureStatusNone;
  if (direct) { if (int rc = ew::resident_enter_locked(dev, s)) return rc; }
  if (int rc = mega::launch_cam(cb, pattern, s)) return rc;
  if (direct) { if (int rc = ew::resident_leave_locked(dev, s)) return rc; }
  return 0;
}

extern "C" int mi_isp_camera_group_reinhard(const uint8_t* const* packed, void* const* images, uint8_t* const* outs, int n,
                                            int H, int W, int pattern, const float* ccm9, const float* prev9, float* state9,
                                            float alpha, float gamma, float intensity, float light_adapt, float color_adapt,
                                            void* scratch, void* ws, void* stream) {
  const char* who = "camera_group_reinhard";
  MI_REQUIRE(packed && outs && prev9 && state9 && scratch && ws, "%s: null pointer", who);
  MI_REQUIRE(n >= 1 && n <= mega::MAX_BATCH, "%s: 1 .. %d cameras per call", who, mega::MAX_BATCH);
  {                                                          // refuse before anything is launched
    strm::SArgs ma = {};
    MI_REQUIRE(camera_group_fits(H, W, pattern, ma),
               "%s: the frame does not fit the resident grid (mi_isp_camera_group_fits); use mi_isp_camera_frame_batch", who);
  }
  // 1. the subsample of every camera, straight from its packed frame
  if (int rc = mi_isp_camera_group_subsample(packed, n, H, W, pattern, ccm9, scratch, stream)) return rc;
  // 2. the rolling metering over the group (its own workspace: the last of the n + 1)
  const int Hs = (H + 7) / 8, Ws = (W + 7) / 8;
  const size_t sub_bytes = mi_isp_camera_group_scratch_bytes(1, H, W);
  const void* subs[mega::MAX_BATCH];
  for (int i = 0; i < n; ++i) subs[i] = static_cast<char*>(scratch) + (size_t)i * sub_bytes;
  const size_t ws_floats = mi_isp_workspace_bytes(H, W) / sizeof(float);
  float* ws_meter = static_cast<float*>(ws) + (size_t)n * ws_floats;
  if (int rc = mi_isp_metering_to(subs, n, Hs, Ws, 1, MI_F16, prev9, state9, alpha, ws_meter, stream)) return rc;
  // 3. the cameras through one resident launch
  return mi_isp_camera_group_tonemap(packed, images, outs, n, H, W, pattern, ccm9, state9, gamma, intensity, light_adapt,
                                     color_adapt, ws, stream);
}

// ---- a batch as a HIP graph: capture once, replay per step ---------------------------------------------------------
// What BatchPipeline(use_graph=True) has, for C callers: the step - fork to `n_streams` internal streams, the launches of
// every frame (frame i on stream i % n_streams), join - is captured into a graph bound to the given buffers; a replay
// has no launch gaps between the dependent kernels of a stream.  whole_frame: every frame through the single-launch
// kernel, one after the other on one stream (two of them must not overlap).
struct BatchGraph {
  bool whole_frame = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> events;
};

static void batch_graph_free(BatchGraph* b) {
  if (!b) return;
  if (b->exec) (void)hipGraphExecDestroy(b->exec);
  if (b->graph) (void)hipGraphDestroy(b->graph);
  for (hipEvent_t e : b->events) (void)hipEventDestroy(e);
  for (hipStream_t s : b->streams) (void)hipStreamDestroy(s);
  delete b;
}

extern "C" int mi_isp_pipeline12_whole_frame_fits(int H, int W, int out_dtype);
extern "C" int mi_isp_pipeline12_graph_create(const uint8_t* const* packed, void* const* out, void* const* work_images,
                                              int n_frames, int H, int W, int ids_format, int pattern,
                                              const float* ccm9, int work_dtype, int out_dtype, float gamma,
                                              float intensity, float light_adapt, float color_adapt, void* ws,
                                              int n_streams, int whole_frame, void** handle) {
  MI_REQUIRE(packed && out && ws && handle, "pipeline12_graph_create: null pointer");
  MI_REQUIRE(n_frames > 0 && n_streams >= 1, "pipeline12_graph_create: need at least one frame and one stream");
  if (whole_frame) {
    n_streams = 1;
    (void)mi_isp_pipeline12_whole_frame_fits(H, W, out_dtype);   // device queries happen outside the capture
  }
  BatchGraph* b = new BatchGraph();
  b->whole_frame = whole_frame != 0;
  bool capturing = false;
  auto fail = [&](int rc) {
    if (capturing) {                                         // a capture must be ended (and its graph dropped) before its stream goes
      hipGraph_t dead = nullptr;
      (void)hipStreamEndCapture(b->streams[0], &dead);
      if (dead) (void)hipGraphDestroy(dead);
      (void)hipGetLastError();
    }
    batch_graph_free(b);
    return rc;
  };
#define MI_HIP_G(expr)                                                                              \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      mi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);     \
      return fail(2);                                                                               \
    }                                                                                               \
  } while (0)
  for (int i = 0; i < n_streams; ++i) {
    hipStream_t s;
    MI_HIP_G(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    b->streams.push_back(s);
  }
  for (int i = 0; i < n_streams; ++i) {
    hipEvent_t e;
    MI_HIP_G(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    b->events.push_back(e);
  }
  hipStream_t s0 = b->streams[0];
  MI_HIP_G(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
  capturing = true;
  MI_HIP_G(hipEventRecord(b->events[0], s0));                                   // fork
  for (int i = 1; i < n_streams; ++i) MI_HIP_G(hipStreamWaitEvent(b->streams[i], b->events[0], 0));
  const size_t ws_floats = mi_isp_workspace_bytes(H, W) / sizeof(float);
  int rc = 0;
  if (whole_frame) {
    rc = mi_isp_pipeline12_reinhard_whole_frame_batch(packed, out, n_frames, H, W, ids_format, pattern, ccm9, out_dtype, gamma,
                                                      intensity, light_adapt, color_adapt, ws, s0);
  } else {
    for (int i = 0; i < n_frames && rc == 0; ++i) {
      float* wsi = static_cast<float*>(ws) + (size_t)i * ws_floats;
      rc = pipeline12_frame(packed[i], out[i], work_images ? work_images[i] : nullptr, H, W, ids_format, pattern, ccm9,
                            work_dtype, out_dtype, gamma, intensity, light_adapt, color_adapt, wsi,
                            b->streams[i % n_streams], "pipeline12_graph_create", 0);
    }
  }
  for (int i = 1; i < n_streams; ++i) {                                          // join
    (void)hipEventRecord(b->events[i], b->streams[i]);
    (void)hipStreamWaitEvent(s0, b->events[i], 0);
  }
  hipGraph_t g = nullptr;
  const hipError_t ec = hipStreamEndCapture(s0, &g);
  capturing = false;
  b->graph = g;
  if (rc != 0) return fail(rc);
  if (ec != hipSuccess) { mi_set_error("pipeline12_graph_create: capture failed: %s", hipGetErrorString(ec)); return fail(2); }
  MI_HIP_G(hipGraphInstantiate(&b->exec, b->graph, nullptr, nullptr, 0));
#undef MI_HIP_G
  *handle = b;
  return 0;
}

extern "C" int mi_isp_pipeline12_graph_launch(void* handle, void* stream) {
  MI_REQUIRE(handle, "pipeline12_graph_launch: null handle");
  BatchGraph* b = static_cast<BatchGraph*>(handle);
  if (b->whole_frame) {                                      // its grids need the chip to themselves, like a direct launch
    int dev = 0;
    MI_HIP(hipGetDevice(&dev));
    MI_REQUIRE(dev >= 0 && dev < 16, "pipeline12_graph_launch: device index %d out of range", dev);
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(mega_mu());             // order, launch and record under one lock (see g_mega)
    if (int rc = ew::resident_enter_locked(dev, s)) return rc;
    MI_HIP(hipGraphLaunch(b->exec, s));
    return ew::resident_leave_locked(dev, s);
  }
  MI_HIP(hipGraphLaunch(b->exec, (hipStream_t)stream));
  return 0;
}

extern "C" int mi_isp_pipeline12_graph_destroy(void* handle) {
  batch_graph_free(static_cast<BatchGraph*>(handle));
  return 0;
}

extern "C" int mi_isp_pipeline12_pass(const uint8_t* packed, void* out, int H, int W, int ids_format, int pattern,
                                      const float* ccm9, int work_dtype, int out_dtype, float gamma,
                                      float light_adapt, float color_adapt, int pass, void* ws, void* stream) {
  MI_REQUIRE(out && ws, "pipeline12_pass: null pointer");
#ifdef MI_ISP_MEASURE
  const int debug_skip = pass >> 4;      // measurement aid (see tile::Params::debug_skip); MI_ISP_MEASURE builds only
#else
  const int debug_skip = 0;
#endif
  pass &= 15;
  MI_REQUIRE(pass >= 0 && pass <= 3, "pipeline12_pass: pass must be 0..3");
  tile::Params p = {};
  if (int rc = pipeline_params(p, H, W, ids_format, pattern, ccm9, work_dtype, out_dtype, gamma, light_adapt,
                               color_adapt))
    return rc;
  if (int rc = packed_params(p, packed, H, W, 12, ids_format, work_dtype, "pipeline12_pass")) return rc;
  p.dst = out;
  p.vec_store = vec_store_ok(out, W, out_dtype);
  float* fp = static_cast<float*>(ws);
  if (use_stream(p, work_dtype, out, out_dtype) && p.vec_store && debug_skip == 0 && !force_cached())
    return pipeline_frame_stream(p, pattern, work_dtype, 1.0f, fp, pass, (hipStream_t)stream);
  if (use_cached(p, work_dtype, out_dtype) && debug_skip == 0)
    return pipeline_frame_cached(p, pattern, work_dtype, gamma, 1.0f, fp, pass, (hipStream_t)stream, out, out, out_dtype);
  p.fp = fp; p.partials = fp + FP_COUNT; p.part_stride = mi_partial_cap(H, W);
  p.debug_skip = debug_skip & 63;
  static const int epi[4] = {tile::EPI_MINMAX, tile::EPI_STATS, tile::EPI_RH_MINMAX, tile::EPI_RH_STORE};
  const int e = (debug_skip & 64) ? tile::EPI_STORE_MINMAX : epi[pass];   // bit 64: the cached pipeline's pass 0
  return tile::launch(p, work_dtype, pattern, e, (hipStream_t)stream);
}

namespace tile { int occupancy_rggb(int epi); }
extern "C" int mi_isp_debug_occupancy(int epi) { return tile::occupancy_rggb(epi); }
