/*
 * mi_isp.h -- C ABI of libmi355_isp.so: the MI355X (gfx950) camera-ISP hot path.
 *
 * This is the drop-in boundary.  The reference (uc-vision/taichi_image) exposes this path as
 * Python functions that launch Taichi-JIT kernels on torch/numpy buffers; there is no FFI in
 * the reference, so each entry point below cites the reference Python function / Taichi
 * kernel it replaces (paths relative to /root/reference/taichi_image/).  A maintainer of the
 * reference would bind these with ctypes exactly as taichi_image_amd/_native.py does
 * (see INTEGRATION.md).
 *
 * Conventions
 *  - every pointer named *_dev is a device (HBM) pointer owned by the caller; the library never
 *    allocates or frees user-visible memory and never synchronises the stream;
 *  - images are C-contiguous, [row][col] or [row][col][3] interleaved RGB;
 *  - `stream` is a hipStream_t (NULL = the default stream); all work is stream-ordered;
 *  - every function returns 0 on success, non-zero on failure; the message for the calling
 *    thread is available from mi_isp_last_error();
 *  - `ws_dev` is a scratch buffer of at least mi_isp_workspace_bytes(H, W) bytes, private to
 *    one in-flight call (use one per stream / per frame in flight).  It must be ZERO-FILLED once
 *    before its first use (hipMemset): the whole-frame kernel of mi_isp_pipeline12_reinhard keeps the
 *    launch count of the workspace and the tagged records of its grid barriers there (a record
 *    counts when its tag equals the launch count + 1, so stale memory must not look like one).
 */
#ifndef MI_ISP_H
#define MI_ISP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types (types.py:12-18 scale factors: u8 255, u16 65535, f16 1, f32 1) */
enum { MI_U8 = 0, MI_U16 = 1, MI_F16 = 2, MI_F32 = 3 };
/* bayer.py:75-79 BayerPattern values */
enum { MI_RGGB = 0, MI_GRBG = 1, MI_GBRG = 2, MI_BGGR = 3 };
/* interpolate.py:9-17 ImageTransform, in declaration order */
enum {
  MI_T_NONE = 0, MI_T_ROTATE_90 = 1, MI_T_ROTATE_180 = 2, MI_T_ROTATE_270 = 3,
  MI_T_TRANSPOSE = 4, MI_T_FLIP_HORIZ = 5, MI_T_FLIP_VERT = 6, MI_T_TRANSVERSE = 7
};
/* camera_isp.py:82-99 loaders */
enum { MI_LOAD_16U = 0, MI_LOAD_32F = 1, MI_LOAD_16F = 2 };

int mi_isp_version(void);
const char* mi_isp_last_error(void);

/* The 4x13x3 integer demosaic weight tables compiled into the kernels (bayer.py:30-55,
 * tap order of bayer.py:15-27).  Host-only; used by the tests to pin the tables. */
int mi_isp_bayer_weights(int32_t out[4 * 13 * 3]);

/* Scratch bytes needed by the calls that take ws_dev, for an H x W frame. */
size_t mi_isp_workspace_bytes(int H, int W);

/* Byte offset inside ws_dev of a uint32 that the whole-frame kernel of mi_isp_pipeline12_reinhard sets to 1
 * when one of its grid barriers timed out (a block of the launch never became resident, e.g. because the
 * caller captured two frames onto parallel branches of one HIP graph).  The frame's output is then invalid.
 * Stays 0 otherwise; hosts that want the check copy these 4 bytes back after synchronising. */
size_t mi_isp_workspace_error_offset(int H, int W);

/* ---- packed.py ------------------------------------------------------------------------ */
/* decode12_kernel (packed.py:92-131): 3 bytes -> two 12-bit values; n_px must be even.
 * scaled: out = cast(f32(v) * f32(scale(out)/4095)).  ids_format: packed.py:37-44 layout. */
int mi_isp_decode12(const uint8_t* enc_dev, void* out_dev, int64_t n_px, int out_dtype,
                    int scaled, int ids_format, void* stream);
/* decode16_kernel (packed.py:135-172): little-endian byte pairs. */
int mi_isp_decode16(const uint8_t* enc_dev, void* out_dev, int64_t n_px, int out_dtype,
                    int scaled, void* stream);
/* encode12_kernel (packed.py:60-89); scaled: v = round_half_away(f32(x) * f32(4095/scale(in))). */
int mi_isp_encode12(const void* values_dev, uint8_t* enc_dev, int64_t n_px, int in_dtype,
                    int scaled, int ids_format, void* stream);

/* ---- camera_isp.py loaders (camera_isp.py:82-99) ---------------------------------------- */
int mi_isp_load_convert(const void* src_dev, void* dst_dev, int64_t n, int mode, int out_dtype,
                        void* stream);

/* ---- bayer.py ------------------------------------------------------------------------- */
/* bayer_to_rgb_kernel (bayer.py:115-190): 13-tap diamond demosaic with border
 * renormalisation; ccm9_host = row-major 3x3 or NULL (host pointer, passed by value). */
int mi_isp_demosaic(const void* cfa_dev, void* rgb_dev, int H, int W, int in_dtype, int out_dtype,
                    int pattern, const float* ccm9_host, void* stream);
/* rgb_to_bayer_kernel (bayer.py:101-112). */
int mi_isp_mosaic(const void* rgb_dev, void* cfa_dev, int H, int W, int dtype, int pattern,
                  void* stream);

/* ---- color/yuv_420.py (the step after the path) ----------------------------------------- */
/* rgb_yuv420_kernel (yuv_420.py:39-66): (H, W, 3) RGB -> (H * 3 / 2, W) planar 4:2:0: rows [0, H) = Y,
 * then two (H/2, W/2) planes, plane 0 = yuv.z, plane 1 = yuv.y.  H, W even.  Reference quirks kept:
 * the matrix sees rgb.bgr, and clamp(0, 1, x) is min(1, x). */
int mi_isp_rgb_to_yuv420(const void* rgb_dev, void* yuv_dev, int H, int W, int in_dtype, int out_dtype,
                         void* stream);
/* yuv420_rgb_kernel (yuv_420.py:68-92): the inverse; H, W are the RGB image's. */
int mi_isp_yuv420_to_rgb(const void* yuv_dev, void* rgb_dev, int H, int W, int in_dtype, int out_dtype,
                         void* stream);

/* ---- interpolate.py -------------------------------------------------------------------- */
/* bilinear_kernel (interpolate.py:19-34,59-86): dst (Hd,Wd,3) <- src (Hs,Ws,3);
 * p = (r/scale0, c/scale1), clamp-to-edge taps, out * scale(out)/scale(in). */
int mi_isp_resize_bilinear(const void* src_dev, void* dst_dev, int Hs, int Ws, int Hd, int Wd,
                           float scale0, float scale1, int in_dtype, int out_dtype, void* stream);
/* transform_kernel (interpolate.py:36-54,93-125); dst is (Ws,Hs,3) for rot90/rot270/transpose. */
int mi_isp_transform(const void* src_dev, void* dst_dev, int Hs, int Ws, int dtype, int transform,
                     void* stream);

/* ---- camera_isp.py: rolling metering + tonemap (stateful ISP semantics) ----------------- */
/* metering_kernel + metering_images (camera_isp.py:142-175): stride-subsampled statistics of
 * n_images (H,W,3) images, blended into state9_dev (f32[9]) with weight alpha.
 * images_host: host array of n_images device pointers.
 * One launch for up to 64 images (a grid barrier inside: at most one block per CU, so it needs no more of the chip than
 * any kernel; a barrier that times out sets the workspace's fault word, mi_isp_workspace_check).  MI_ISP_METERING_LAUNCHES=4
 * in the environment selects the four-launch path (bounds pass, finalize, statistics pass, finalize). */
int mi_isp_metering(const void* const* images_host, int n_images, int H, int W, int stride,
                    int dtype, float* state9_dev, float alpha, void* ws_dev, void* stream);
/* The same with the previous state only READ and the new one only WRITTEN (camera_isp.py:172-173 clones the previous
 * state and lets the kernel update the clone; this form needs no copy).  prev9_dev == state9_dev is mi_isp_metering.
 * After a barrier timeout state9_dev holds the previous state. */
int mi_isp_metering_to(const void* const* images_host, int n_images, int H, int W, int stride, int dtype,
                       const float* prev9_dev, float* state9_dev, float alpha, void* ws_dev, void* stream);
/* The two data passes of the same kernel, split so that a cross-GPU reduction can be
 * inserted between them (one process per GPU, see taichi_image_amd/distributed.py):
 *  bounds: raw (min, max) of the subsample                          -> out2_dev  (f32[2])
 *  sums  : given blended bounds, [log_min, log_max, sum_log, sum_gray, sum_r, sum_g, sum_b, n]
 *                                                                    -> out8_dev  (f32[8]) */
int mi_isp_metering_bounds(const void* const* images_host, int n_images, int H, int W, int stride,
                           int dtype, float* out2_dev, void* ws_dev, void* stream);
int mi_isp_metering_sums(const void* const* images_host, int n_images, int H, int W, int stride,
                         int dtype, const float* bounds2_dev, float* out8_dev, void* ws_dev,
                         void* stream);
/* The sharded batch after an all-gather of the ranks' partials (one all-gather per round, then one of these):
 *  combine_bounds: gathered_dev = n_ranks x [min, max] -> bounds2_out_dev = the blended bounds of camera_isp.py:156-157
 *  combine_sums  : gathered_dev = n_ranks x the 8 floats of mi_isp_metering_sums -> state9_dev updated in place
 *                  (camera_isp.py:131-134,164-166), equal to what mi_isp_metering computes over all ranks' images. */
int mi_isp_metering_combine_bounds(const float* gathered_dev, int n_ranks, const float* state9_dev, float alpha,
                                   float* bounds2_out_dev, void* stream);
int mi_isp_metering_combine_sums(const float* gathered_dev, int n_ranks, const float* bounds2_dev, float* state9_dev,
                                 float alpha, void* stream);
/* reinhard_kernel (camera_isp.py:177-218): pass 1 writes p back into image_dev IN PLACE (as the
 * reference does) and reduces max(p); pass 2 writes u8.  transform != NONE applies
 * interpolate.transform (camera_isp.py:403) while storing; out_dev is then the transformed shape. */
int mi_isp_reinhard(void* image_dev, uint8_t* out_dev, int H, int W, int dtype,
                    const float* state9_dev, float gamma, float intensity, float light_adapt,
                    float color_adapt, int transform, void* ws_dev, void* stream);
/* tonemap_reinhard of a list of images straight to planar YUV 4:2:0 (u8, layout of mi_isp_rgb_to_yuv420): the
 * second Reinhard pass (camera_isp.py:215-218) fused with color/yuv_420.py:39-66; equals
 * mi_isp_rgb_to_yuv420(mi_isp_reinhard_batch(...)) without writing and re-reading the u8 RGB images.
 * Like the reference's pass 1 it overwrites the input images.  H even, W % 16 == 0, no orientation transform. */
int mi_isp_reinhard_batch_yuv420(void* const* images_host, uint8_t* const* yuv_outs_host, int n, int H, int W,
                                 int dtype, const float* state9_dev, float gamma, float intensity,
                                 float light_adapt, float color_adapt, void* ws_dev, void* stream);

/* The per-image loop of ISP.tonemap_reinhard / tonemap_linear (camera_isp.py:399-403,409-413) in
 * one call: n images of the same shape, 4 (Reinhard) or 2 (linear) launches in total instead of per
 * image.  images_host / outs_host: host arrays of device pointers. */
int mi_isp_reinhard_batch(void* const* images_host, uint8_t* const* outs_host, int n, int H, int W,
                          int dtype, const float* state9_dev, float gamma, float intensity,
                          float light_adapt, float color_adapt, int transform, void* ws_dev,
                          void* stream);
/* Extension - NOT the reference's semantics: the u8 outputs of mi_isp_reinhard_batch, bit for bit, without overwriting the
 * images with the mapped values (camera_isp.py:211 does overwrite them).  Pass 1 only reduces, pass 2 recomputes p from the
 * untouched image and rounds it to the image dtype as the write-back would have: a third of the two passes' bytes stays
 * where it is.  For callers that drop the images after the tonemap, or want them as loaded. */
int mi_isp_reinhard_batch_keep(const void* const* images_host, uint8_t* const* outs_host, int n_images, int H, int W,
                               int dtype, const float* state9_dev, float gamma, float intensity, float light_adapt,
                               float color_adapt, int transform, void* ws_dev, void* stream);
int mi_isp_linear_batch(const void* const* images_host, uint8_t* const* outs_host, int n, int H,
                        int W, int dtype, const float* state9_dev, float gamma, int transform,
                        void* ws_dev, void* stream);
/* linear_kernel (camera_isp.py:220-227 -> tonemap.py:12-17). */
int mi_isp_linear(const void* image_dev, uint8_t* out_dev, int H, int W, int dtype,
                  const float* state9_dev, float gamma, int transform, void* ws_dev, void* stream);

/* ---- tonemap.py (stateless, per-image statistics) --------------------------------------- */
/* linear_kernel (tonemap.py:27-46). */
int mi_isp_tonemap_linear(const void* src_dev, void* dst_dev, int H, int W, int in_dtype,
                          int out_dtype, float gamma, void* ws_dev, void* stream);
/* reinhard_kernel (tonemap.py:135-168): bounds -> normalise -> metering -> Reinhard -> bounds ->
 * gamma; the f32 `temp` image of the reference is recomputed per pass, never materialised. */
int mi_isp_tonemap_reinhard(const void* src_dev, void* dst_dev, int H, int W, int in_dtype,
                            int out_dtype, float gamma, float intensity, float light_adapt,
                            float color_adapt, void* ws_dev, void* stream);

/* ---- fused hot path ---------------------------------------------------------------------- */
/* ISP.load_packed12 / load_packed16 (camera_isp.py:333-347,371-373,302-315) in one pass over
 * the packed frame: unpack (bits = 12|16) -> demosaic (+ccm) -> [bilinear resize] -> rgb_dev
 * (Hd,Wd,3) of work_dtype (MI_F16 = Camera16, MI_F32 = Camera32).  scale <= 0: no resize
 * (Hd,Wd must equal H,W).  The intermediate CFA / full-resolution RGB are rounded to
 * work_dtype exactly where the reference stores them. */
int mi_isp_load_packed(const uint8_t* packed_dev, void* rgb_dev, int H, int W, int bits,
                       int ids_format, int pattern, const float* ccm9_host, int work_dtype,
                       int Hd, int Wd, float scale, void* stream);
/* The same, and the image's metering subsample on the way: sub_dev receives rgb[::sub_stride, ::sub_stride] as a dense
 * (ceil(Hd / sub_stride), ceil(Wd / sub_stride), 3) image of work_dtype - what ISP.update_metering reads of every image
 * (camera_isp.py:168-170).  mi_isp_metering(..) on these buffers with stride 1 gives the bits of mi_isp_metering on the
 * images with stride sub_stride (same samples, same order) without the strided gather over the full-size images.  With
 * sub_stride 8 and no resize the subsample is written by the load kernel itself; otherwise by a small gather behind it. */
int mi_isp_load_packed_metered(const uint8_t* packed_dev, void* rgb_dev, int H, int W, int bits,
                               int ids_format, int pattern, const float* ccm9_host, int work_dtype,
                               int Hd, int Wd, float scale, void* sub_dev, int sub_stride, void* stream);
/* n frames of one size (the cameras of a group: one ISP.load_packed12 / 16 each, camera_isp.py:333-347) in ONE launch
 * per 8 frames - same arithmetic and bits as n calls of mi_isp_load_packed[_metered], without n - 1 launches' dispatch,
 * table build and drain.  packed_host / rgb_host / subs_host: host arrays of n device pointers; subs_host may be NULL (no
 * metering subsamples); frames the streaming kernels do not take are loaded one by one. */
int mi_isp_load_packed_batch(const uint8_t* const* packed_host, void* const* rgb_host, void* const* subs_host, int n,
                             int H, int W, int bits, int ids_format, int pattern, const float* ccm9_host, int work_dtype,
                             int Hd, int Wd, float scale, int sub_stride, void* stream);
/* 1 if mi_isp_load_packed_metered (scale <= 0, 16-byte aligned buffers) writes the subsample from inside the load
 * kernel, 0 if it would need the gather behind it (then the caller may as well let mi_isp_metering gather). */
int mi_isp_load_packed_metered_is_fused(int H, int W, int bits, int ids_format, int work_dtype, int sub_stride);
/* 1 if mi_isp_load_packed can fuse a resize by `scale` (its LDS tile holds the source region of a
 * 64x16 destination tile for scale >= ~0.39, any upscale); otherwise demosaic at full size and
 * call mi_isp_resize_bilinear. */
int mi_isp_load_packed_scale_supported(float scale);
/* The stateless chain of test/pipeline.py:26-32 (BASELINE config 2) fused:
 * decode12(scaled, work_dtype) -> bayer_to_rgb -> tonemap_reinhard(dtype=out_dtype) in four data passes.
 * The demosaiced work-dtype image is kept between the passes in out_dev itself when out_dtype ==
 * work_dtype, else in work_image_dev (H * W * 3 work-dtype elements, 16-byte aligned, caller-owned scratch);
 * with work_image_dev == NULL and different dtypes every pass re-derives it from the packed frame
 * (minimal HBM traffic, about 1.5x the time). */
int mi_isp_pipeline12_reinhard(const uint8_t* packed_dev, void* out_dev, void* work_image_dev, int H, int W,
                               int ids_format, int pattern, const float* ccm9_host,
                               int work_dtype, int out_dtype, float gamma, float intensity,
                               float light_adapt, float color_adapt, void* ws_dev, void* stream);
/* One camera group from packed bytes to u8 outputs in one call on the caller's stream (ISP.load_packed12/16 per
 * camera, camera_isp.py:333-347; the rolling metering over the group, :376-385 -> :142-175; ISP.tonemap_reinhard or
 * tonemap_linear, :394-413, with the orientation transform folded into the store).
 *   packed_host / images_host / outs_host: host arrays of n device pointers; images are (Hd, Wd, 3) work-dtype buffers
 *   owned by the caller - after the call they hold what the reference leaves in them (the loaded image for the linear
 *   map, the Reinhard-mapped p for Reinhard, camera_isp.py:211); outs are u8 (Hd, Wd, 3), or the transformed shape.
 *   bits 12 / 16; scale > 0: demosaic + bilinear resize fused (Hd, Wd = ISP.resize_image's size; the scale must
 *   satisfy mi_isp_load_packed_scale_supported), scale <= 0: Hd == H, Wd == W.
 *   state9_dev: the ISP's metering 9-vector (in/out); alpha: 0 for the first group, 1 - moving_alpha afterwards
 *   (camera_isp.py:376-385).  tonemap: 0 = Reinhard (gamma, intensity, light_adapt, color_adapt), 1 = linear (gamma). */
int mi_isp_camera_frame_batch(const uint8_t* const* packed_host, void* const* images_host, uint8_t* const* outs_host,
                              int n, int H, int W, int bits, int ids_format, int pattern, const float* ccm9_host,
                              int work_dtype, int Hd, int Wd, float scale, int metering_stride, float* state9_dev,
                              float alpha, int tonemap, float gamma, float intensity, float light_adapt,
                              float color_adapt, int transform, void* ws_dev, void* stream);

/* One FULL-RESOLUTION camera group from packed bytes to u8 outputs without the image in between - what the reference's
 * bench does per step (taichi_image/bench/camera_isp.py:19-28, Processor.__call__: ISP.load_packed12 per camera,
 * camera_isp.py:333-340; ISP.tonemap_reinhard over the list, :394-403 -> update_metering :376-385 + reinhard_kernel :177-218;
 * the loaded images are dropped).  Three steps on `stream`: the stride-8 subsample of every camera straight from its
 * packed frame (only the rows r % 8 == 0 are demosaiced), the rolling metering over the subsamples (mi_isp_metering), and
 * ONE persistent launch that walks through the cameras: demosaic -> the f16 pixels resident on the chip -> Reinhard and
 * its maximum -> grid barrier (max_out, :213) -> u8 = 255 (p / max_out)^(1 / gamma).  HBM sees the packed frame in and the
 * u8 image out.  Same bits as mi_isp_camera_frame_batch(scale <= 0, tonemap 0, no transform) in outs, state9 and images.
 *   packed_host / outs_host: host arrays of n device pointers (12-bit standard layout; u8 (H, W, 3), 8-byte aligned).
 *   images_host: NULL (the bench's case: p is not stored anywhere), or n device pointers to (H, W, 3) f16 buffers
 *     (16-byte aligned) that receive what the reference leaves in its loaded images: p, camera_isp.py:211.
 *   prev9_dev -> state9_dev, alpha: the metering state before and after this group (mi_isp_metering_to; the two may be the
 *     same buffer), alpha as mi_isp_camera_frame_batch.  Metering stride 8, f16 work dtype (Camera16), 1 <= n <= 64.
 *   scratch_dev: mi_isp_camera_group_scratch_bytes(n, H, W) bytes (the subsamples).
 *   ws_dev: (n + 1) x mi_isp_workspace_bytes(H, W) bytes, zero-filled once (a workspace per camera + the metering's).
 * mi_isp_camera_group_fits: 1 if the frame fits the resident grid (as mi_isp_pipeline12_whole_frame_fits) with this
 *   pattern, work dtype (MI_F16 only) and metering stride (8 only); otherwise use mi_isp_camera_frame_batch.
 * A grid barrier that times out (a foreign kernel holding CUs) sets the camera's workspace fault word and the device's
 * camera-group mailbox word: mi_isp_camera_group_faults(clear) reads it (a host read, no synchronisation); the outputs of
 * that call are invalid.  mi_isp_camera_group_set_poll_limit(polls): poll budget of later launches (0 = default; tests: 1).
 * Launched in the one order of the library's resident-grid kernels (see mi_isp_whole_frame_set_sabotage). */
int mi_isp_camera_group_reinhard(const uint8_t* const* packed_host, void* const* images_host, uint8_t* const* outs_host,
                                 int n, int H, int W, int pattern, const float* ccm9_host, const float* prev9_dev,
                                 float* state9_dev, float alpha, float gamma, float intensity, float light_adapt,
                                 float color_adapt, void* scratch_dev, void* ws_dev, void* stream);
/* The same in its steps, for callers that put something between them (taichi_image_amd: the sharded metering of a
 * multi-GPU group, two all-gathers between the subsample and the tone map):
 *   mi_isp_camera_group_subsample: image[::8, ::8] of every camera's (never materialised) image - (ceil(H / 8), ceil(W / 8), 3)
 *     f16 each, camera i at scratch_dev + i * mi_isp_camera_group_scratch_bytes(1, H, W) - for mi_isp_metering (stride 1);
 *   mi_isp_camera_group_tonemap: the persistent launch, with the Reinhard scalars of state9_dev (read on the device);
 *     ws_dev: n x mi_isp_workspace_bytes(H, W). */
int mi_isp_camera_group_subsample(const uint8_t* const* packed_host, int n, int H, int W, int pattern, const float* ccm9_host,
                                  void* scratch_dev, void* stream);
int mi_isp_camera_group_tonemap(const uint8_t* const* packed_host, void* const* images_host, uint8_t* const* outs_host, int n,
                                int H, int W, int pattern, const float* ccm9_host, const float* state9_dev, float gamma,
                                float intensity, float light_adapt, float color_adapt, void* ws_dev, void* stream);
int mi_isp_camera_group_fits(int H, int W, int pattern, int work_dtype, int metering_stride);
size_t mi_isp_camera_group_scratch_bytes(int n, int H, int W);
int mi_isp_camera_group_faults(int clear);
int mi_isp_camera_group_set_poll_limit(unsigned polls);

/* The same chain (test/pipeline.py:26-32) as ONE persistent launch (csrc/isp_mega.h): the frame is demosaiced once,
 * the f16 RGB image stays in registers and LDS, the three global dependencies of tonemap.py:146-154 are grid
 * barriers inside the kernel; HBM sees the packed frame in and the output out.  f16 work dtype; out_dtype u8 / u16 /
 * f16; frames up to 2 x CUs x 4 waves of 512 x 12 pixels (4096 x 3072 on MI355X) - mi_isp_pipeline12_whole_frame_fits
 * tells.  Results are within the tonemap tolerance of mi_isp_pipeline12_reinhard (same per-pixel functions).
 * The kernel occupies the whole device: launches on different streams of one device are serialised by the library;
 * do not capture two of them onto parallel branches of one HIP graph (the kernel would time out, set the error
 * word of mi_isp_workspace_error_offset and leave an invalid frame - it never hangs). */
int mi_isp_pipeline12_reinhard_whole_frame(const uint8_t* packed_dev, void* out_dev, int H, int W, int ids_format,
                                           int pattern, const float* ccm9_host, int out_dtype, float gamma,
                                           float intensity, float light_adapt, float color_adapt, void* ws_dev,
                                           void* stream);
int mi_isp_pipeline12_whole_frame_fits(int H, int W, int out_dtype);
/* n_frames frames (same size, parameters and pattern) through ONE launch of that kernel per 64 frames: the grid stays
 * resident and walks through the frames, so dispatch, the decode table, drain and launch gap are paid per launch, not
 * per frame (test/pipeline.py:26-32 per frame, as above).  packed_host / out_host: host arrays of n_frames device
 * pointers; ws_dev: n_frames consecutive workspaces (mi_isp_workspace_bytes each, zero-filled once), one per frame. */
int mi_isp_pipeline12_reinhard_whole_frame_batch(const uint8_t* const* packed_host, void* const* out_host, int n_frames,
                                                 int H, int W, int ids_format, int pattern, const float* ccm9_host,
                                                 int out_dtype, float gamma, float intensity, float light_adapt,
                                                 float color_adapt, void* ws_dev, void* stream);
/* What happens when the whole-frame kernel cannot have the chip to itself (a foreign kernel, another process): a wave
 * whose peers do not arrive within the poll budget gives up - the frame is INVALID, nothing hangs - and says so twice:
 *   - the frame's workspace: the 32-bit word at mi_isp_workspace_error_offset() is set (sticky until cleared);
 *   - a host-mapped mailbox word per device, visible to the host WITHOUT synchronising: mi_isp_whole_frame_faults().
 * mi_isp_workspace_check: synchronises `stream`, reports per frame whether its word is set (failed_host[i] = 0 / 1, may
 *   be NULL), clears the set words and returns their number in *n_failed.  The caller re-issues the failed frames
 *   through mi_isp_pipeline12_reinhard (the multi-pass chain needs no co-residency); taichi_image_amd.pipeline does.
 * mi_isp_whole_frame_faults(clear): the mailbox of the current device - non-zero when any whole-frame launch of this
 *   process on this device has timed out since it was last cleared; a plain host read.
 * mi_isp_whole_frame_set_poll_limit(polls): the poll budget of the following launches (0 = the default, ~100 ms);
 *   a diagnostic knob - tests/ use a budget of 1 to provoke the fault path. */
int mi_isp_workspace_check(void* ws_dev, int n_frames, int H, int W, int* failed_host, int* n_failed, void* stream);
int mi_isp_whole_frame_faults(int clear);
int mi_isp_whole_frame_set_poll_limit(unsigned polls);
/* Round 4.  A block one of whose barriers has timed out polls every later barrier of the launch ONCE: it walks through
 * the frames that are left (posting, so that nobody waits for it; marking the fault word of every frame whose records it
 * does not find), so a launch that lost one block ends after about one poll budget (~0.1 - 0.2 s) instead of one budget
 * per remaining barrier; mi_isp_workspace_check also wipes the barrier records of the frames it reports.
 * mi_isp_whole_frame_set_sabotage(block): test hook - that block of every later launch does not post its record at the
 *   first barrier of the launch's first frame (what a block that is not resident looks like to the others); -1 = off.
 * The resident-grid kernels of the library (this one, the one-launch metering, the fused ISP tonemap) are launched in
 * ONE order per device and process: a launch on another stream than the previous one waits for an event recorded behind
 * that one. */
int mi_isp_whole_frame_set_sabotage(int block);
/* The one-launch update_metering (mi_isp_metering) when ITS barrier times out: state9 is left exactly as it was, the
 * workspace's fault word is set and the device's metering mailbox word is stored to.
 * mi_isp_metering_faults(clear): that mailbox word of the current device - a plain host read, no synchronisation.
 * mi_isp_metering_set_poll_limit(polls): poll budget of the following launches (0 = default, ~1 s; tests use 1). */
int mi_isp_metering_faults(int clear);
int mi_isp_metering_set_poll_limit(unsigned polls);
/* Experimental (off by default; MI_ISP_REINHARD_LAUNCHES=1 in the environment enables it): mi_isp_reinhard_batch
 * (reinhard_kernel of camera_isp.py:177-218 for a list of images) as ONE persistent launch when no orientation transform
 * is asked for, the buffers are 16-byte aligned, H * W is a multiple of 512 and an image's mapped values fit the chip's
 * registers (up to 3.1 MP per image pipelined, 6.3 MP one at a time): p is written in place as the reference does
 * (camera_isp.py:211) and kept on chip for the second pass.  Same results as the two launches, bit for bit; measured
 * slower than them, hence off.
 * Its one grid-wide wait (max_out, camera_isp.py:213) can time out like the others: fault word of the workspace +
 * mi_isp_reinhard_faults(clear), the device's mailbox word of this kernel (a plain host read); the outputs of that call
 * are then invalid.  mi_isp_reinhard_set_poll_limit(polls): poll budget of the following launches (0 = default). */
int mi_isp_reinhard_faults(int clear);
int mi_isp_reinhard_set_poll_limit(unsigned polls);

/* The same for n_frames independent frames, frame i on streams_host[i % n_streams]
 * (one frame per stream in flight); ws_dev holds n_frames consecutive workspaces;
 * work_images_host: one scratch image per frame, or NULL. */
int mi_isp_pipeline12_reinhard_batch(const uint8_t* const* packed_host, void* const* out_host,
                                     void* const* work_images_host, int n_frames, int H, int W,
                                     int ids_format, int pattern,
                                     const float* ccm9_host, int work_dtype, int out_dtype,
                                     float gamma, float intensity, float light_adapt,
                                     float color_adapt, void* ws_dev, void* const* streams_host,
                                     int n_streams);


/* A batch of the chain above as a HIP graph (what hipStreamBeginCapture around mi_isp_pipeline12_reinhard_batch gives,
 * done inside the library): create() captures the step for the given buffers - fork to n_streams internal streams,
 * frame i on stream i % n_streams, join - and instantiates it; launch() replays it on `stream` (stream-ordered like any
 * other call; a replay has no launch gaps between the dependent kernels of a stream); destroy() frees it.  The buffers
 * must keep their addresses for the lifetime of the graph; ws_dev holds n_frames workspaces (zero-filled once).
 * whole_frame != 0: the frames through mi_isp_pipeline12_reinhard_whole_frame_batch (one launch, frames one after the other). */
int mi_isp_pipeline12_graph_create(const uint8_t* const* packed_dev, void* const* out_dev, void* const* work_images_dev,
                                   int n_frames, int H, int W, int ids_format, int pattern, const float* ccm9_host,
                                   int work_dtype, int out_dtype, float gamma, float intensity, float light_adapt,
                                   float color_adapt, void* ws_dev, int n_streams, int whole_frame, void** handle);
int mi_isp_pipeline12_graph_launch(void* handle, void* stream);
int mi_isp_pipeline12_graph_destroy(void* handle);

/* ---- measurement aid ----------------------------------------------------------------------- */
/* Launches ONE data pass (0 = demosaic + bounds, 1 = metering sums, 2 = Reinhard bounds, 3 = final
 * map + store) of mi_isp_pipeline12_reinhard, so that bench.py can time each kernel in isolation with
 * events on its own stream.  ws_dev must hold the scalars and per-block partials left by a previous
 * full mi_isp_pipeline12_reinhard call on the same frame (passes 1-3 fold their predecessor's
 * partials in their prologue). */
int mi_isp_pipeline12_pass(const uint8_t* packed_dev, void* out_dev, int H, int W, int ids_format,
                           int pattern, const float* ccm9_host, int work_dtype, int out_dtype,
                           float gamma, float light_adapt, float color_adapt, int pass,
                           void* ws_dev, void* stream);

/* Events around the four data passes of following mi_isp_pipeline12_reinhard[_batch] frames,
 * recorded on the stream each pass runs on.  enable(n, every): time every `every`-th frame, up to n
 * frames (n = 0: off); an event between two launches costs a gap on the stream, so sampling keeps
 * the measured run representative.  collect(): waits for the recorded events; avg_us[k] = average
 * duration of pass k in microseconds, *count = frames timed. */
int mi_isp_profile_enable(int max_frames, int every);
int mi_isp_profile_collect(float avg_us[4], int* count);

#ifdef __cplusplus
}
#endif
#endif /* MI_ISP_H */
