// Internal launch interface of isp_elementwise.hip (used by isp_api.hip).
#pragma once
#include "isp_common.h"

namespace ew {

struct PtrList { const void* p[64]; };

}  // namespace ew
#include "isp_finalize.h"
namespace ew {
int finalize(int mode, const FinArgs& a, hipStream_t s);

// The three elementwise passes of tonemap.py:147-154 on an RGB image whose per-block bounds partials
// `bounds` are in the workspace; the finalize steps between the passes are pulled into the prologue of
// each consuming pass (no finalize launches).  dst may be src.
// which: -1 = all three, 1..3 = only that data pass (measurement aid).
struct PullSrc { const float* partials; int stride, n, bounds_post; };
int tail_blocks(int H, int W);       // blocks (= partials) the elementwise passes of an H x W image use
int tonemap_reinhard_tail(const void* src, void* dst, int H, int W, int in_dtype, int out_dtype, float gamma,
                          float intensity, float la, float ca, float* ws, int which, const PullSrc& bounds,
                          hipStream_t s);

// image[::stride, ::stride] -> dense (ceil(H / stride), ceil(W / stride), 3) image of the same dtype
int subsample(const void* img, void* sub, int H, int W, int stride, int dtype, hipStream_t s);

// ISP reinhard scalars from state9 -> FrameParams (camera_isp.py:186-195)
int isp_reinhard_prep(const float* state9, float* fp, float intensity, float ca, hipStream_t s);

}  // namespace ew
