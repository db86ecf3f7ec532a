// Internal launch interface of isp_elementwise.hip (used by isp_api.hip).
#pragma once
#include "isp_common.h"
#include <mutex>

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

// ---- kernels whose blocks meet at a grid barrier (mega::frame_kernel, metering_fused_kernel, isp fused tonemap) ----
// Each of them needs ALL its blocks resident; two such grids launched from two streams can each get a part of the chip
// and wait for the rest of it - until the poll budgets run out: fault words, lost frames.  So every launch of such a
// kernel by this process is put in ONE order per device, whatever its kind: under `mu` the launching stream first waits
// for the event recorded behind the previous resident-grid launch (when that was on another stream), launches, and
// records the event again.  The lock is held over all three steps.  (Round 3 kept one order for the whole-frame kernel
// and another for the metering kernel: a metering grid and a whole-frame grid on two streams could time each other out.)
// The host-mapped mailbox page of a device (16 words) is where these kernels report a timeout without a synchronisation:
// word 0 the whole-frame kernel, word 1 the metering kernel, word 2 the fused ISP tonemap, word 3 the camera-group kernel.
struct ResidentOrder {
  std::mutex mu;
  hipEvent_t done[16] = {};
  hipStream_t last[16] = {};
  bool has_last[16] = {};
  unsigned* mailbox_host[16] = {};
  unsigned* mailbox_dev[16] = {};
};
enum { MAILBOX_WHOLE_FRAME = 0, MAILBOX_METERING = 1, MAILBOX_ISP_TONEMAP = 2, MAILBOX_CAMERA_GROUP = 3 };
ResidentOrder& resident_order();
// callers hold resident_order().mu; dev in [0, 16)
int resident_mailbox_locked(int dev);                       // allocates the device's mailbox page on first use
int resident_enter_locked(int dev, hipStream_t s);          // mailbox + wait for the previous resident-grid launch
int resident_leave_locked(int dev, hipStream_t s);          // record the event behind the launch just made

// ISP reinhard scalars from state9 -> FrameParams (camera_isp.py:186-195)
int isp_reinhard_prep(const float* state9, float* fp, float intensity, float ca, hipStream_t s);

}  // namespace ew
