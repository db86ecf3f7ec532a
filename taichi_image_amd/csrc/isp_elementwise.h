// Internal launch interface of isp_elementwise.hip (used by isp_api.hip).
#pragma once
#include "isp_common.h"

namespace ew {

struct PtrList { const void* p[64]; };

}  // namespace ew
#include "isp_finalize.h"
namespace ew {
int finalize(int mode, const FinArgs& a, hipStream_t s);

// ISP reinhard scalars from state9 -> FrameParams (camera_isp.py:186-195)
int isp_reinhard_prep(const float* state9, float* fp, float intensity, float ca, hipStream_t s);

}  // namespace ew
