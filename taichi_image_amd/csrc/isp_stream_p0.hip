// Stream kernels for the RGGB pattern: parity offsets (PR, PC) = (0, 0).
#define PAT_PR 0
#define PAT_PC 0
#define PAT_FN launch_rggb
#define PAT_SUB_FN launch_sub_rggb
#include "isp_stream_inst.inc"
