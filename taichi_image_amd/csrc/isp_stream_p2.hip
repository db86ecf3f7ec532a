// Stream kernels for the GBRG pattern: parity offsets (PR, PC) = (1, 0).
#define PAT_PR 1
#define PAT_PC 0
#define PAT_FN launch_gbrg
#define PAT_SUB_FN launch_sub_gbrg
#include "isp_stream_inst.inc"
