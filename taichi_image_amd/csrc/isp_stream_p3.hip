// Stream kernels for the BGGR pattern: parity offsets (PR, PC) = (1, 1).
#define PAT_PR 1
#define PAT_PC 1
#define PAT_FN launch_bggr
#define PAT_SUB_FN launch_sub_bggr
#include "isp_stream_inst.inc"
