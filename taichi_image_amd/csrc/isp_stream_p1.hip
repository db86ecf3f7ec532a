// Stream kernels for the GRBG pattern: parity offsets (PR, PC) = (0, 1).
#define PAT_PR 0
#define PAT_PC 1
#define PAT_FN launch_grbg
#define PAT_SUB_FN launch_sub_grbg
#include "isp_stream_inst.inc"
