// Whole-frame kernel for the GRBG pattern: parity offsets (PR, PC) = (0, 1).
#define PAT_PR 0
#define PAT_PC 1
#define PAT_FN launch_grbg
#define PAT_OCC blocks_per_cu_grbg
#include "isp_mega_inst.inc"
