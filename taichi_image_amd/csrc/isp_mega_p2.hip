// Whole-frame kernel for the GBRG pattern: parity offsets (PR, PC) = (1, 0).
#define PAT_PR 1
#define PAT_PC 0
#define PAT_FN launch_gbrg
#define PAT_OCC blocks_per_cu_gbrg
#include "isp_mega_inst.inc"
