// Whole-frame kernel for the RGGB pattern: parity offsets (PR, PC) = (0, 0).
#define PAT_PR 0
#define PAT_PC 0
#define PAT_FN launch_rggb
#define PAT_OCC blocks_per_cu_rggb
#include "isp_mega_inst.inc"
