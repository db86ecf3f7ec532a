// Whole-frame kernel for the BGGR pattern: parity offsets (PR, PC) = (1, 1).
#define PAT_PR 1
#define PAT_PC 1
#define PAT_FN launch_bggr
#define PAT_OCC blocks_per_cu_bggr
#include "isp_mega_inst.inc"
