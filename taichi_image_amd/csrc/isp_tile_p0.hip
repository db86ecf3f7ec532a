// Tile kernels for the RGGB pattern (BayerPattern value 0): parity offsets (PR, PC) = (0, 0).
#define PAT_PR 0
#define PAT_PC 0
#define PAT_FN launch_rggb
#include "isp_tile_inst.inc"
