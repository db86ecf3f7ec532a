// Tile kernels for the BGGR pattern (BayerPattern value 3): parity offsets (PR, PC) = (1, 1).
#define PAT_PR 1
#define PAT_PC 1
#define PAT_FN launch_bggr
#include "isp_tile_inst.inc"
