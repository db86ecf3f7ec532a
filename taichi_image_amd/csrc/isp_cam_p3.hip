// The camera-group kernel (isp_mega_cam.h) for the BGGR pattern: parity offsets (PR, PC) = (1, 1).
#define PAT_PR 1
#define PAT_PC 1
#define PAT_FN launch_cam_bggr
#define PAT_OCC cam_blocks_per_cu_bggr
#include "isp_cam_inst.inc"
