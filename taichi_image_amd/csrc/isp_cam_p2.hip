// The camera-group kernel (isp_mega_cam.h) for the GBRG pattern: parity offsets (PR, PC) = (1, 0).
#define PAT_PR 1
#define PAT_PC 0
#define PAT_FN launch_cam_gbrg
#define PAT_OCC cam_blocks_per_cu_gbrg
#include "isp_cam_inst.inc"
