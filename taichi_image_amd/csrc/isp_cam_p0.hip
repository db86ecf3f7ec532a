// The camera-group kernel (isp_mega_cam.h) for the RGGB pattern: parity offsets (PR, PC) = (0, 0).
#define PAT_PR 0
#define PAT_PC 0
#define PAT_FN launch_cam_rggb
#define PAT_OCC cam_blocks_per_cu_rggb
#include "isp_cam_inst.inc"
