// The camera-group kernel (isp_mega_cam.h) for the GRBG pattern: parity offsets (PR, PC) = (0, 1).
#define PAT_PR 0
#define PAT_PC 1
#define PAT_FN launch_cam_grbg
#define PAT_OCC cam_blocks_per_cu_grbg
#include "isp_cam_inst.inc"
