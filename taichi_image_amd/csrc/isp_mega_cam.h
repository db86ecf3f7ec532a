// A camera group's frames from packed bytes to u8 outputs in ONE persistent launch (round 4): what the reference's own
// bench does per step (bench/camera_isp.py:19-28, Processor.__call__: `load_packed12` per camera, `tonemap_reinhard`
// over the list, the loaded images dropped) without the image ever leaving the chip between the two.
//
// The ISP's Reinhard (camera_isp.py:177-218) needs no statistics of the image it maps - its scalars come from the rolling
// metering (state9), which the caller updates BEFORE this launch from the stride-8 subsample (strm::sub_kernel: straight
// from the packed frames) - and one global value per image: max_out, the largest mapped value (camera_isp.py:213).  So
// the whole-frame design of isp_mega.h applies with ONE grid barrier per frame instead of two or three:
//
//   phase A  streaming demosaic of the wave's 12 rows x 512 columns (isp_stream.h machinery) -> the f16 pixels the
//            reference's load_packed12 materialises, resident: 5 rows in LDS, 7 in VGPRs
//   phase C  p = reinhard(pixel) from the resident rows (camera_isp.py:198-210), its maximum, and p ROUNDED TO THE IMAGE
//            DTYPE - the bytes the reference writes back over its image (:211) - into the same resident slots
//   barrier  -> max_out (tagged records, isp_mega.h: block_reduce_post / barrier_fold)
//   phase D  u8 = 255 (p / max_out)^(1 / gamma) (:216-218), wave-contiguous streamed stores; p itself is stored only
//            when the caller wants the reference's side effect (the image overwritten with p)
//
// HBM sees the packed frame in (18.9 MB at 4K) and the u8 image out (37.7 MB), plus 75.5 MB when p is kept - against
// 358 MB of the load -> metering -> pass 1 -> pass 2 sequence.  The arithmetic is that of strm::stream_kernel<S_STORE> and
// rgb_pass_kernel<PM_ISP_RH_P1 / P2> instruction for instruction (same helpers): tests/ compare the two paths bit for bit.
//
// The skeleton of the frame loop (geometry, first loads, phase A) is frame_kernel's, REPEATED here rather than shared: that
// body sits at exactly 256 VGPRs without a spill, and factoring its lambdas out moves its register allocation (every edit
// of isp_mega.h in rounds 3 and 4 was checked against the kernel's assembly).  The barrier, the posts, the reductions and the
// row helpers are shared (isp_mega.h, isp_stream.h).
#pragma once
#include "isp_mega.h"

#pragma clang fp contract(off)

namespace mega {

struct CamIO {
  const void* src;                   // packed frame
  void* p_out;                       // (H, W, 3) f16: receives p (what the reference leaves in the loaded image), or NULL
  void* out;                         // (H, W, 3) u8
  float* ws;                         // the frame's own workspace (epoch / error words, barrier records)
};
struct CBatch {
  MArgs m;                           // m.s.t: geometry, colour matrix, light / colour adaptation; m.s.intensity
  const float* state9;               // the ISP's metering vector (device), already updated for this group
  float gamma_inv;
  int n_frames;
  CamIO io[MAX_BATCH];
};

// 12 floats -> 12 u8, packed into three dwords: cast_out<uint8_t> (clamp to [0, 255], truncate, NaN -> 0) in ONE
// instruction per value.  v_cvt_pk_u8_f32 converts, saturates and drops the byte into its place in a dword; it rounds by
// the MODE register's single-precision rounding mode, so the conversions sit between two s_setreg (round toward zero, back
// to nearest-even) inside one asm block that nothing else can be scheduled into.  scratch/cvt_pk_u8_test.hip: equal to
// clamp + truncation for fractions, ties, out-of-range values, infinities and NaN.  Against v_med3 + v_cvt_u32 + the
// byte packing (v_perm / v_or3) it saves ~2 instructions per value: phase D is issue-bound when gamma != 1.
MI_DEV void pack12_u8_rtz(const float* v, uint32_t (&d)[3]) {
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "s_nop 1\n\t"
      "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\t"
      "v_cvt_pk_u8_f32 %1, %7, 0, 0\n\t"
      "v_cvt_pk_u8_f32 %2, %11, 0, 0\n\t"
      "v_cvt_pk_u8_f32 %0, %4, 1, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %8, 1, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %12, 1, %2\n\t"
      "v_cvt_pk_u8_f32 %0, %5, 2, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %9, 2, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %13, 2, %2\n\t"
      "v_cvt_pk_u8_f32 %0, %6, 3, %0\n\t"
      "v_cvt_pk_u8_f32 %1, %10, 3, %1\n\t"
      "v_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"
      "s_nop 1"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]),
        "v"(v[11]));
}

template <int PR, int PC>
__global__ __launch_bounds__(THREADS, 2) void camera_kernel(const CBatch cb) {
  typedef half_t E;
  const MArgs& m = cb.m;
  const SArgs& a = m.s;
  const Params& p = a.t;
  __shared__ __attribute__((aligned(16))) uint4 xl[WAVES][NL][ROW_U4];   // resident rows 0..NL-1 of each wave; output staging
  __shared__ float lut[4096];
  __shared__ float red[WAVES][16];
  __shared__ float sh_fp[FP_COUNT];
  __shared__ unsigned arrived;
  __shared__ FoldLds fl;
  if (threadIdx.x < 4) { fl.ticket[threadIdx.x] = 0; fl.done[threadIdx.x] = 0; fl.flag[threadIdx.x] = 0; }
  if (threadIdx.x == 0) { arrived = 0; fl.faulted = 0; }
  if (threadIdx.x < FP_COUNT) sh_fp[threadIdx.x] = 0.f;
  // (wave 0 executes the two statements in order: thread 0's scalars land on the zero-filled table)
  if (threadIdx.x == 0) isp_reinhard_scalars(cb.state9, sh_fp, a.intensity, p.ca);      // camera_isp.py:186-195

  struct Geo {
    int lane, wave, g, bx, c0, r_begin, r_end, active_lanes;
    bool wave_ok, col_ok, is_left, is_right, any_left, any_right;
    uint32_t col_off, ext_off;
  };
  const uint32_t pitch = (uint32_t)p.W * 3 / 2;
  auto geo = [&](int tid, int bid) {
    Geo G;
    G.lane = tid & 63;
    G.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    G.g = bid * WAVES + G.wave;
    G.wave_ok = G.g < a.n_waves;
    const int by = G.g / a.bands_x;
    G.bx = G.g - by * a.bands_x;
    G.c0 = G.bx * BAND + G.lane * 8;
    G.r_begin = by * ROWS;
    G.r_end = G.wave_ok ? (G.r_begin + ROWS < p.H ? G.r_begin + ROWS : p.H) : G.r_begin;
    G.col_ok = G.wave_ok && G.c0 < p.W;
    G.active_lanes = !G.wave_ok ? 0 : (p.W - G.bx * BAND >= BAND ? 64 : (p.W - G.bx * BAND) / 8);
    G.col_off = G.col_ok ? (uint32_t)G.c0 * 3 / 2 : INVALID_OFF;
    const bool ext_ok = G.col_ok && ((G.lane == 0 && G.c0 > 0) || (G.lane == 63 && G.c0 + 8 < p.W));
    G.ext_off = ext_ok ? (uint32_t)G.c0 * 3 / 2 + (G.lane == 0 ? -4 : 12) : INVALID_OFF;
    G.is_left = G.col_ok && G.c0 == 0; G.is_right = G.col_ok && G.c0 + 8 == p.W;
    G.any_left = __builtin_amdgcn_ballot_w64(G.is_left) != 0; G.any_right = __builtin_amdgcn_ballot_w64(G.is_right) != 0;
    return G;
  };
  auto load_row = [&](const Geo& G, const __amdgpu_buffer_rsrc_t rsrc, int r, uint32_t (&d)[4]) {
    const uint32_t row_off = (r >= 0 && r < p.H && r < G.r_end + 2) ? (uint32_t)r * pitch : INVALID_OFF;     // scalar
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    const u3 q = __builtin_amdgcn_raw_buffer_load_b96(rsrc, G.col_off + row_off, 0, 0);
    d[0] = q.x; d[1] = q.y; d[2] = q.z;
    d[3] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, G.ext_off + row_off, 0, 0);
  };
  auto src_rsrc = [&](const void* src) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, (int)((uint32_t)p.H * pitch), 0x00020000);
  };
  uint32_t pro[4][4];
  uint32_t raw[2][2][4];
  auto first_loads = [&](const Geo& G, const void* src) {
    const __amdgpu_buffer_rsrc_t rs = src_rsrc(src);
#pragma unroll
    for (int q = 0; q < 4; ++q) load_row(G, rs, G.r_begin - 2 + q, pro[q]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      load_row(G, rs, G.r_begin + 2 + 2 * j, raw[j][0]);
      load_row(G, rs, G.r_begin + 3 + 2 * j, raw[j][1]);
    }
  };
  first_loads(geo(threadIdx.x, blockIdx.x), cb.io[0].src);

  for (int e = threadIdx.x; e < 4096; e += THREADS) lut[e] = tile::decode_scaled<E>((uint32_t)e, p.k_decode);
  __syncthreads();                                    // table, scalars, tickets, flags, `arrived`: the only workgroup barrier

  const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  auto thread_id = [&]() {                              // (as frame_kernel: nothing but the frame counter lives around the loop)
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return wave_s * 64 + l;
  };
  for (int f = 0; f < cb.n_frames; ++f) {
  int tid_ = thread_id(), bid_ = blockIdx.x;
  asm volatile("" : "+s"(bid_));
  const Geo G = geo(tid_, bid_);
  const int lane = G.lane, wave = G.wave, g = G.g, bx = G.bx, r_begin = G.r_begin, r_end = G.r_end, active_lanes = G.active_lanes;
  const bool younger = bid_ >= (a.n_blocks >> 1);
  const bool wave_ok = G.wave_ok, col_ok = G.col_ok, is_left = G.is_left, is_right = G.is_right, any_left = G.any_left,
             any_right = G.any_right;
  (void)g; (void)wave_ok;
  const CamIO io = cb.io[f];
  const unsigned seq = (unsigned)f + 1u;
  float* const ws = io.ws;
  float* const partials = ws + FP_COUNT;
  const __amdgpu_buffer_rsrc_t rsrc = src_rsrc(io.src);
  const uint32_t epoch = __builtin_amdgcn_readfirstlane(reinterpret_cast<const unsigned*>(ws)[FP_EPOCH]);
  const uint32_t tag_ = m.launch_id * 0x9E3779B1u + epoch + 1u;       // (frame_kernel: the workspace's and the host's launch counts)
  const uint32_t tag = tag_ == 0u ? 1u : tag_;

  // ================================ phase A: demosaic once ================================
  float wq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wq[i] = vgpr(wq_value(i));
  WinRow win[6];
  uint32_t xr[NR][12];                                // resident rows NL..ROWS-1 (packed f16 pairs)
#pragma unroll
  for (int q = 0; q < 4; ++q) decode_row(pro[q], lut, lane, win[q]);

  static_for<0, ROWS / 2>([&](auto ibc) {
    constexpr int IB = decltype(ibc)::value, PH = IB % 3;
    const int r = r_begin + 2 * IB;
    decode_row(raw[IB % 2][0], lut, lane, win[(2 * PH + 4) % 6]);
    decode_row(raw[IB % 2][1], lut, lane, win[(2 * PH + 5) % 6]);
    if constexpr (IB + 2 < ROWS / 2) {
      load_row(G, rsrc, r + 6, raw[IB % 2][0]);
      load_row(G, rsrc, r + 7, raw[IB % 2][1]);
    }
    if constexpr (2 * IB >= NL) fresh(xr[2 * IB - NL]);
    if constexpr (2 * IB + 1 >= NL) fresh(xr[2 * IB + 1 - NL]);
    if (r < r_end) {                          // wave-uniform
      WinRow w6[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) w6[k] = win[(2 * PH + k) % 6];
      static_for<0, 2>([&](auto ic) {
        constexpr int I = decltype(ic)::value, RR = 2 * IB + I;
        const int row = r + I;
        prio_turn(RR, younger);
        float v[24];
        accumulate_row<PR, PC, I, true>(w6, wq, v);
        if (row < 2 || row >= p.H - 2) border_fix_rows<PR, PC, I>(v, tile::inside_mask(row, p.H), is_left, is_right);
        else if (any_left || any_right) border_fix_cols<PR, PC, I>(v, is_left, is_right, any_left, any_right);
        if (p.has_ccm) {                              // bayer.py:152-153, sequential fp32 dot
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float x = v[3 * k], y = v[3 * k + 1], z = v[3 * k + 2];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
              v[3 * k + ch] = (p.ccm[3 * ch] * x + p.ccm[3 * ch + 1] * y) + p.ccm[3 * ch + 2] * z;
          }
        }
        // the pixel as load_packed12 materialises it: clamped (bayer.py:155), rounded to f16
        uint32_t pk[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) pk[j] = tile::cvt_pk_f16_clamp01(v[2 * j], v[2 * j + 1]);
        if constexpr (RR < NL) {
          uint4 mine[3];
          __builtin_memcpy(mine, pk, sizeof(mine));
#pragma unroll
          for (int j = 0; j < 3; ++j) xl[wave][RR][lane * 3 + j] = mine[j];
        } else {
#pragma unroll
          for (int j = 0; j < 12; ++j) xr[RR - NL][j] = pk[j];
        }
      });
    }
  });

  auto resident_pk = [&](auto rrc, uint32_t (&pk)[12]) {
    constexpr int RR = decltype(rrc)::value;
    if constexpr (RR < NL) {
      uint4 mine[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) mine[j] = xl[wave][RR][lane * 3 + j];
      __builtin_memcpy(pk, mine, sizeof(mine));
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) pk[j] = xr[RR - NL][j];
    }
  };

  // ================================ phase C: p and its maximum (camera_isp.py:198-213) ================================
  // The operands of the phase are set up at its start, every frame (frame_kernel: no registers to park them in).
  ReinhardK rk;
  const bool ca0 = p.ca == 0.f;
  const float inv = vgpr(sh_fp[FP_INV]);
  rk.la = vgpr(p.la); rk.ca = vgpr(p.ca);
  rk.map_key = vgpr(sh_fp[FP_MAPKEY]); rk.ei = vgpr(sh_fp[FP_EI]);
  rk.mean3[0] = vgpr(sh_fp[FP_MEAN3]); rk.mean3[1] = vgpr(sh_fp[FP_MEAN3 + 1]); rk.mean3[2] = vgpr(sh_fp[FP_MEAN3 + 2]);
  float vmax = -__builtin_inff();
  // isp_norm (camera_isp.py:200: no clamp) of element e of a packed row: the difference straight from the half
  // (v_fma_mix_f32: h * 1 + (-lo), the half widened exactly, one rounding - the bits of converting first and subtracting),
  // then the product, rounded where it is made (isp_math.h: isp_norm)
  const float nlo = vgpr(-sh_fp[FP_LO]);
  auto norm_pk = [&](auto ec, const uint32_t (&pk)[12]) __attribute__((always_inline)) {
    constexpr int e = decltype(ec)::value;
    float d;
    if constexpr ((e & 1) == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pk[e / 2]), "v"(nlo));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pk[e / 2]), "v"(nlo));
    return f32_rounded(d * inv);
  };
  auto tone_row = [&](auto ca0_c, const uint32_t (&pk)[12], float (&q)[24]) {
    constexpr bool CA0 = decltype(ca0_c)::value;
    static_for<0, 8>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      float x[3], o[3];
      x[0] = norm_pk(std::integral_constant<int, 3 * k>{}, pk);
      x[1] = norm_pk(std::integral_constant<int, 3 * k + 1>{}, pk);
      x[2] = norm_pk(std::integral_constant<int, 3 * k + 2>{}, pk);
      reinhard_px<CA0>(x, rk, o);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) q[3 * k + ch] = o[ch];
      vmax = fmaxf(vmax, fmaxf(o[0], fmaxf(o[1], o[2])));
    });
  };
  static_for<0, ROWS>([&](auto rrc) {
    constexpr int RR = decltype(rrc)::value;
    prio_turn(RR, younger);
    if (r_begin + RR < r_end) {
      uint32_t pk[12];
      resident_pk(rrc, pk);
      float q[24];
      if (ca0) tone_row(std::true_type{}, pk, q);
      else tone_row(std::false_type{}, pk, q);
      // p as the reference stores it over its image (camera_isp.py:211: ti.cast(p, f16)), in the pixel's place
#pragma unroll
      for (int j = 0; j < 12; ++j) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[j]) : "v"(f32_rounded(q[2 * j])), "v"(f32_rounded(q[2 * j + 1])));
      if constexpr (RR < NL) {
        uint4 mine[3];
        __builtin_memcpy(mine, pk, sizeof(mine));
#pragma unroll
        for (int j = 0; j < 3; ++j) xl[wave][RR][lane * 3 + j] = mine[j];
      } else {
#pragma unroll
        for (int j = 0; j < 12; ++j) xr[RR - NL][j] = pk[j];
      }
    }
  });
  float* rows_max = partials + (size_t)MROW_BAR2 * p.part_stride;
  {
    if (!col_ok) vmax = -__builtin_inff();                // lanes beyond the image mapped zeros
    const float v2[2] = {vmax, vmax};
    const int op[2] = {0, 1};
    block_reduce_post<2>(v2, op, red, &arrived, rows_max, p.part_stride, blockIdx.x, wave, lane, tag, 2,
                         f == 0 && m.sabotage_block == (int)blockIdx.x);
  }
  // ================================ barrier: max_out (camera_isp.py:190,213) ================================
  barrier_fold<2, ew::FIN_MAXOUT>(m, ws, seq, 2, rows_max, tag, sh_fp, fl, lane);
  const float maxout_inv = vgpr(1.0f / sh_fp[FP_MAXOUT]);
  const float gamma_inv = cb.gamma_inv;

  // ================================ phase D: the u8 image (camera_isp.py:215-218) ================================
  uint32_t off8[6], off16[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    off8[j] = (j * 64 + lane) < active_lanes * 3 ? (uint32_t)(j * 64 + lane) * 8u : INVALID_OFF;       // 24 u8 = 3 x 8 bytes
    off16[j] = (j * 64 + lane) < active_lanes * 3 ? (uint32_t)(j * 64 + lane) * 16u : INVALID_OFF;     // 24 f16 = 3 x 16 bytes
  }
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(io.out, 0, (int)((uint32_t)p.H * (uint32_t)p.W * 3u), 0x00020000);
  const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
      io.p_out, 0, io.p_out ? (int)((uint32_t)p.H * (uint32_t)p.W * 6u) : 0, 0x00020000);
  const bool keep_p = io.p_out != nullptr;              // (uniform)
  auto finish_row = [&](auto rrc) {
    constexpr int RR = decltype(rrc)::value;
    uint32_t pk[12];
    resident_pk(rrc, pk);
    float o[24];
    // p / max_out straight from the packed halves (v_fma_mix_f32 with a zero addend: the half is widened exactly, the product
    // rounded once - the bits of converting first and multiplying, one instruction instead of two)
    static_for<0, 24>([&](auto ec) { constexpr int e = decltype(ec)::value; o[e] = mul_mix_h<e>(pk[e / 2], maxout_inv); });
    if (gamma_inv != 1.f) {
      asm volatile("" ::: "memory");                    // (a real branch: see linear_n)
#pragma unroll
      for (int j = 0; j < 24; ++j) o[j] = hw_pow(o[j], gamma_inv);
    }
#pragma unroll
    for (int j = 0; j < 24; ++j) o[j] *= 255.f;
    // staging: the LDS slot of a row that has been consumed (its own, or row 0's for the register rows)
    uint4* stage = xl[wave][RR < NL ? RR : 0];
    const uint32_t row = (uint32_t)(r_begin + RR);
    if (keep_p) {
      uint4 mine[3];
      __builtin_memcpy(mine, pk, sizeof(mine));
      wave_store_units<uint4, 3, ST_KEEP>(prsrc, row * (uint32_t)p.W * 6u + (uint32_t)bx * BAND * 6u, off16, lane, stage, mine);
    }
    uint32_t d0[3], d1[3];
    pack12_u8_rtz(o, d0);
    pack12_u8_rtz(o + 12, d1);
    const uint2 mine8[3] = {make_uint2(d0[0], d0[1]), make_uint2(d0[2], d1[0]), make_uint2(d1[1], d1[2])};
    wave_store_units<uint2, 3, ST_STREAM>(orsrc, row * (uint32_t)p.W * 3u + (uint32_t)bx * BAND * 3u, off8, lane, stage, mine8);
  };
  static_for<0, ROWS>([&](auto rrc) {
    constexpr int RR = decltype(rrc)::value;
    if (r_begin + RR < r_end) finish_row(rrc);
    // the next frame's first rows are asked for while the last rows of this one leave (unconditional: frame_kernel)
    if constexpr (RR == NL) first_loads(G, cb.io[f + 1 < cb.n_frames ? f + 1 : f].src);
  });

  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned*>(ws)[FP_EPOCH] = epoch + 1u;
  }   // frames
}

int launch_cam_rggb(const CBatch& cb, hipStream_t stream);
int launch_cam_grbg(const CBatch& cb, hipStream_t stream);
int launch_cam_gbrg(const CBatch& cb, hipStream_t stream);
int launch_cam_bggr(const CBatch& cb, hipStream_t stream);
static inline int launch_cam(const CBatch& cb, int pattern, hipStream_t stream) {
  switch (pattern) {
    case MI_RGGB: return launch_cam_rggb(cb, stream);
    case MI_GRBG: return launch_cam_grbg(cb, stream);
    case MI_GBRG: return launch_cam_gbrg(cb, stream);
    default: return launch_cam_bggr(cb, stream);
  }
}
int cam_blocks_per_cu_rggb();
int cam_blocks_per_cu_grbg();
int cam_blocks_per_cu_gbrg();
int cam_blocks_per_cu_bggr();
static inline int cam_blocks_per_cu(int pattern) {
  switch (pattern) {
    case MI_RGGB: return cam_blocks_per_cu_rggb();
    case MI_GRBG: return cam_blocks_per_cu_grbg();
    case MI_GBRG: return cam_blocks_per_cu_gbrg();
    default: return cam_blocks_per_cu_bggr();
  }
}

}  // namespace mega
