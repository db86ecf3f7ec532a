// The fused config-2 chain (test/pipeline.py:26-32: decode12 -> bayer_to_rgb -> tonemap_reinhard) as ONE
// persistent launch: the frame is demosaiced once, the f16 RGB image never leaves the chip, and the three
// global dependencies of tonemap.py:146-154 (bounds -> statistics -> bounds of the mapped image -> final
// map) are grid barriers inside the kernel instead of kernel boundaries.
//
// Why: every pass of this chain is bound by instruction throughput, not bandwidth, a kernel
// boundary costs ~5 us of drain + launch + refill, and re-deriving the image per pass repeats the demosaic
// (the largest single cost) three times.  Here HBM sees exactly the algorithmic bytes - the packed frame in,
// the output out - and the demosaic runs once.
//
// Residency: a 4096 x 3072 frame is 75.5 MB of f16 RGB; the chip's register files hold 128 MB and its LDS
// 40 MB.  Every wave (2048 of them, two per SIMD, all resident) keeps its 12 rows x 512 columns as packed
// f16 pairs: 5 rows in LDS (3 KB per row and wave), 7 rows in 84 VGPRs.
//
//   phase A  streaming demosaic of the wave's rows (isp_stream.h machinery) -> bounds of the image, the
//            statistics of tonemap.py:147-149 under the assumption that the bounds are exactly (0, 1), and
//            the rounded f16 pixels into their resident slots
//   barrier  -> bounds; when they are not (0, 1): phase B recomputes the statistics from the resident
//            image (no memory traffic) and a second barrier follows
//   phase C  Reinhard on the resident image -> bounds of the mapped image
//   barrier
//   phase D  Reinhard again, final normalisation, cast, wave-contiguous stores
//
// Grid barrier = tagged partial rows, no counters: the last wave of a block stores the block's partial results as
// 16-byte chunks {3 values, tag} write-through (sc1) and goes on; tag = the launch count of the workspace + 1.  One
// wave per block polls the chunks of ALL blocks with sc1 loads (one aligned 16-byte load returns values and tag
// together) until every tag matches, folds them in block order - identical arithmetic in every block - and hands
// the scalars to the other waves of its block through LDS.  Against the counter design (store, wait for the
// write-through, atomic add; poll the counters, then load the partials) two of the four dependent memory round trips
// after the last arrival are gone.  Every block must be resident at once: the host launches at most 2 blocks per CU
// and refuses larger frames (they take the multi-pass chain); a bounded poll turns a missing peer into an error flag
// in the workspace (and a store to the device's host-mapped mailbox) instead of a hang.
//
// Rounds 2 and 3 rested on an ASSUMPTION: a 16-byte sc1 store to a 16-byte aligned address is observed WHOLE by a 16-byte
// sc1 load of that address - a poller never sees the new tag next to old values.  The ISA manual does not promise it (the
// evidence was empirical: 2 000 000 frames compared bit for bit).  Round 4: the records check THEMSELVES.  The fourth word
// of a chunk is tag ^ rec_hash(x, y, z); a poller accepts a chunk only when w ^ rec_hash(x, y, z) == tag, so a chunk made
// of old and new words is - like a chunk that has not been posted yet - simply asked for again.  A torn read is now a
// detected condition (one more round trip) instead of a silently wrong frame.
//
// Round 3: ONE launch walks through a batch of frames (MBatch): the grid stays resident, the counters in LDS run on over
// the frames, a wave that has stored its rows of frame f goes straight on to frame f + 1.
#pragma once
#include "isp_stream.h"

#pragma clang fp contract(off)

namespace mega {

using namespace strm;

// Reading aid (scripts/isa_census.py compiles isp_mega_p0.hip with -DMI_MEGA_CENSUS to assembly): the run-time choices
// of the headline configuration become constants (interior band, no colour matrix, bounds (0, 1), color_adapt == 0, f16
// output) so that each phase is one straight-line stretch between two "; MI_MARK n" comments.  Never part of the library.
#if defined(MI_MEGA_CENSUS)
#define MI_CENSUS(expr, val) (val)
#define MI_MSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MI_MARK " #i); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MI_CMARK(i) MI_MSTAMP(i)
#else
#define MI_CENSUS(expr, val) (expr)
#define MI_CMARK(i) do {} while (0)
#if defined(MI_STREAM_STAMPS)           /* the stamps of a frame go behind the partial rows of ITS workspace */
#define MI_MSTAMP(i)                                                                                          \
  do {                                                                                                        \
    if (lane == 0 && wave_ok)                                                                                 \
      reinterpret_cast<unsigned*>(partials + (size_t)PART_ROWS * p.part_stride)[g * 16 + (i)] = MI_STAMP_NOW();  \
  } while (0)
#else
#define MI_MSTAMP(i) do {} while (0)
#endif
#endif

constexpr int ROWS = 12;             // rows per wave
constexpr int NL = 5;                // of which live in LDS ...
constexpr int NR = ROWS - NL;        // ... and in registers
constexpr int ROW_U4 = 64 * 3;       // one f16 row of a wave: 64 lanes x 3 x 16 B

// Tagged partial rows of the three barriers, behind the rows of the multi-pass chain (strm::MEGA_ROW_BASE): every barrier
// owns 8 partial rows (>= 128 KB: part_stride >= 4096), block b's record is the REC bytes at byte offset b * REC of that
// area and chunk c of its ceil(NV / 3) chunks the 16 bytes at + 16 c.  One record per 256-byte block of memory: 512
// pollers read every record several times, and packed records (8 KB for all of them) would queue on one or two
// memory channels.  A record is written once per launch and polled until its tags match; the tag of a launch is
// FP_EPOCH + 1, and block 0 advances FP_EPOCH as its last action (every block has read it long before: it has passed
// all barriers by then).  The workspace must be zero-filled once (include/mi_isp.h): tag 1 of the first launch then
// never matches stale memory.
#ifndef MI_MEGA_REC
#define MI_MEGA_REC 256
#endif
constexpr int REC = MI_MEGA_REC;
constexpr int MROW_BAR0 = strm::MEGA_ROW_BASE, MROW_BAR1 = MROW_BAR0 + 8, MROW_BAR2 = MROW_BAR1 + 8;
constexpr int MROW_END = MROW_BAR2 + 8;
static_assert(MROW_END <= strm::PART_ROWS, "the whole-frame kernel's rows must fit the workspace");
static_assert(REC >= 64 && REC % 16 == 0, "three 16-byte chunks and the beacon per record");
constexpr int FP_EPOCH = 60, FP_ERROR = 62;           // uint32 words inside FrameParams (slots no pass uses)

// cache policy bits of the barrier's posts and polls.  16 = sc1.  Measured per frame (unit / non-unit): posts 17 (sc0 sc1)
// 44.60 / 52.39 and 18 (nt sc1) 44.84 / 52.91, polls 17 44.58 / 52.27 and 18 45.57 / 54.33, both 17 44.61 / 52.26, both 19
// 45.99 / 55.02, against 44.58 / 52.42 as shipped: nothing to gain.
#ifndef MI_MEGA_POST_AUX
#define MI_MEGA_POST_AUX 16
#endif
#ifndef MI_MEGA_POLL_AUX
#define MI_MEGA_POLL_AUX 16
#endif
// check word of a record chunk: two multiplies and a shift (a torn chunk passes only if the words that differ collide
// in 32 bits of a multiplicative hash)
MI_DEV uint32_t rec_hash(uint32_t x, uint32_t y, uint32_t z) {
  uint32_t h = ((x * 0x9E3779B1u) ^ y) * 0x85EBCA6Bu ^ z;
  return h ^ (h >> 15);
}
constexpr uint32_t BEACON_OFF = 48u + 12u;   // fourth chunk of a record: {0, 0, 0, tag} in the clear, for the watch stage only

struct MArgs {
  SArgs s;                           // geometry and parameters shared by the frames of a launch (s.t.src / dst / fp / partials
                                     // and s.fp_w are per frame: FrameIO)
  unsigned spin_limit;               // polls before a wave gives up (error flag, garbage frame, no hang)
  unsigned poll_sleep;               // units of 512 cycles between two polls of the partial rows
  unsigned l2_first;                 // the first round of a fold reads through the L2
  unsigned* mailbox;                 // host-mapped word of the device: stored to when a barrier times out (no sync needed to see it)
  uint32_t launch_id;                // the host's count of whole-frame launches: part of the tag (see the frame loop)
  int sabotage_block;                // tests: this block does not post at barrier 0 of the launch's first frame (-1: none)
};

// One launch takes a BATCH of frames (same size and parameters), one after the other: the grid stays resident, a wave
// that has stored its rows of frame f goes straight on to its rows of frame f + 1.  Per launch instead of per frame:
// dispatch, the decode table, drain and launch gap (round 2 paid ~5 us of them per frame).
constexpr int MAX_BATCH = 64;
struct FrameIO {
  const void* src;                   // packed frame
  void* dst;                         // output image
  float* ws;                         // the frame's own workspace (FrameParams, epoch / error words, barrier records)
};
struct MBatch {
  MArgs m;
  int n_frames;
  FrameIO io[MAX_BATCH];
};

// Wave reductions for the posts and folds - the code every wave of the chip waits for.  The same DPP tree as
// isp_common.h's wave_min / wave_max / wave_sum (same order, same bits), but ONE instruction per step (v_min_f32_dpp ...)
// instead of four: written in C, every step became v_max x, x (the compiler quiets possible signalling NaNs around
// fminf / fmaxf of values it cannot trace) + v_mov_dpp + v_max + v_min.  The s_nop 1 in front of each step is the DPP
// hazard (two wait states after the VALU write of the source), which nobody inserts inside inline assembly.
#define MI_DPP_STEP(OP, CTRL) "s_nop 1\n\t" OP " %0, %0, %0 " CTRL "\n\t"
#define MI_DPP_TREE(OP)                                                                       \
  MI_DPP_STEP(OP, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")                           \
  MI_DPP_STEP(OP, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")                           \
  MI_DPP_STEP(OP, "row_half_mirror row_mask:0xf bank_mask:0xf")                               \
  MI_DPP_STEP(OP, "row_mirror row_mask:0xf bank_mask:0xf")                                    \
  MI_DPP_STEP(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf")                                  \
  MI_DPP_STEP(OP, "row_bcast:31 row_mask:0xc bank_mask:0xf")
MI_DEV float wmin(float v) {
  asm volatile(MI_DPP_TREE("v_min_f32_dpp") : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
MI_DEV float wmax(float v) {
  asm volatile(MI_DPP_TREE("v_max_f32_dpp") : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
MI_DEV float wsum(float v) {
  asm volatile(MI_DPP_TREE("v_add_f32_dpp") : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef MI_DPP_TREE
#undef MI_DPP_STEP

#ifndef MI_MEGA_FLAG_SLEEP            /* units of 64 cycles between two looks at the block's LDS flag; swept: 0 43.70, 1 43.86, 2 43.66, 4 43.77 us */
#define MI_MEGA_FLAG_SLEEP 2
#endif
// The block's contribution to a grid-wide reduction = its arrival at the barrier: every wave reduces in registers and
// leaves its row in LDS; the wave that arrives last combines the rows in wave order and stores the block's chunks
// {3 values, tag} write-through.  No wait, no counter, no workgroup barrier.
template <int NV>
MI_DEV void block_reduce_post(const float (&v)[NV], const int (&op)[NV], float (*red)[16], unsigned* arrived,
                              float* area, int stride, int block, int wave, int lane, uint32_t tag, int n_live = NV,
                              bool withhold = false) {
  constexpr int NCH = (NV + 2) / 3;
  float r[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {                      // values from n_live on are known to be zero (wave-uniform): not reduced
    if (k < n_live) r[k] = op[k] == 0 ? wmin(v[k]) : (op[k] == 1 ? wmax(v[k]) : wsum(v[k]));
    else r[k] = 0.f;
  }
  unsigned before = 0;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) red[wave][k] = r[k];
    before = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  before = __builtin_amdgcn_readfirstlane(before);
  if ((before & (WAVES - 1)) == WAVES - 1) {          // wave-uniform: this wave arrived last (the count runs on over the phases)
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 mine = {0u, 0u, 0u, tag};                       // lane c holds chunk c, lane NCH the beacon {0, 0, 0, tag}
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      float x = red[0][k];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) {
        const float o = red[w][k];
        x = op[k] == 0 ? fminf(x, o) : (op[k] == 1 ? fmaxf(x, o) : x + o);
      }
      if (lane == k / 3) mine[k % 3] = __builtin_bit_cast(uint32_t, x);
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(area, 0, (int)((size_t)8 * stride * sizeof(float)), 0x00020000);
    if (lane < NCH) mine.w = tag ^ rec_hash(mine.x, mine.y, mine.z);
    const uint32_t off = withhold ? INVALID_OFF
                                  : (lane < NCH ? (uint32_t)block * REC + 16u * lane : (lane == NCH ? (uint32_t)block * REC + 48u : INVALID_OFF));
    __builtin_amdgcn_raw_buffer_store_b128(mine, rs, off, 0, MI_MEGA_POST_AUX);   // aux 16 = sc1: write-through, visible to the other XCDs
  }
}

// LDS of the barrier folds: per barrier a ticket that hands out the roles, the partial folds of the four roles, a count
// of the roles that are done and the flag the block waits for.
struct FoldLds {
  unsigned ticket[4], done[4], flag[4];   // running counts over the frames of a launch: role = ticket % WAVES, flag = frames passed
  float mm[WAVES][4];
  float sum[WAVES][5];
  float keep[7];                          // barrier 0's totals of the speculative statistics, for barrier 1 (bounds other than (0, 1))
  unsigned faulted;                       // latched by the first wave of the block whose poll budget runs out (barrier_fold: budget)
};

// Wait for the barrier whose records live in `area` and derive the scalars of the next phase.  Every wave of the block
// takes a role r = its arrival order at this barrier and owns the records of blocks [128 r, 128 r + 128), two per lane:
// it polls all their chunks (6 loads in flight, only the chunks still missing are asked for again: an out-of-range
// offset makes no memory request), folds them in block order and leaves the partial fold in LDS; the role that finishes
// last combines the four partial folds in role order - identical arithmetic in every block - and publishes the scalars.
// NV values per block: 2 = {min, max}; 7 = statistics; 9 = bounds followed by the speculative statistics (finalized
// only when the bounds turn out to be (0, 1)).
// Plain sc1 buffer loads (aux 16), compiler-visible: an asm load would hand its destination registers back to the
// allocator while the data is still in flight (that was a memory fault), and atomic loads are waited for one by one.
template <int NV, int FIN>
MI_DEV void barrier_fold(const MArgs& m, float* ws, unsigned seq, int bar, const float* area, uint32_t tag, float* sh_fp,
                         FoldLds& fl, int lane, bool rgb_sums = true, unsigned* stamps = nullptr) {
  const SArgs& a = m.s;
  constexpr int NCH = (NV + 2) / 3;
  constexpr int NMM = NV == 9 ? 4 : (NV == 1 ? 0 : 2);   // leading min / max values (alternating)
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  unsigned role = 0;
  if (lane == 0) role = __hip_atomic_fetch_add(fl.ticket + bar, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  role = __builtin_amdgcn_readfirstlane(role) & (WAVES - 1);
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(area), 0, (int)((size_t)8 * a.t.part_stride * sizeof(float)), 0x00020000);
  u4 v[2][NCH];
  bool have[2][NCH];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      v[u][c] = u4{0u, 0u, 0u, 0u};
      // the per-channel sums are the last chunk: nobody posted any when color_adapt == 0 (wave-uniform), skip it
      have[u][c] = (int)role * 128 + u * 64 + lane >= a.n_blocks || (NV >= 7 && c == NCH - 1 && !rgb_sums);
    }
  unsigned spins = 0;
  // The poll budget.  Once a wave of this block has run out of it (FoldLds::faulted, latched for the rest of the launch)
  // the block's budget is ONE round per barrier: the launch is lost for it, and it walks through what is left - posting,
  // so that nobody waits for it, and marking the fault word of every frame whose records it does not find - instead of
  // spending ~100 ms again at each of the 3 x 64 barriers to come (round 3 did: one disturbance at launch turned a 2.8 ms
  // launch into tens of seconds of a spinning resident grid).  Leaving the launch altogether was built first: an exit from
  // the middle of the frame loop cost 24 - 72 spilled registers in every instantiation.
  const unsigned budget = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&fl.faulted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                              ? 1u : m.spin_limit;
  // Stage 1: all records of this role.  The first round is served by the XCD's L2 (sc0: only the CU's own cache is
  // bypassed): by now almost every record has been in memory for a while, the first wave of an XCD to ask brings a line
  // in and the other 255 waves hit it - 2048 waves asking memory for 256 chunks each took 1.6 us.  A line the L2 holds
  // from before its record was posted shows the old tag; such chunks, and every later round, bypass the L2 (sc1).
  auto round = [&](auto auxc) {
    constexpr int AUX = decltype(auxc)::value;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const uint32_t off = have[u][c] ? INVALID_OFF : (uint32_t)((int)role * 128 + u * 64 + lane) * REC + 16u * c;
        const u4 t = __builtin_amdgcn_raw_buffer_load_b128(prs, off, 0, AUX);
        v[u][c] = have[u][c] ? v[u][c] : t;
      }
    bool all = true;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        have[u][c] = have[u][c] || (v[u][c].w ^ rec_hash(v[u][c].x, v[u][c].y, v[u][c].z)) == tag;
        all = all && have[u][c];
      }
    return __builtin_amdgcn_ballot_w64(!all);
  };
  unsigned long long missing = ~0ull;
  // Stage 0: watch TWO typical records (lanes 0 and 1: the last block of the first half of the grid and the third last
  // block) until both are there - by then nine blocks in ten have posted.  2048 waves polling all records while most
  // blocks are still in their phase slow those down (their loads and posts queue behind the polls: posts took up to
  // 4.6 us instead of 1); two requests per wave and round do not.  Four polls in flight, a new one every 512 cycles.
  // Whichever blocks post last are deliberately NOT waited for here: stage 1 asks for whatever is still missing every
  // round, so a straggler's record is seen one round trip after it lands.
  if (budget > 1) {                                   // (a budget of 1 - the fault test - polls at once)
    const int watch = lane == 0 ? a.n_blocks / 2 - 1 : a.n_blocks - 3;
    const uint32_t off = lane < 2 && watch >= 0 ? (uint32_t)watch * REC + BEACON_OFF : INVALID_OFF;
    const bool idle = !(lane < 2 && watch >= 0);
    auto ask = [&]() { return __builtin_amdgcn_raw_buffer_load_b32(prs, off, 0, MI_MEGA_POLL_AUX); };
    auto nap = [&]() { for (unsigned z = 0; z <= m.poll_sleep; ++z) __builtin_amdgcn_s_sleep(8); };
    auto there = [&](uint32_t q) { return __builtin_amdgcn_ballot_w64(!idle && q != tag) == 0; };
    uint32_t q0 = ask(); nap();
    uint32_t q1 = ask(); nap();
    uint32_t q2 = ask(); nap();
    uint32_t q3 = ask();
    for (;;) {
      if (there(q0)) break;
      nap(); q0 = ask();
      if (there(q1)) break;
      nap(); q1 = ask();
      if (there(q2)) break;
      nap(); q2 = ask();
      if (there(q3)) break;
      nap(); q3 = ask();
      if ((spins += 4) > budget) break;         // stage 1 raises the error
    }
  }
  // (Measured and taken out: a wave that found both watched records at its first look - a late arrival - skipping the L2
  // round, on the theory that what its XCD's L2 holds of the records is mostly stale: 44.02 us per frame against 43.83.
  // The L2 round pays for the late waves too.)
  missing = m.l2_first && budget > 1 ? round(std::integral_constant<int, 1>{}) : ~0ull;
  while (missing != 0) {
    // The budget is checked BEFORE a round.  A budget of 1 (tests/: mi_isp_whole_frame_set_poll_limit(1)) therefore means
    // "one round, straight after the wave's own post, no watch stage": the first block to arrive cannot find the others'
    // records there, so at least one block reports the fault - deterministically.
    if (spins >= budget) {                      // a peer is not resident: give up loudly instead of hanging
      if (lane == 0) {
        __hip_atomic_store(reinterpret_cast<unsigned*>(ws) + FP_ERROR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m.mailbox) __hip_atomic_store(m.mailbox, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // latched: from the next barrier on the block's budget is one round
        __hip_atomic_store(&fl.faulted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      break;
    }
    ++spins;
    // many records missing: the phase is still running elsewhere, poll rarely; few: the last arrivals, poll at once
    const unsigned naps = spins > 1 && __builtin_popcountll(missing) > 16 ? 4u * m.poll_sleep + 1u : m.poll_sleep;
    for (unsigned z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(8);
    missing = round(std::integral_constant<int, MI_MEGA_POLL_AUX>{});
  }
  if (stamps && lane == 0 && role == 0) stamps[0] = MI_STAMP_NOW();
  // The fold: fp32 (round 2 folded the sums in fp64 - 0.8 us of DPP trees per barrier with every wave of the chip
  // waiting; 512 fp32 partials summed pairwise lose ~1e-7 relative, the scalars' contract is 1e-4).  The order is fixed
  // (lanes, then roles), so every block derives the same bits.
  float mm[NMM > 0 ? NMM : 1];
  float sum[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NMM; ++k) mm[k] = (k & 1) ? -__builtin_inff() : __builtin_inff();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const bool real = (int)role * 128 + u * 64 + lane < a.n_blocks;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const uint32_t w3[3] = {v[u][c].x, v[u][c].y, v[u][c].z};   // (element access by a loop index was miscompiled)
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int k = 3 * c + e;
        if (k < NV) {
          const float x = __builtin_bit_cast(float, w3[e]);
          if (k < NMM) { if (real) mm[k] = (k & 1) ? fmaxf(mm[k], x) : fminf(mm[k], x); }
          else sum[k - NMM] += real ? x : 0.f;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NMM; ++k) mm[k] = (k & 1) ? wmax(mm[k]) : wmin(mm[k]);
  if (NV == 1) sum[0] = wsum(sum[0]);
  if (NV >= 7) {
    sum[0] = wsum(sum[0]); sum[1] = wsum(sum[1]);
    if (rgb_sums) {                                    // else they are zero
#pragma unroll
      for (int k = 2; k < 5; ++k) sum[k] = wsum(sum[k]);
    }
  }
  unsigned finished = 0;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NMM; ++k) fl.mm[role][k] = mm[k];
#pragma unroll
    for (int k = 0; k < 5; ++k) fl.sum[role][k] = sum[k];
    finished = __hip_atomic_fetch_add(fl.done + bar, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  finished = __builtin_amdgcn_readfirstlane(finished) & (WAVES - 1);
  if (stamps && lane == 0 && role == 0) stamps[1] = MI_STAMP_NOW();
  if (finished == WAVES - 1) {
    // ---- the last role combines the partial folds (role order) and publishes the scalars ----
    if (lane == 0) {
      float tot[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < NMM; ++k) {
        float x = fl.mm[0][k];
#pragma unroll
        for (int r = 1; r < WAVES; ++r) x = (k & 1) ? fmaxf(x, fl.mm[r][k]) : fminf(x, fl.mm[r][k]);
        tot[k] = x;
      }
      if (NV >= 7 || NV == 1) {
#pragma unroll
        for (int k = 0; k < (NV == 1 ? 1 : 5); ++k) tot[NMM + k] = ((fl.sum[0][k] + fl.sum[1][k]) + fl.sum[2][k]) + fl.sum[3][k];
      }
      ew::FinArgs fa = {};
      // (laundered: what the finalize derives from these alone - 1 / n, exp(-intensity) - is loop-invariant, and hoisted out
      // of the frame loop it is spilled at the kernel's entry and reloaded here, in every block, for every frame)
      float npx = a.n_px, intensity = a.intensity;
      asm volatile("" : "+s"(npx), "+s"(intensity));
      fa.fp = sh_fp; fa.n_px = npx; fa.intensity = intensity; fa.la = a.t.la; fa.ca = a.t.ca;
      fa.bounds_post = a.bounds_post;
      if constexpr (NV == 9) {
        ew::finalize_scalars_fast(ew::FIN_BOUNDS, fa, tot);
        if (sh_fp[FP_LO] == 0.f && sh_fp[FP_INV] == 1.f) {
          ew::finalize_scalars_fast(ew::FIN_STATS, fa, tot + 2);
        } else {
#pragma unroll
          for (int k = 0; k < 7; ++k) fl.keep[k] = tot[2 + k];     // raw gray min / max, (sum log2), sum gray, channel sums
        }
      } else if constexpr (NV == 1) {
        // The statistics of the NORMALISED image (tonemap.py:147-149) from those of the raw one: n = (x - lo) * inv is
        // affine, the weights of rgb_gray sum to 1, and the clamp to [0, 1] is the identity between the image's own
        // bounds - so gray(n) = (gray(x) - lo) * inv, its min / max and every sum follow from the raw ones (within fp32
        // rounding: ~1e-7 relative, the contract of these scalars is 1e-4); only the sum of logarithms needs the pixels
        // (phase B: tot[0]).  fp32 throughout: the sums are fp32 sums of 12.6 M values to begin with (~1e-7 relative),
        // and n * lo takes away at most the part of them that the image's minimum accounts for.
        const float lo = sh_fp[FP_LO], inv = sh_fp[FP_INV], nlo = npx * sh_fp[FP_LO];
        float t7[7];
        t7[0] = (fl.keep[0] - lo) * inv; t7[1] = (fl.keep[1] - lo) * inv;
        t7[2] = tot[0];
#pragma unroll
        for (int k = 3; k < 7; ++k) t7[k] = (fl.keep[k] - nlo) * inv;
        ew::finalize_scalars_fast(ew::FIN_STATS, fa, t7);
      } else {
        ew::finalize_scalars_fast(FIN, fa, tot);
      }
      if (stamps) stamps[2] = MI_STAMP_NOW();
      __hip_atomic_store(fl.flag + bar, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      // block 0 leaves the frame's scalars in FrameParams, as the multi-pass chain does (callers may read them back)
      if (blockIdx.x == 0) {
        for (int i = 0; i <= FP_MAXOUT; ++i) ws[i] = sh_fp[i];
      }
    }
  } else {
    unsigned naps = 0;
    while ((int)(__hip_atomic_load(fl.flag + bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - seq) < 0) {
      __builtin_amdgcn_s_sleep(MI_MEGA_FLAG_SLEEP);
      if (++naps > 64u * m.spin_limit) break;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// 24 f16 values of a resident row (12 packed registers) -> fp32
MI_DEV void unpack_row(const uint32_t (&pk)[12], float (&t)[24]) {
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    half_t h[2];
    __builtin_memcpy(h, &pk[j], 4);
    t[2 * j] = (float)h[0]; t[2 * j + 1] = (float)h[1];
  }
}

// rgb_gray (color/__init__.py:7-10) of pixel K of a packed row, straight from the f16 halves: three v_fma_mix_f32 (the
// half is widened exactly, the fma rounds once in fp32 - the bits of converting first and then mul / fma / fma, without
// the three conversions).  w0..w2: the weights in VGPRs.
template <int E> MI_DEV float fma_mix_h(uint32_t pk, float w, float acc) {
  float r;
  if constexpr ((E & 1) == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(w), "v"(acc));
  else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(w), "v"(acc));
  return r;
}
template <int E> MI_DEV float mul_mix_h(uint32_t pk, float w) {
  float r;
  if constexpr ((E & 1) == 0) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(w));
  else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(w));
  return r;
}
template <int K> MI_DEV float gray_pk(const uint32_t (&pk)[12], float w0, float w1, float w2) {
  constexpr int e = 3 * K;
  const float a = mul_mix_h<e>(pk[e / 2], w0);
  const float b = fma_mix_h<e + 1>(pk[(e + 1) / 2], w1, a);
  return fma_mix_h<e + 2>(pk[(e + 2) / 2], w2, b);
}

// "Defines" registers without an instruction.  The resident rows are written and read under wave-uniform conditions (a
// band may end before its 12th row).  Inside the frame loop a variable that is assigned under a condition carries, for
// the compiler, its value of the PREVIOUS frame along the other arm - all 252 registers of resident rows would be live
// around the loop.  An unconditional (empty) definition right before the conditional one cuts that.
// (__builtin_nondeterministic_value = a frozen undefined value: no instruction; an empty asm with an output costs a
// v_mov and a hazard s_nop each.)
template <class T, int N> MI_DEV void fresh(T (&x)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = __builtin_nondeterministic_value(x[i]);
}

// Issue priority of the wave.  The two blocks of a CU put one wave each on every SIMD, and between two waves that are
// both ready the hardware picks the OLDER one - the block that was dispatched first.  Measured (scripts/wf_skew.py): the
// waves of blocks 0 .. n/2 - 1 end phase A 3.6 us before those of blocks n/2 .. n - 1, frame after frame (correlation
// 0.91), then idle at the barrier while their SIMD runs the other wave alone - at half its issue rate.  The waves
// therefore swap priority every row of phases A, B and C: `turn` counts rows, `younger` is the block's half; whoever has
// the turn wins the ties.  46.9 -> 44.8 us per frame (a turn per row PAIR of phase A: 46.0; none in phase C: + 1.1; two
// turns per row - one for the demosaic, one for the rest - 46.9: worse than none); with a turn per pair the halves ended
// phase A 0.7 us apart (3.8 before).  Also measured: slices of the CU's clock instead of turns (46.0 with 8192-cycle
// slices, 46.4 with 2048, 47.8 with 32768), the younger block always first (the asymmetry flips), priority level 3
// instead of 1 (the same), turns in phase D (nothing).
MI_DEV void prio_turn(int turn, bool younger) {
  if (((turn & 1) != 0) == younger) asm volatile("s_setprio 1");
  else asm volatile("s_setprio 0");
}

// bounds of a packed f16 row: four values per instruction (gfx950: v_pk_minimum3_f16 / v_pk_maximum3_f16; the values are
// clamped and finite, so the NaN rule of the IEEE-2019 minimum does not come into play)
MI_DEV void pk_bounds_row(const uint32_t (&pk)[12], uint32_t& mn, uint32_t& mx) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    asm("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(mn) : "v"(pk[2 * j]), "v"(pk[2 * j + 1]));
    asm("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(mx) : "v"(pk[2 * j]), "v"(pk[2 * j + 1]));
  }
}
MI_DEV float pk_lo(uint32_t v) { half_t h[2]; __builtin_memcpy(h, &v, 4); return (float)h[0]; }
MI_DEV float pk_hi(uint32_t v) { half_t h[2]; __builtin_memcpy(h, &v, 4); return (float)h[1]; }

// where phase D issues the first loads of the NEXT frame of the batch: behind resident row PREFETCH_AT (registers have
// come free by then: 24 VGPRs per retired register row)

#ifndef MI_MEGA_PREFETCH_AT
#define MI_MEGA_PREFETCH_AT 3            /* register rows first: behind LDS row 3 */
#endif
#ifndef MI_MEGA_PREFETCH_AT_LATE
#define MI_MEGA_PREFETCH_AT_LATE 8       /* LDS rows first: behind register row 8 */
#endif
// cache policy bits of the output stores: 2 = nt (streamed).  Measured per frame: 0 (none) 49.9, 1 (sc0) 51.4, 2 (nt) 44.9,
// 3 (nt sc0) 44.8, 16 (sc1) 45.1, 17 (sc0 sc1) 45.1, 18 (nt sc1) 45.0, 19 (all) 45.0 us.
#ifndef MI_MEGA_ST_AUX
#define MI_MEGA_ST_AUX ST_STREAM
#endif
// LDS rows whose second Reinhard evaluation (it needs nothing from barrier 2) runs between a wave's post and its first
// poll of that barrier, i.e. inside the wait for the slowest block
#ifndef MI_MEGA_PRE2
#define MI_MEGA_PRE2 1
#endif

// RGB: color_adapt != 0 (per-channel sums in the statistics).  Two kernels instead of a run-time flag: with both kinds of
// statistics in one body the register allocator spilled 13 VGPRs to scratch in phase A and reloaded them in B / C / D,
// which cost 3.3 us per frame (58.2 -> 54.9).  Split, every variant fits 256 VGPRs without scratch - as long as the
// Reinhard dispatch below stays a run-time branch on `ca0` (made compile-time, the RGB = false kernel spilled 17):
// tests/test_abi.py::test_whole_frame_kernel_uses_no_scratch compiles the kernels and checks.
// (A variant without the row conditions for frames whose height is a multiple of 12 was tried: with no branches between
// them the phases' rows become one scheduling region, and the scheduler's reordering cost 225 - 279 spills.  The
// conditions stay.)
template <int PR, int PC, bool RGB>
__global__ __launch_bounds__(THREADS, 2) void frame_kernel(const MBatch mb) {
  typedef half_t E;
  const MArgs& m = mb.m;
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

  // Where a wave works.  Derived twice: here for the first frame's first loads, and again at the top of every frame from
  // laundered thread / block ids - so that NOTHING but the frame counter and the prefetched rows is live around the
  // frame loop.  (With the geometry computed once outside, the compiler hoisted the rows' offset arithmetic out of the
  // loop as well and the allocator, with every register taken inside the body, spilled 500 - 600 VGPRs.)
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
  // the loads a frame starts with: the four rows above / at the top of the band and the first two row pairs
  uint32_t pro[4][4];
  uint32_t raw[2][2][4];                              // row pairs in flight: two bodies ahead (~8k cycles) covers the latency
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
  first_loads(geo(threadIdx.x, blockIdx.x), mb.io[0].src);

  // ---- once per launch: the decode table ----
  for (int e = threadIdx.x; e < 4096; e += THREADS) lut[e] = tile::decode_scaled<E>((uint32_t)e, p.k_decode);
  __syncthreads();                                    // table, tickets, flags, `arrived`: the kernel's only workgroup barrier
  constexpr bool want_rgb = RGB;                        // the host picks the kernel by p.ca != 0

  const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // the thread's id, re-derived (lane from the exec mask, volatile: not hoisted) - so that not even threadIdx.x occupies
  // a register around the loop
  auto thread_id = [&]() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return wave_s * 64 + l;
  };
  for (int f = 0; f < mb.n_frames; ++f) {
  int tid_ = thread_id(), bid_ = blockIdx.x;
  asm volatile("" : "+s"(bid_));
  const Geo G = geo(tid_, bid_);
  const int lane = G.lane, wave = G.wave, g = G.g, bx = G.bx, r_begin = G.r_begin, r_end = G.r_end, active_lanes = G.active_lanes;
  const bool younger = bid_ >= (a.n_blocks >> 1);       // the second of the two blocks of its CU (dispatch order)
  const bool wave_ok = G.wave_ok, col_ok = G.col_ok, is_left = G.is_left, is_right = G.is_right, any_left = G.any_left,
             any_right = G.any_right;
  // output rows leave through a buffer resource (wave_store_units): unit j * 64 + lane of the band's row
  const int osz = (int)mi_dtype_size_dev(p.out_dtype);
  const int unit_bytes = osz == 1 ? 8 : 16, units_per_lane = 24 * osz / unit_bytes;
  const uint32_t out_pitch = (uint32_t)p.W * 3u * (uint32_t)osz, band_base = (uint32_t)bx * BAND * 3u * (uint32_t)osz;
  const FrameIO io = mb.io[f];
  const unsigned seq = (unsigned)f + 1u;              // what the LDS flags of this frame's barriers count up to
  float* const ws = io.ws;
  float* const partials = ws + FP_COUNT;
  const __amdgpu_buffer_rsrc_t rsrc = src_rsrc(io.src);
  const uint32_t epoch = __builtin_amdgcn_readfirstlane(reinterpret_cast<const unsigned*>(ws)[FP_EPOCH]);
  // The tag of this launch's records in this workspace: the workspace's own launch count (it covers graph replays, whose
  // kernel arguments are frozen) mixed with the host's launch count (it tells two launches apart even for a block that
  // came to life after block 0 of ITS launch had advanced the epoch - such a block now posts records no launch will ever
  // take for its own).  Never 0: a zero-filled workspace matches no launch.
  const uint32_t tag_ = m.launch_id * 0x9E3779B1u + epoch + 1u;
  const uint32_t tag = tag_ == 0u ? 1u : tag_;
  MI_MSTAMP(0);
#ifdef MI_STREAM_STAMPS
  unsigned* st_ = reinterpret_cast<unsigned*>(partials + (size_t)PART_ROWS * p.part_stride) + g * 16;
#ifdef MI_STAMP_HWID                                     // where the wave runs: HW_ID (SE / CU / SIMD / slot) and XCC_ID
  if (lane == 0 && wave_ok) { st_[15] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); st_[14] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); }
#endif
#endif
  // ================================ phase A: demosaic once ================================
  // The uniform operands of a phase are set up at ITS start, every frame (vgpr() is opaque, so they are neither hoisted
  // out of the frame loop nor kept alive through the other phases: this kernel has no registers to park them in).
  float wq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wq[i] = vgpr(wq_value(i));
  const float gw0 = vgpr(0.299f), gw1 = vgpr(0.587f), gw2 = vgpr(0.114f);
  WinRow win[6];
  uint32_t xr[NR][12];                                // resident rows NL..ROWS-1 (packed f16 pairs)
#pragma unroll
  for (int q = 0; q < 4; ++q) decode_row(pro[q], lut, lane, win[q]);
  MI_CMARK(10);
#if defined(MI_STREAM_STAMPS) && !defined(MI_STAMP_HWID)
  if (lane == 0 && wave_ok) st_[15] = MI_STAMP_NOW();      // prologue done: first four rows decoded
#endif
  uint32_t bmin = 0x7C007C00u, bmax = 0xFC00FC00u;    // packed f16 {+inf, +inf} / {-inf, -inf}
  Stats2 st; st.init();

  static_for<0, ROWS / 2>([&](auto ibc) {
    constexpr int IB = decltype(ibc)::value, PH = IB % 3;
    const int r = r_begin + 2 * IB;
    decode_row(raw[IB % 2][0], lut, lane, win[(2 * PH + 4) % 6]);
    decode_row(raw[IB % 2][1], lut, lane, win[(2 * PH + 5) % 6]);
    if constexpr (IB + 2 < ROWS / 2) {
      load_row(G, rsrc, r + 6, raw[IB % 2][0]);
      load_row(G, rsrc, r + 7, raw[IB % 2][1]);
    }
#if defined(MI_STREAM_STAMPS) && !defined(MI_STAMP_HWID)
    if constexpr (IB == 3) { if (lane == 0 && wave_ok) st_[14] = MI_STAMP_NOW(); }      // half of the rows done
#endif
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
        if (MI_CENSUS(row < 2 || row >= p.H - 2, false)) border_fix_rows<PR, PC, I>(v, tile::inside_mask(row, p.H), is_left, is_right);
        else if (MI_CENSUS(any_left || any_right, false)) border_fix_cols<PR, PC, I>(v, is_left, is_right, any_left, any_right);
        if (MI_CENSUS(p.has_ccm, false)) {            // bayer.py:152-153, sequential fp32 dot
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float x = v[3 * k], y = v[3 * k + 1], z = v[3 * k + 2];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
              v[3 * k + ch] = (p.ccm[3 * ch] * x + p.ccm[3 * ch + 1] * y) + p.ccm[3 * ch + 2] * z;
          }
        }
        // the pixel as the reference materialises it: clamped (bayer.py:155), rounded to f16
        uint32_t pk[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) pk[j] = tile::cvt_pk_f16_clamp01(v[2 * j], v[2 * j + 1]);
        // bounds of the image (tonemap.py:146) on the rounded values, four per instruction
        pk_bounds_row(pk, bmin, bmax);
        // statistics of tonemap.py:147-149 under the assumption that the bounds are (0, 1)
        if constexpr (want_rgb) {
          float t[24];
          unpack_row(pk, t);
          st.add8<true>(t);
        } else {
          float g8[8];
          static_for<0, 8>([&](auto kc) { constexpr int K = decltype(kc)::value; g8[K] = gray_pk<K>(pk, gw0, gw1, gw2); });
          st.add8_gray(g8);
        }
        // the pixels stay on the chip
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

  MI_MSTAMP(1);
  float* rows_bounds = partials + (size_t)MROW_BAR0 * p.part_stride;
  float* rows_stats = partials + (size_t)MROW_BAR1 * p.part_stride;
  float* rows_bounds2 = partials + (size_t)MROW_BAR2 * p.part_stride;
  float vmin, vmax;
  {
    // gray min / max are posted as they are: the clamp max(gray, 1e-4) of tonemap.py:86 (monotone) is applied to the
    // reduced values by the finalize step, and the raw values serve the affine route of barrier 1
    vmin = fminf(pk_lo(bmin), pk_hi(bmin)); vmax = fmaxf(pk_lo(bmax), pk_hi(bmax));
    if (!col_ok) { vmin = __builtin_inff(); vmax = -__builtin_inff(); st.init(); }
    const float v9[9] = {vmin, vmax, st.gmin, st.gmax, st.slog, st.sgray, st.s0, st.s1, st.s2};
    const int op[9] = {0, 1, 0, 1, 2, 2, 2, 2, 2};
    block_reduce_post<9>(v9, op, red, &arrived, rows_bounds, p.part_stride, blockIdx.x, wave, lane, tag, want_rgb ? 9 : 6,
                         f == 0 && m.sabotage_block == (int)blockIdx.x);
  }

  // the resident row RR: packed, and as 24 fp32 values
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
  auto resident = [&](auto rrc, float (&t)[24]) {
    uint32_t pk[12];
    resident_pk(rrc, pk);
    unpack_row(pk, t);
  };

  // ================================ barrier 0: bounds (tonemap.py:146) ================================
  MI_MSTAMP(2);
#ifdef MI_STREAM_STAMPS
  barrier_fold<9, ew::FIN_BOUNDS>(m, ws, seq, 0, rows_bounds, tag, sh_fp, fl, lane, want_rgb, st_ + 9);
#else
  barrier_fold<9, ew::FIN_BOUNDS>(m, ws, seq, 0, rows_bounds, tag, sh_fp, fl, lane, want_rgb);
#endif
  MI_MSTAMP(3);
  const float lo_s = sh_fp[FP_LO], inv_s = sh_fp[FP_INV];
  const bool unit = MI_CENSUS(lo_s == 0.f && inv_s == 1.f, true);
  // per-pixel operands live in VGPRs: a VALU instruction with an SGPR operand issues at half rate
  const float lo = vgpr(lo_s), inv = vgpr(inv_s);
  // the normalisation of tonemap.py:13 in phases C and D; only frames whose bounds are not (0, 1) come here.  (As ONE fma,
  // x * inv - lo * inv, measured on one box, unit / non-unit frames: 44.05 / 50.13 us with it, 43.81 / 50.90 without - the
  // frames it is not executed for pay 0.24 us for the other arm's different register allocation.  Not taken: the headline
  // is the unit frame.)
  auto norm_fma = [&](float x) __attribute__((always_inline)) { return norm01(x, lo, inv); };
  if (!unit) {
    // ============================ phase B: the statistics for bounds other than (0, 1) ============================
    // Only the sum of log(gray) needs the pixels again; the other statistics of the normalised image follow from the
    // raw ones of phase A (barrier_fold<1>).  gray(n) = (gray(x) - lo) * inv: see there.
    float sl0 = 0.f, sl1 = 0.f;
    const float gwi0 = vgpr(0.299f * inv_s), gwi1 = vgpr(0.587f * inv_s), gwi2 = vgpr(0.114f * inv_s), gwc = vgpr(-lo_s * inv_s);
    static_for<0, ROWS>([&](auto rrc) {
      constexpr int RR = decltype(rrc)::value;
      prio_turn(RR, younger);
      if (r_begin + RR < r_end) {
        uint32_t pk[12];
        resident_pk(rrc, pk);
        // one logarithm per row: the product of its eight clamped gray values (Stats2::add8_gray).  The normalised gray in
        // three instructions: (gray(x) - lo) * inv = (w0 inv) r + (w1 inv) g + (w2 inv) b - lo inv, the constant as the first
        // fma's addend (differs from subtract-then-multiply by ~1e-7 relative; the scalars' contract is 1e-4)
        float c[8];
        static_for<0, 8>([&](auto kc) {
          constexpr int K = decltype(kc)::value, e = 3 * K;
          const float a0 = fma_mix_h<e>(pk[e / 2], gwi0, gwc);
          const float a1 = fma_mix_h<e + 1>(pk[(e + 1) / 2], gwi1, a0);
          c[K] = fmaxf(fma_mix_h<e + 2>(pk[(e + 2) / 2], gwi2, a1), 1e-4f);
        });
        sl0 += hw_log2(((c[0] * c[1]) * (c[2] * c[3])) * ((c[4] * c[5]) * (c[6] * c[7])));
      }
    });
    const float v1[1] = {col_ok ? sl0 + sl1 : 0.f};
    const int op[1] = {2};
    block_reduce_post<1>(v1, op, red, &arrived, rows_stats, p.part_stride, blockIdx.x, wave, lane, tag);
    barrier_fold<1, ew::FIN_STATS>(m, ws, seq, 1, rows_stats, tag, sh_fp, fl, lane);
  }
  MI_MSTAMP(4);
  ReinhardK rk;
  const bool ca0 = MI_CENSUS(p.ca == 0.f, true);        // runtime on purpose: see the note on register allocation at the kernel's head
  rk.la = vgpr(p.la);
  rk.map_key = vgpr(sh_fp[FP_MAPKEY]); rk.ei = vgpr(sh_fp[FP_EI]);
  rk.mean3[0] = vgpr(sh_fp[FP_MEAN3]);
  if constexpr (RGB) {
    rk.ca = vgpr(p.ca); rk.mean3[1] = vgpr(sh_fp[FP_MEAN3 + 1]); rk.mean3[2] = vgpr(sh_fp[FP_MEAN3 + 2]);
  } else {
    // this kernel is only launched with color_adapt == 0 (mean3 is then the same for the three channels, tonemap.py:119):
    // the operands of the other arm - never executed here - share registers instead of taking three more
    rk.ca = rk.la; rk.mean3[1] = rk.mean3[0]; rk.mean3[2] = rk.mean3[0];
  }

  // Reinhard of one resident row (tonemap.py:120-131): q[24].  UNIT: bounds exactly (0, 1), the normalisation is the
  // identity; CA0: color_adapt == 0, one pow per pixel.  The variant is chosen ONCE per phase, outside the row loops: the
  // executed code of a phase is then one contiguous stretch (the instruction cache is shared by two CUs and a wave's
  // straight-line code is ~100 KB; interleaved dead variants cost misses).
  // (Measured and taken out, round 4 - profiles/r04_barrier_and_phaseC_experiments.txt: the Reinhard chains of 2 or 4
  // pixels in LOCKSTEP, stage by stage between scheduling barriers.  The 146 hazard s_nop of phase C went, the time did
  // not: 44.19 - 44.44 against 44.30 us per frame - with two waves per SIMD the other wave already fills the slots a
  // dependent chain leaves open; phase C's ~10 us are its instruction count (440 transcendentals at 7.4 cycles, ~610
  // single-rate and ~1260 double-rate instructions per wave), not its order.  Likewise the gray of the resident pixels
  // computed ahead, in barrier 0's wait: two rows' worth fit the registers, nothing measurable.)
  auto tone_row = [&](auto unit_c, auto ca0_c, const float (&t)[24], float (&q)[24]) {
    constexpr bool UNIT = decltype(unit_c)::value, CA0 = decltype(ca0_c)::value;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float x[3], o[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) x[ch] = UNIT ? t[3 * k + ch] : norm_fma(t[3 * k + ch]);
      reinhard_px<CA0>(x, rk, o);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) q[3 * k + ch] = o[ch];
    }
  };
  auto dispatch = [&](auto&& phase) {
    if (ca0) {
      if (unit) phase(std::true_type{}, std::true_type{});
      else phase(std::false_type{}, std::true_type{});
    } else {
      phase(std::false_type{}, std::false_type{});
    }
  };

  // ================================ phase C: bounds of the mapped image (tonemap.py:150-153) ================================
  // The mapped values of the register-resident rows are KEPT (24 fp32 registers per row take the place of the 12 packed
  // ones, which phase D no longer needs): phase D then only normalises and stores them, and recomputes Reinhard for
  // the LDS rows alone.
  vmin = __builtin_inff(); vmax = -__builtin_inff();
  float qr_[NR][24];
  // An LDS row's mapped values are not kept, only their bounds are wanted - and with color_adapt == 0 the three channels
  // of a pixel share the adaptation term ad, q = t / (ad + t) is increasing in t, so the pixel's smallest and largest
  // mapped value are those of its smallest and largest channel: two reciprocals per pixel instead of three (the value
  // is the same instruction sequence on the same operand; where the hardware reciprocal is not monotone to the last
  // bit the bound can differ from the three-channel one by an ulp - the scalars' contract is 1e-4).
  auto tone_bounds_row = [&](auto unit_c, const float (&t)[24]) {
    constexpr bool UNIT = decltype(unit_c)::value;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float x[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) x[ch] = UNIT ? t[3 * k + ch] : norm_fma(t[3 * k + ch]);
      const float ad = reinhard_adapt_ca0(x, rk);         // (the very function phase D evaluates again)
      const float xmin = fminf(x[0], fminf(x[1], x[2])), xmax = fmaxf(x[0], fmaxf(x[1], x[2]));
      vmin = fminf(vmin, reinhard_map(xmin, ad));
      vmax = fmaxf(vmax, reinhard_map(xmax, ad));
    }
  };
  static_for<0, ROWS>([&](auto rrc) {
    constexpr int RR = decltype(rrc)::value;
    if constexpr (RR >= NL) fresh(qr_[RR - NL]);
    prio_turn(RR, younger);
    if (r_begin + RR < r_end) {
      float t[24];
      resident(rrc, t);
      if (RR < NL && ca0) {                            // (wave-uniform)
        if (unit) tone_bounds_row(std::true_type{}, t);
        else tone_bounds_row(std::false_type{}, t);
      } else {
        float q[24];
        dispatch([&](auto unit_c, auto ca0_c) { tone_row(unit_c, ca0_c, t, q); });
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          vmin = fminf(vmin, fminf(q[3 * k], fminf(q[3 * k + 1], q[3 * k + 2])));
          vmax = fmaxf(vmax, fmaxf(q[3 * k], fmaxf(q[3 * k + 1], q[3 * k + 2])));
        }
        if constexpr (RR >= NL) {
#pragma unroll
          for (int j = 0; j < 24; ++j) qr_[RR - NL][j] = q[j];
        }
      }
    }
  });
  MI_MSTAMP(5);
  // From here to the first row of the next frame the waves keep the priority of phase C's last turn: the younger block's
  // waves 1, the older one's 0.  Measured per frame: as is 44.4 us; both at 0 (the older wave wins the ties of phase D)
  // 45.18; both at 1: 44.89; the younger one's at 3: 44.50; turns per row in phase D 45.02.
  {
    if (!col_ok) { vmin = __builtin_inff(); vmax = -__builtin_inff(); }
    const float v2[2] = {vmin, vmax};
    const int op[2] = {0, 1};
    block_reduce_post<2>(v2, op, red, &arrived, rows_bounds2, p.part_stride, blockIdx.x, wave, lane, tag);
  }
  MI_MSTAMP(6);
  // Between the post and the first poll: the second Reinhard evaluation of the first LDS rows.  It needs nothing from
  // the barrier, and every wave has ~2 us to wait for the last block's record anyway.
  constexpr int PRE2 = RGB ? 0 : MI_MEGA_PRE2;          // (the color_adapt != 0 kernel has no room for it: 2 spills)
  float qpre[PRE2 > 0 ? PRE2 : 1][24];
  static_for<0, PRE2>([&](auto rrc) { fresh(qpre[decltype(rrc)::value]); });
  dispatch([&](auto unit_c, auto ca0_c) {
    static_for<0, PRE2>([&](auto rrc) {
      constexpr int RR = decltype(rrc)::value;
      if (r_begin + RR < r_end) {
        float t[24];
        resident(rrc, t);
        tone_row(unit_c, ca0_c, t, qpre[RR]);
      }
    });
  });
#ifdef MI_STREAM_STAMPS
  barrier_fold<2, ew::FIN_BOUNDS2>(m, ws, seq, 2, rows_bounds2, tag, sh_fp, fl, lane, true, st_ + 12);
#else
  barrier_fold<2, ew::FIN_BOUNDS2>(m, ws, seq, 2, rows_bounds2, tag, sh_fp, fl, lane);
#endif
  MI_MSTAMP(7);
  const float lo2 = vgpr(sh_fp[FP_LO2]), inv2 = vgpr(sh_fp[FP_INV2]);
  const float out_scale = vgpr(p.out_scale);

  // ================================ phase D: final map (tonemap.py:154) ================================
  uint32_t lane_off[6];
  {
    const int lane_d = lane;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      lane_off[j] = (j * 64 + lane_d) < active_lanes * units_per_lane ? (uint32_t)(j * 64 + lane_d) * unit_bytes : INVALID_OFF;
  }
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(
      io.dst, 0, (int)((uint32_t)p.H * (uint32_t)p.W * 3u * (uint32_t)osz), 0x00020000);
  auto finish_row = [&](auto rrc, float (&q)[24]) {
    constexpr int RR = decltype(rrc)::value;
    linear_n<24>(q, lo2, inv2, p.gamma_inv, out_scale);
    // staging: the LDS slot of a row that has been consumed (its own, or row 0's for the register rows)
    uint4* stage = xl[wave][RR < NL ? RR : 0];
    const uint32_t row_base = (uint32_t)(r_begin + RR) * out_pitch + band_base;
    switch (MI_CENSUS(p.out_dtype, MI_F16)) {
      case MI_U8: wave_store_row_t<uint8_t, ST_STREAM>(drsrc, row_base, lane_off, lane, stage, q); break;
      case MI_U16: wave_store_row_t<uint16_t, ST_STREAM>(drsrc, row_base, lane_off, lane, stage, q); break;
      default: {                                        // f16: pairs leave through v_cvt_pk_f16_f32 (half the conversions)
        uint32_t pk[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[j]) : "v"(q[2 * j]), "v"(q[2 * j + 1]));
        uint4 mine[3];
        __builtin_memcpy(mine, pk, sizeof(mine));
        // Measurement variants (scripts/build_variant.sh; DESIGN.md 5.1d "what the stores cost"): 44.9 us per frame as shipped,
        // 37.5 with MI_MEGA_TEST_NOSTORE, 39.4 with ..._STAGE_ONLY, 44.8 with ..._STORE_ONLY, 54.6 / 96.1 with ..._DIRECT=0 / 2.
        wave_store_units<uint4, 3, MI_MEGA_ST_AUX>(drsrc, row_base, lane_off, lane, stage, mine);   // streamed: nobody reads the output back
        break;
      }
    }
  };
  // Order: the row(s) evaluated before the barrier, then the REGISTER rows - their mapped values are
  // ready, so their 7 x 3 KB of stores per wave are under way within a microsecond and the memory system is busy from
  // the start of the phase -, then Reinhard again for the other LDS rows (their mapped values had no room to stay) while
  // those stores drain.  Order 0 (round 2): LDS rows first, the burst of the register rows' stores at the end.
  constexpr bool REGS_FIRST = PRE2 >= 1;   // (row 0's slot, the register rows' staging, must be free)
  auto lds_rows = [&](auto regs_first_c) {
    constexpr bool REGS_FIRST = decltype(regs_first_c)::value;
    dispatch([&](auto unit_c, auto ca0_c) {
      static_for<PRE2, NL>([&](auto rrc) {
        constexpr int RR = decltype(rrc)::value;
        if (r_begin + RR < r_end) {
          float t[24], q[24];
          resident(rrc, t);
          tone_row(unit_c, ca0_c, t, q);
          finish_row(rrc, q);
        }
        // the next frame's first rows are asked for while the last rows of this one are mapped and stored (unconditional -
        // the last frame asks for its own rows once more: a condition would keep the OLD contents of these 32 registers
        // alive from phase A to here, as the other arm of the merge)
        if constexpr (REGS_FIRST && RR == (MI_MEGA_PREFETCH_AT < NL - 1 ? MI_MEGA_PREFETCH_AT : NL - 1))
          first_loads(G, mb.io[f + 1 < mb.n_frames ? f + 1 : f].src);
      });
    });
  };
  auto reg_rows = [&](auto regs_first_c) {
    constexpr bool REGS_FIRST = decltype(regs_first_c)::value;
    static_for<NL, ROWS>([&](auto rrc) {
      constexpr int RR = decltype(rrc)::value;
      if (r_begin + RR < r_end) {
        float q[24];
#pragma unroll
        for (int j = 0; j < 24; ++j) q[j] = qr_[RR - NL][j];
        finish_row(rrc, q);
      }
      if constexpr (!REGS_FIRST && RR == (MI_MEGA_PREFETCH_AT_LATE < ROWS - 1 ? MI_MEGA_PREFETCH_AT_LATE : ROWS - 1))
        first_loads(G, mb.io[f + 1 < mb.n_frames ? f + 1 : f].src);
    });
  };
  static_for<0, PRE2>([&](auto rrc) {
    constexpr int RR = decltype(rrc)::value;
    if (r_begin + RR < r_end) finish_row(rrc, qpre[RR]);
  });
  // (Measured and taken out: the two blocks of a CU taking phase D in opposite orders - the younger one's waves storing
  // their register rows at once, the older one's mapping their LDS rows first: both orders in one body cost 54 - 73 spills.)
  if constexpr (REGS_FIRST) {
    reg_rows(std::true_type{});
    lds_rows(std::true_type{});
  } else {
    lds_rows(std::false_type{});
    reg_rows(std::false_type{});
  }

  MI_MSTAMP(8);
  // the launch count of the workspace: every block read it at entry and has passed all barriers before block 0 gets here
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned*>(ws)[FP_EPOCH] = epoch + 1u;
  }   // frames
}

// ---- host side ---------------------------------------------------------------------------------
// The whole-frame kernel takes: f16 work dtype, 12-bit standard layout (strm::supported), outputs of 1 or 2 bytes per
// element (the output row is staged in a 3 KB resident-row slot), and a frame whose waves (512 columns x 12 rows each)
// are all resident at once: at most 2 blocks of 4 waves per CU.
static inline bool geometry(int H, int W, int n_cus, SArgs& a) {
  a.bands_x = (W + BAND - 1) / BAND;
  a.rows_per_wave = ROWS;
  const int bands_y = (H + ROWS - 1) / ROWS;
  a.n_waves = a.bands_x * bands_y;
  a.n_blocks = (a.n_waves + WAVES - 1) / WAVES;
  // all blocks resident (2 per CU) and their records inside the 8 partial rows of a barrier (part_stride >= 4096)
  return a.n_blocks <= 2 * n_cus && (size_t)a.n_blocks * REC <= (size_t)8 * 4096 * sizeof(float);
}

int launch_rggb(const MBatch& mb, hipStream_t stream);
int launch_grbg(const MBatch& mb, hipStream_t stream);
int launch_gbrg(const MBatch& mb, hipStream_t stream);
int launch_bggr(const MBatch& mb, hipStream_t stream);
static inline int launch(const MBatch& mb, int pattern, hipStream_t stream) {
  switch (pattern) {
    case MI_RGGB: return launch_rggb(mb, stream);
    case MI_GRBG: return launch_grbg(mb, stream);
    case MI_GBRG: return launch_gbrg(mb, stream);
    default: return launch_bggr(mb, stream);
  }
}
int blocks_per_cu_rggb();
int blocks_per_cu_grbg();
int blocks_per_cu_gbrg();
int blocks_per_cu_bggr();
static inline int blocks_per_cu(int pattern) {
  switch (pattern) {
    case MI_RGGB: return blocks_per_cu_rggb();
    case MI_GRBG: return blocks_per_cu_grbg();
    case MI_GBRG: return blocks_per_cu_gbrg();
    default: return blocks_per_cu_bggr();
  }
}

}  // namespace mega
