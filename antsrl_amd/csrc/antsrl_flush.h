// antsrl_flush.h — index plan of the observation copy-out (LDS staging image -> global rows), shared by
// k_act (antsrl_act.hip), k_perceive (antsrl_perceive.hip) and the host-side bounds test
// (tests/test_flush_plan.py compiles this header with g++ and enumerates every lane of every shape).
//
// A wave has staged `rowp` consecutive output elements (one or two observation rows, contiguous in
// memory) in LDS at stage[mis .. mis + rowp), where `mis` is the misalignment of the destination against
// a 16-byte boundary, in elements: the image is shifted so that 16-byte LDS reads line up with 16-byte
// global stores.  dst_al = dst - mis is 16-byte aligned.  The run leaves as
//   * 16-byte stores over the interior pieces [j_lo, j_hi): the first 128 (float32) / 64 (bfloat16) of
//     them START ON A 128-BYTE LINE when the run is long enough (`head` pieces in front go with the
//     trailing store), so those store instructions cover whole lines;
//   * one element-wide store for the <= 2*(pieces-1) edge elements in front of / behind the interior.
// Lanes with nothing left repeat a store that another lane makes too (same address, same value: the
// kernel's memory operations stay unconditional).  INVARIANT (checked exhaustively on the host): every
// index a lane touches lies inside [mis, mis + rowp) elements, every element is covered, and an
// element is only ever stored with its own staged value — no wave writes a byte outside its own rows.
#pragma once
#include <stdint.h>

#if !defined(__HIPCC__) && !defined(__host__)
#define __host__
#define __device__
#endif

struct FlushPlanF32 {
    uint32_t j1, j2, j3; // float4 indices (relative to dst_al / the staging image)
    uint32_t fe;         // float index of this lane's edge element
};

// float32 rows: 8 <= rowp <= 2 * 368 floats, mis in 0..3, line_phase = ((uintptr_t)dst_al >> 4) & 7
__host__ __device__ inline FlushPlanF32 flush_plan_f32(uint32_t lane, uint32_t mis, uint32_t rowp, uint32_t line_phase)
{
    FlushPlanF32 f;
    const uint32_t j_lo = (mis + 3) >> 2, j_hi = (mis + rowp) >> 2; // interior float4s [j_lo, j_hi)
    const uint32_t head = (j_hi - j_lo >= 72u) ? ((8u - ((line_phase + j_lo) & 7u)) & 7u) : 0u;
    const uint32_t last = j_hi - 1;
    const uint32_t a = j_lo + head + lane, b = j_lo + head + 64u + lane;
    const uint32_t c = lane < head ? j_lo + lane : j_lo + 128u + lane;
    f.j1 = a < last ? a : last;
    f.j2 = b < last ? b : last;
    f.j3 = c < last ? c : last;
    const uint32_t hd = 4 * j_lo - mis, tl = mis + rowp - 4 * j_hi;
    f.fe = lane < hd ? mis + lane : (lane - hd < tl ? 4 * j_hi + (lane - hd) : mis);
    return f;
}

struct FlushPlanB16 {
    uint32_t g1, g2; // 16-byte (8-element) piece indices
    uint32_t fe;     // element index of this lane's edge element
};

// bfloat16 rows: 16 <= rowp <= 2 * 368 elements (so that an interior piece exists), mis in 0..7, line_phase = ((uintptr_t)dst_al >> 4) & 7
__host__ __device__ inline FlushPlanB16 flush_plan_b16(uint32_t lane, uint32_t mis, uint32_t rowp, uint32_t line_phase)
{
    FlushPlanB16 f;
    const uint32_t g_lo = (mis + 7) >> 3, g_hi = (mis + rowp) >> 3; // interior pieces [g_lo, g_hi)
    const uint32_t head = (g_hi - g_lo >= 72u) ? ((8u - ((line_phase + g_lo) & 7u)) & 7u) : 0u;
    const uint32_t last = g_hi - 1;
    const uint32_t a = g_lo + head + lane;
    const uint32_t b = lane < head ? g_lo + lane : g_lo + 64u + lane;
    f.g1 = a < last ? a : last;
    f.g2 = b < last ? b : last;
    const uint32_t hd = 8 * g_lo - mis, tl = mis + rowp - 8 * g_hi;
    f.fe = lane < hd ? mis + lane : (lane - hd < tl ? 8 * g_hi + (lane - hd) : mis);
    return f;
}

// (The whole-line copy-out with a carry — LineFlush / line_flush / line_piece, round 2 — lost to this plan inside the full
// kernel at every stage and left the tree in round 4: profiles/r04/perceive_cleanup.patch.)
