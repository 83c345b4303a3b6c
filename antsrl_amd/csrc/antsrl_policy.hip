// antsrl_policy.hip — in-loop policy inference for BASELINE config 5.
//
// The reference's DQN agents are tiny linear nets with no activation
// (agents/explore_agent_pytorch.py:24-45, agents/collect_agent.py:24-51):
//     out        = layer1(cat[state.view(-1, F), agent_state.view(-1, 2)])        F = P*P*K, 32 hidden
//     rotations  = layer2(out)   (3)        pheromones = layer3(out)   (3)
//     rotation   = argmax(rotations) - rotations//2,  pheromone = argmax(pheromones)
//                  (agents/collect_agent_memory.py:196-199)
// This IS a dense contraction, so it is the one place on the path that uses MFMA: bf16 operands,
// fp32 accumulation (v_mfma_f32_32x32x16_bf16).  One wave owns tiles of 32 ants; the first layer is
// computed TRANSPOSED (A = W1 [32 hidden x k], B = X^T [k x 32 ants]) so that a lane ends up holding
// 16 of the 32 hidden values of ITS ant, and the heads are a second MFMA that takes that accumulator
// as its B operand without any lane movement.  W1 is converted to bf16 once per workgroup into LDS;
// the observation rows stream from HBM exactly once, as 32-byte pieces per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "antsrl_device.h"
#define ANTSRL_MAX_DEVICES 64 // per-device launch bookkeeping (dynamic-LDS opt-in), as in antsrl_util.h

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define POL_HIDDEN 32
#define POL_MAX_KSTEPS 64 // (F + 2) <= 1024 inputs: W1 as bf16 fits 64 KiB of LDS

struct __attribute__((packed, aligned(4))) F4 { float v[4]; }; // 4-byte aligned 16-byte load
struct __attribute__((packed, aligned(4))) D3 { uint32_t v[3]; }; // 4-byte aligned 12-byte load
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t stream4u;
#define POL_KC 64                 // inputs per staged chunk (4 MFMA k-steps)
#define POL_SROW (POL_KC + 8)     // bf16 per staged row: 144-byte stride = 16-byte aligned rows, 4-bank skew per row

// bit shift that aligns the bf16 pair stream of POL_LOAD16: 16 when the element index of (row a, input k) is odd
__device__ __forceinline__ uint32_t pol_shift(int a, int rows, int F, int k)
{
    return (((uint32_t)min(a, rows - 1) * (uint32_t)F + (uint32_t)k) & 1u) * 16u;
}

__device__ __forceinline__ void pol_wave_sync()
{
    // LDS hand-off inside one wave: its LDS instructions execute in order, only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <bool OBS16> // OBS16: `obs` points to bfloat16 rows (antsrl_set_obs_format), same shape
__global__ void __launch_bounds__(256)
k_policy_mlp(const float *__restrict__ obs, const float *__restrict__ agent_state, const float *__restrict__ w1,
             const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
             const float *__restrict__ w3, const float *__restrict__ b3, int8_t *__restrict__ rot_out,
             int8_t *__restrict__ ph_out, float *__restrict__ logits_out, const int M, const int F,
             const int ksteps, const int newest_first)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __bf16 *w1s = reinterpret_cast<__bf16 *>(smem); // [32][KP], KP = 16*ksteps + 8 (8 = bank skew)
    const int KP = 16 * ksteps + 8;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int IN = F + 2;
    // W1 -> bf16 in LDS: a wave takes 8 of the 32 rows; for each 64-column slice its 8 loads are issued
    // together (one round trip per slice, not one per element)
    for (int k = lane; k < 16 * ksteps; k += 64) {
        float wv[POL_HIDDEN / 4];
#pragma unroll
        for (int j = 0; j < POL_HIDDEN / 4; ++j) {
            const int row = (int)(threadIdx.x >> 6) + 4 * j;
            const float wl = w1[(size_t)row * IN + min(k, IN - 1)]; // unconditional (clamped) load, then select
            wv[j] = k < IN ? wl : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < POL_HIDDEN / 4; ++j) w1s[((int)(threadIdx.x >> 6) + 4 * j) * KP + k] = (__bf16)wv[j];
    }
    // Heads as a second MFMA: logits^T [32 (6 used) x 32 ants] = W23 [32 x 32 hidden] . H^T, with the
    // first accumulator reused AS the B operand: its registers 8s..8s+7 (converted to bf16) are the
    // fragment of k-step s, in the permuted k order  k(j, h) = 16 s + 8 (j >> 2) + 4 h + (j & 3)
    // (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"), so W23's fragment
    // is loaded in that same order.  Output row o lands in register o of half-wave 0 (o < 4) or
    // register o - 4 of half-wave 1.
    bf16x8 a2[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int hid = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            // unconditional loads on clamped rows + selects: a load inside a per-lane branch gets its own
            // s_waitcnt vmcnt(0), i.e. 16 serial round trips in every workgroup's prologue
            const float wa = w2[min(r, 2) * POL_HIDDEN + hid];
            const float wb = w3 ? w3[min(max(r - 3, 0), 2) * POL_HIDDEN + hid] : 0.0f; // (w3 == NULL is uniform)
            a2[s][j] = (__bf16)(r < 3 ? wa : (r < 6 ? wb : 0.0f));
        }
    float bias1[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) bias1[g] = b1[(g & 3) + 8 * (g >> 2) + 4 * h];
    float hb[6];
#pragma unroll
    for (int o = 0; o < 6; ++o) hb[o] = o < 3 ? b2[o] : (w3 ? b3[o - 3] : 0.0f);
    __syncthreads();

    // The 32 rows of a tile are CONTIGUOUS in `obs` (32 * F floats).  They are streamed in chunks of
    // POL_KC inputs: a lane group of 16 reads 256 contiguous bytes of one ant's row, 4 ants per load
    // instruction (coalesced; a lane-per-ant read touches 32+ cache lines per instruction), converted to
    // bf16 and transposed through a wave-private LDS tile into the MFMA's B-fragment order (lane (r, h)
    // of k-step s holds inputs 16 s + 8 h .. + 7 of ant r).  Chunk c + 1 is in flight while chunk c is
    // converted and multiplied.
    const int ntiles = (M + 31) / 32;
    const int wib = threadIdx.x >> 6;                                  // wave in block
    __bf16 *stg = w1s + POL_HIDDEN * KP + (size_t)wib * 32 * POL_SROW; // [32][POL_SROW]
    const int la = lane >> 4, lf = lane & 15;                          // loader role: ant la + 4 i, floats 4 lf .. 4 lf + 3
    const int nchunks = (IN + POL_KC - 1) / POL_KC;
    for (int t0 = wave; t0 < ntiles; t0 += nwaves) {
        const int t = newest_first ? ntiles - 1 - t0 : t0; // (A/B switch, see antsrl_launch_policy)
        const int ant = min(t * 32 + r, M - 1); // clamped: duplicates are not written back
        const int rows = min(32, M - t * 32);
        const float *tile = obs + (size_t)t * 32 * F;
        const uint16_t *tile16 = reinterpret_cast<const uint16_t *>(obs) + (size_t)t * 32 * F;
        const __bf16 *wrow = w1s + r * KP + 8 * h;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
#define POL_LOAD(V, CH)                                                                                  \
    {                                                                                                    \
        const int k_ = POL_KC * (CH) + 4 * lf;                                                           \
        if (POL_KC * ((CH) + 1) <= F) { /* every input of the chunk lies inside the observation rows */  \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                \
            {                                                                                            \
                const int a = min(la + 4 * i, rows - 1);                                                 \
                    V[i] = *reinterpret_cast<const F4 *>(tile + (size_t)a * F + k_);                     \
            }                                                                                            \
        } else if ((CH) < nchunks) { /* end of the row, the two agent_state inputs, zero pad.  Every load */ \
            /* is UNCONDITIONAL on a clamped address and the choice is a select: with the loads inside    */ \
            /* branches each one was followed by its own s_waitcnt vmcnt(0) - 48 serial round trips/tile */ \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                \
            {                                                                                            \
                const int a = min(la + 4 * i, rows - 1);                                                 \
                const float as0_ = agent_state[((size_t)t * 32 + a) * 2], as1_ = agent_state[((size_t)t * 32 + a) * 2 + 1]; \
                float x_[4];                                                                             \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) x_[j] = tile[(size_t)a * F + min(k_ + j, F - 1)]; \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                            \
                {                                                                                        \
                    const int kk = k_ + j;                                                               \
                    V[i].v[j] = kk < F ? x_[j] : (kk == F ? as0_ : (kk == F + 1 ? as1_ : 0.0f));         \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
    }
        // bfloat16 rows: 4 elements per lane and ant stay PACKED in two registers from the load to the LDS
        // tile (three ALIGNED dwords of the tile — its base is 64-byte aligned — and a funnel shift stand in
        // for a 2-byte aligned 8-byte load), which leaves room for two chunks in flight.
#define POL_LOAD16_FULL(V, CH) /* chunk CH lies inside the rows (+ 4 bytes of read-ahead) */          \
    {                                                                                                    \
        const int k_ = POL_KC * (CH) + 4 * lf;                                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                    \
        {                                                                                                \
            const uint32_t ei = (uint32_t)min(la + 4 * i, rows - 1) * (uint32_t)F + (uint32_t)k_;        \
            /* RAW dwords: the funnel shift waits until the chunk is consumed (POL_PACK16_FULL), or the */ \
            /* load would be waited for right here and nothing would be in flight                      */ \
            V[i] = *reinterpret_cast<const D3 *>(reinterpret_cast<const uint32_t *>(tile16) + (ei >> 1)); \
        }                                                                                                \
    }
#define POL_LOAD16_TAIL(V, CH) /* end of the row, the two agent_state inputs, zero pad: unconditional */ \
    {                          /* loads on clamped addresses + selects (see POL_LOAD), packed here     */ \
        const int k_ = POL_KC * (CH) + 4 * lf;                                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                    \
        {                                                                                                \
            const int a = min(la + 4 * i, rows - 1);                                                     \
            const uint32_t as0_ = __builtin_bit_cast(uint16_t, (__bf16)agent_state[((size_t)t * 32 + a) * 2]);     \
            const uint32_t as1_ = __builtin_bit_cast(uint16_t, (__bf16)agent_state[((size_t)t * 32 + a) * 2 + 1]); \
            uint32_t x_[4], e_[4];                                                                       \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) x_[j] = tile16[(size_t)a * F + min(k_ + j, F - 1)]; \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                \
            {                                                                                            \
                const int kk = k_ + j;                                                                   \
                e_[j] = kk < F ? x_[j] : (kk == F ? as0_ : (kk == F + 1 ? as1_ : 0u));                   \
            }                                                                                            \
            V[i].v[0] = e_[0] | (e_[1] << 16);                                                           \
            V[i].v[1] = e_[2] | (e_[3] << 16);                                                           \
            V[i].v[2] = 0u;                                                                              \
        }                                                                                                \
    }
        // the 4 packed elements of lane (la + 4 i, lf) of chunk CH, from the registers the loaders filled
#define POL_PACK16_FULL(V, CH, I)                                                                        \
    make_uint2(__builtin_amdgcn_alignbit(V[I].v[1], V[I].v[0], pol_shift(la + 4 * (I), rows, F, POL_KC * (CH) + 4 * lf)), \
               __builtin_amdgcn_alignbit(V[I].v[2], V[I].v[1], pol_shift(la + 4 * (I), rows, F, POL_KC * (CH) + 4 * lf)))
#define POL_CONSUME16(CH, FULL)                                                                          \
    {                                                                                                    \
        pol_wave_sync(); /* the previous chunk's fragment reads are done */                              \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                    \
            *reinterpret_cast<uint2 *>(stg + (la + 4 * i) * POL_SROW + 4 * lf) =                         \
                (FULL) ? POL_PACK16_FULL(q0, CH, i) : make_uint2(q0[i].v[0], q0[i].v[1]);                \
        pol_wave_sync();                                                                                 \
        const int s_end = min(ksteps - (POL_KC / 16) * (CH), POL_KC / 16);                               \
        for (int s = 0; s < s_end; ++s) {                                                                \
            const bf16x8 bfrag = *reinterpret_cast<const bf16x8 *>(stg + r * POL_SROW + 16 * s + 8 * h); \
            const bf16x8 afrag = *reinterpret_cast<const bf16x8 *>(wrow + 16 * ((POL_KC / 16) * (CH) + s)); \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc, 0, 0, 0);                   \
        }                                                                                                \
    }
        // register rotation BEFORE the next chunk is requested: copying q2 waits for the load issued one
        // whole iteration ago, not for the one just issued
#define POL_ROTATE16                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                        \
    {                                                                                                    \
        q0[i] = q1[i];                                                                                   \
        q1[i] = q2[i];                                                                                   \
    }
        if constexpr (OBS16) {
            // chunks [0, nfull) are whole; the steady-state loop below is BRANCH-FREE around its loads, so the
            // compiler's vmcnt bookkeeping is exact and two chunks really are in flight (with the load
            // kind chosen by an if / else inside the loop it falls back to waiting for nearly everything)
            const int nfull = min(nchunks, max(0, (F - 2) / POL_KC));
            D3 q0[8], q1[8], q2[8];
            if (nfull > 0) POL_LOAD16_FULL(q0, 0) else POL_LOAD16_TAIL(q0, 0)
            if (nfull > 1) POL_LOAD16_FULL(q1, 1) else if (nchunks > 1) POL_LOAD16_TAIL(q1, 1)
            int c = 0;
            for (; c + 2 < nfull; ++c) {
                if (c > 0) { POL_ROTATE16 }
                POL_LOAD16_FULL(q2, c + 2)
                POL_CONSUME16(c, true)
            }
            for (; c < nchunks; ++c) { // drain: at most the last two whole chunks and the partial ones
                if (c > 0) { POL_ROTATE16 }
                if (c + 2 < nchunks) POL_LOAD16_TAIL(q2, c + 2) // (c + 2 >= nfull here)
                if (c < nfull) POL_CONSUME16(c, true) else POL_CONSUME16(c, false)
            }
        } else {
        F4 v[8], nx[8];
        POL_LOAD(v, 0)
        for (int c = 0; c < nchunks; ++c) {
            POL_LOAD(nx, c + 1)
            pol_wave_sync(); // the previous chunk's fragment reads are done
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = (__bf16)v[i].v[j];
                *reinterpret_cast<bf16x4 *>(stg + (la + 4 * i) * POL_SROW + 4 * lf) = pk;
            }
            pol_wave_sync();
            const int s_end = min(ksteps - (POL_KC / 16) * c, POL_KC / 16);
            for (int s = 0; s < s_end; ++s) {
                const bf16x8 bfrag = *reinterpret_cast<const bf16x8 *>(stg + r * POL_SROW + 16 * s + 8 * h);
                const bf16x8 afrag = *reinterpret_cast<const bf16x8 *>(wrow + 16 * ((POL_KC / 16) * c + s));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = nx[i];
        }
        }
#undef POL_LOAD
#undef POL_LOAD16_FULL
#undef POL_LOAD16_TAIL
#undef POL_PACK16_FULL
#undef POL_CONSUME16
#undef POL_ROTATE16
        // acc[g] = hidden[(g&3) + 8*(g>>2) + 4*h] of ant r (before bias)
        f32x16 acc2;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc2[g] = 0.0f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 hfrag;
#pragma unroll
            for (int j = 0; j < 8; ++j) hfrag[j] = (__bf16)(acc[8 * s + j] + bias1[8 * s + j]);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s], hfrag, acc2, 0, 0, 0);
        }
        // logits of ant r: rows 0..3 in this lane if h == 0, rows 4,5 in lane r + 32
        float lg[6];
        lg[0] = acc2[0]; lg[1] = acc2[1]; lg[2] = acc2[2]; lg[3] = acc2[3];
        lg[4] = __shfl(acc2[0], r + 32); lg[5] = __shfl(acc2[1], r + 32);
#pragma unroll
        for (int o = 0; o < 6; ++o) lg[o] += hb[o];
        if (h == 0 && t * 32 + r < M) {
            // torch.max(...).indices: first maximum wins
            int ar = 0, ap = 0;
            if (lg[1] > lg[ar]) ar = 1;
            if (lg[2] > lg[ar]) ar = 2;
            if (lg[4] > lg[3 + ap]) ap = 1;
            if (lg[5] > lg[3 + ap]) ap = 2;
            rot_out[ant] = (int8_t)(ar - 1); // - rotations // 2
            if (ph_out) ph_out[ant] = (int8_t)ap;
            if (logits_out)
#pragma unroll
                for (int o = 0; o < 6; ++o) logits_out[(size_t)ant * 6 + o] = lg[o];
        }
    }
}

// ===================================================================================
// k_policy_flat — the same network, with each 32-ant tile of the observation tensor streamed as ONE flat,
// aligned, fully coalesced run (a tile's rows are contiguous: 32 * F elements) into a wave-private LDS
// image in bf16; the MFMA B fragments (lane (r, h), k-step s = inputs 16 s + 8 h .. + 7 of ant r) are then
// read from that image at whatever 2-byte alignment r * F gives them: five aligned dwords and a funnel
// shift.  20 load instructions per tile instead of ~90 row-strided ones.  W1's columns for k >= F are
// zero in LDS, so whatever follows a row in the image (the next row, the zeroed pad) contributes
// nothing; the two agent_state inputs are a rank-2 update in registers on bf16-rounded operands.
// Used whenever W1 + two tile images fit in LDS (any F the reference's perception can produce).
// ===================================================================================
template <bool OBS16>
__global__ void __launch_bounds__(128)
k_policy_flat(const float *__restrict__ obs, const float *__restrict__ agent_state, const float *__restrict__ w1,
              const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
              const float *__restrict__ w3, const float *__restrict__ b3, int8_t *__restrict__ rot_out,
              int8_t *__restrict__ ph_out, float *__restrict__ logits_out, const int M, const int F,
              const int ksteps /* ceil(F / 16) */, const int tile_elems /* LDS image size per wave, bf16 */,
              const int newest_first)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __bf16 *w1s = reinterpret_cast<__bf16 *>(smem); // [32][KP], KP = 16*ksteps + 8 (8 = bank skew)
    const int KP = 16 * ksteps + 8;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wib = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int IN = F + 2;
    uint16_t *img = reinterpret_cast<uint16_t *>(w1s + POL_HIDDEN * KP) + (size_t)wib * tile_elems;
    for (int k = lane; k < 16 * ksteps; k += 64) { // W1 -> bf16, observation columns only (a wave takes 16 rows)
        float wv[POL_HIDDEN / 2];
#pragma unroll
        for (int j = 0; j < POL_HIDDEN / 2; ++j) wv[j] = w1[(size_t)(wib + 2 * j) * IN + min(k, F - 1)];
#pragma unroll
        for (int j = 0; j < POL_HIDDEN / 2; ++j) w1s[(wib + 2 * j) * KP + k] = (__bf16)(k < F ? wv[j] : 0.0f);
    }
    for (int i = lane; i < tile_elems / 8; i += 64) reinterpret_cast<uint4 *>(img)[i] = make_uint4(0, 0, 0, 0);
    bf16x8 a2[2]; // heads, see k_policy_mlp
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int hid = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            const float wa = w2[min(r, 2) * POL_HIDDEN + hid];
            const float wb = w3 ? w3[min(max(r - 3, 0), 2) * POL_HIDDEN + hid] : 0.0f;
            a2[s][j] = (__bf16)(r < 3 ? wa : (r < 6 ? wb : 0.0f));
        }
    float bias1[16], was0[16], was1[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int hid = (g & 3) + 8 * (g >> 2) + 4 * h;
        bias1[g] = b1[hid];
        was0[g] = (float)(__bf16)w1[(size_t)hid * IN + F];
        was1[g] = (float)(__bf16)w1[(size_t)hid * IN + F + 1];
    }
    float hb[6];
#pragma unroll
    for (int o = 0; o < 6; ++o) hb[o] = o < 3 ? b2[o] : (w3 ? b3[o - 3] : 0.0f);
    __syncthreads();

    const int ntiles = (M + 31) / 32;
    const __bf16 *wrow = w1s + r * KP + 8 * h;
    for (int t0 = wave; t0 < ntiles; t0 += nwaves) {
        const int t = newest_first ? ntiles - 1 - t0 : t0; // (A/B switch, see antsrl_launch_policy)
        const int ant = min(t * 32 + r, M - 1); // clamped: duplicates are not written back
        const int rows = min(32, M - t * 32);
        const uint32_t nelem = (uint32_t)rows * (uint32_t)F;
        const float as0 = (float)(__bf16)agent_state[(size_t)ant * 2], as1 = (float)(__bf16)agent_state[(size_t)ant * 2 + 1];
        pol_wave_sync(); // the previous tile's fragment reads are done
        // ---- producer: the tile as a flat run of 16-byte pieces, 8 loads in flight per batch
        if constexpr (OBS16) {
            const uint16_t *src = reinterpret_cast<const uint16_t *>(obs) + (size_t)t * 32 * F; // 64-byte aligned
            const uint32_t n16 = nelem >> 3; // whole 8-element pieces; the < 8 elements left over go one by one
            for (uint32_t b0 = 0; b0 < n16; b0 += 8 * 64) {
                uint4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const uint4 *>(src)[min(b0 + 64u * j + lane, n16 - 1)];
#pragma unroll
                for (int j = 0; j < 8; ++j) reinterpret_cast<uint4 *>(img)[min(b0 + 64u * j + lane, n16 - 1)] = v[j];
            }
            const uint32_t rem = nelem & 7u, e0 = n16 << 3;
            const uint16_t tail = src[min(e0 + (uint32_t)lane, nelem - 1)]; // unconditional, clamped
            if ((uint32_t)lane < rem) img[e0 + lane] = tail;
        } else {
            const float *src = obs + (size_t)t * 32 * F; // 128-byte aligned
            const uint32_t n4 = nelem >> 2;
            for (uint32_t b0 = 0; b0 < n4; b0 += 8 * 64) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const float4 *>(src)[min(b0 + 64u * j + lane, n4 - 1)];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bf16x4 pk;
                    pk[0] = (__bf16)v[j].x; pk[1] = (__bf16)v[j].y; pk[2] = (__bf16)v[j].z; pk[3] = (__bf16)v[j].w;
                    reinterpret_cast<bf16x4 *>(img)[min(b0 + 64u * j + lane, n4 - 1)] = pk;
                }
            }
            const uint32_t rem = nelem & 3u, e0 = n4 << 2;
            const float tail = src[min(e0 + (uint32_t)lane, nelem - 1)];
            if ((uint32_t)lane < rem) img[e0 + lane] = __builtin_bit_cast(uint16_t, (__bf16)tail);
        }
        pol_wave_sync();
        // ---- consumer
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
        const uint32_t off0 = (uint32_t)r * (uint32_t)F + 8u * h;
        for (int s = 0; s < ksteps; ++s) {
            const uint32_t off = off0 + 16u * s, sh = (off & 1u) * 16u;
            const uint32_t *d = reinterpret_cast<const uint32_t *>(img) + (off >> 1);
            const uint32_t d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
            const stream4u packed = {__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh),
                                     __builtin_amdgcn_alignbit(d3, d2, sh), __builtin_amdgcn_alignbit(d4, d3, sh)};
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, packed);
            const bf16x8 afrag = *reinterpret_cast<const bf16x8 *>(wrow + 16 * s);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc, 0, 0, 0);
        }
        f32x16 acc2;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc2[g] = 0.0f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 hfrag;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                hfrag[j] = (__bf16)(acc[8 * s + j] + (as0 * was0[8 * s + j] + as1 * was1[8 * s + j]) + bias1[8 * s + j]);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s], hfrag, acc2, 0, 0, 0);
        }
        float lg[6];
        lg[0] = acc2[0]; lg[1] = acc2[1]; lg[2] = acc2[2]; lg[3] = acc2[3];
        lg[4] = __shfl(acc2[0], r + 32); lg[5] = __shfl(acc2[1], r + 32);
#pragma unroll
        for (int o = 0; o < 6; ++o) lg[o] += hb[o];
        if (h == 0 && t * 32 + r < M) {
            int ar = 0, ap = 0; // torch.max(...).indices: first maximum wins
            if (lg[1] > lg[ar]) ar = 1;
            if (lg[2] > lg[ar]) ar = 2;
            if (lg[4] > lg[3 + ap]) ap = 1;
            if (lg[5] > lg[3 + ap]) ap = 2;
            rot_out[ant] = (int8_t)(ar - 1);
            if (ph_out) ph_out[ant] = (int8_t)ap;
            if (logits_out)
#pragma unroll
                for (int o = 0; o < 6; ++o) logits_out[(size_t)ant * 6 + o] = lg[o];
        }
    }
}

// ===================================================================================
// k_policy_pack — the weights as k_perceive's in-loop policy takes them (policy_tile, antsrl_perceive.hip): W1's
// observation columns as bf16 MFMA A fragments per (k-step, lane), the heads' A fragments, and the per-lane fp32 terms
// (bias1, the bf16-rounded weights of the two agent_state inputs, the head biases) — exactly what k_policy_flat's
// prologue builds per workgroup, built once per antsrl_set_inloop_policy.
// ===================================================================================
__global__ void __launch_bounds__(64)
k_policy_pack(unsigned char *__restrict__ pack, const float *__restrict__ w1, const float *__restrict__ b1,
              const float *__restrict__ w2, const float *__restrict__ b2, const float *__restrict__ w3,
              const float *__restrict__ b3, const int F, const int ksteps)
{
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int IN = F + 2;
    bf16x8 *wpack = reinterpret_cast<bf16x8 *>(pack);
    bf16x8 *a2pack = reinterpret_cast<bf16x8 *>(pack + ANTSRL_POL_WPACK_BYTES);
    float *lanepack = reinterpret_cast<float *>(pack + ANTSRL_POL_WPACK_BYTES + ANTSRL_POL_A2PACK_BYTES);
    for (int s = 0; s < ksteps; ++s) {
        bf16x8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * h + j;
            a[j] = (__bf16)(k < F ? w1[(size_t)r * IN + k] : 0.0f);
        }
        wpack[s * 64 + lane] = a;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bf16x8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int hid = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            const float wa = w2[min(r, 2) * POL_HIDDEN + hid];
            const float wb = w3 ? w3[min(max(r - 3, 0), 2) * POL_HIDDEN + hid] : 0.0f;
            a[j] = (__bf16)(r < 3 ? wa : (r < 6 ? wb : 0.0f));
        }
        a2pack[s * 64 + lane] = a;
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int hid = (g & 3) + 8 * (g >> 2) + 4 * h;
        lanepack[g * 64 + lane] = b1[hid];
        lanepack[(16 + g) * 64 + lane] = (float)(__bf16)w1[(size_t)hid * IN + F];
        lanepack[(32 + g) * 64 + lane] = (float)(__bf16)w1[(size_t)hid * IN + F + 1];
    }
    if (lane < 6) lanepack[48 * 64 + lane] = lane < 3 ? b2[lane] : (w3 ? b3[lane - 3] : 0.0f);
}

hipError_t antsrl_launch_policy_pack(unsigned char *pack, const float *w1, const float *b1, const float *w2, const float *b2,
                                     const float *w3, const float *b3, int F, hipStream_t st)
{
    const int ks = (F + 15) / 16;
    if (ks > ANTSRL_POL_MAX_KSTEPS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_policy_pack, dim3(1), dim3(64), 0, st, pack, w1, b1, w2, b2, w3, b3, F, ks);
    return hipGetLastError();
}

hipError_t antsrl_launch_policy(const float *obs, const float *agent_state, const float *w1, const float *b1,
                                const float *w2, const float *b2, const float *w3, const float *b3, int8_t *rot,
                                int8_t *ph, float *logits, int M, int F, hipStream_t st, bool obs_bf16)
{
    // tile order (profiling switch): newest rows first was tried for Infinity Cache hits on what k_perceive has just
    // written and measured no better (c5 0.1301 vs 0.1294 ms/step, profiles/r02/policy_order_ab.txt)
    const int newest_first = PROF_ENV("ANTSRL_POLICY_NEWEST_FIRST") ? 1 : 0;

    if (M < 1 || F < 1) return hipErrorInvalidValue;
    {
        // flat-stream form: W1 (observation columns) + two wave-private tile images (32 F elements + the
        // read-ahead of the last k-step, zero padded) in LDS
        static const bool off = PROF_ENV("ANTSRL_POLICY_CHUNKED") != nullptr; // A/B: the chunked kernel
        const int ks = (F + 15) / 16, tile_elems = (32 * F + 32 + 7) / 8 * 8;
        const size_t lds = (size_t)POL_HIDDEN * (16 * ks + 8) * 2 + 2 * (size_t)tile_elems * 2;
        if (!off && ks <= POL_MAX_KSTEPS && lds <= 160 * 1024) {
            const int ntiles = (M + 31) / 32;
            int per_cu = (int)((160 * 1024) / lds);
            per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
            int blocks = (ntiles + 1) / 2;
            if (blocks > 256 * per_cu) blocks = 256 * per_cu;
            hipError_t e = hipSuccess;
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ANTSRL_MAX_DEVICES) return hipErrorInvalidDevice;
            if (obs_bf16) {
                static size_t attr[ANTSRL_MAX_DEVICES] = {}; // per kernel function and per device
                if (lds > attr[dev]) { e = hipFuncSetAttribute((const void *)k_policy_flat<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr[dev] = lds; }
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k_policy_flat<true>, dim3(blocks), dim3(128), lds, st, obs, agent_state, w1, b1, w2, b2, w3,
                                   b3, rot, ph, logits, M, F, ks, tile_elems, newest_first);
            } else {
                static size_t attr[ANTSRL_MAX_DEVICES] = {}; // per kernel function and per device
                if (lds > attr[dev]) { e = hipFuncSetAttribute((const void *)k_policy_flat<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr[dev] = lds; }
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k_policy_flat<false>, dim3(blocks), dim3(128), lds, st, obs, agent_state, w1, b1, w2, b2, w3,
                                   b3, rot, ph, logits, M, F, ks, tile_elems, newest_first);
            }
            return hipGetLastError();
        }
    }
    const int ksteps = (F + 2 + 15) / 16;
    if (ksteps > POL_MAX_KSTEPS) return hipErrorInvalidValue;
    const int ntiles = (M + 31) / 32;
    int blocks = (ntiles + 3) / 4;
    const size_t lds = (size_t)POL_HIDDEN * (16 * ksteps + 8) * 2 + 4 * 32 * (size_t)POL_SROW * 2; // W1 + 4 wave tiles
    int per_cu = (int)((160 * 1024) / lds); // resident workgroups per CU by LDS (38-41 KiB each for the reference's sizes)
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    if (blocks > 256 * per_cu) blocks = 256 * per_cu; // tiles are looped: one round of workgroups
    // more than 64 KiB of dynamic LDS is an opt-in per kernel function and per device (like k_policy_flat, k_act)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ANTSRL_MAX_DEVICES) return hipErrorInvalidDevice;
    if (obs_bf16) {
        static size_t attr[ANTSRL_MAX_DEVICES] = {};
        if (lds > attr[dev]) {
            hipError_t e = hipFuncSetAttribute((const void *)k_policy_mlp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr[dev] = lds;
        }
        hipLaunchKernelGGL(k_policy_mlp<true>, dim3(blocks), dim3(256), lds, st, obs, agent_state, w1, b1, w2, b2, w3, b3,
                           rot, ph, logits, M, F, ksteps, newest_first);
    } else {
        static size_t attr[ANTSRL_MAX_DEVICES] = {};
        if (lds > attr[dev]) {
            hipError_t e = hipFuncSetAttribute((const void *)k_policy_mlp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr[dev] = lds;
        }
        hipLaunchKernelGGL(k_policy_mlp<false>, dim3(blocks), dim3(256), lds, st, obs, agent_state, w1, b1, w2, b2, w3, b3,
                           rot, ph, logits, M, F, ksteps, newest_first);
    }
    return hipGetLastError();
}
