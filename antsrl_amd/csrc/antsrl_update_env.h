// antsrl_update_env.h — the update phases of ONE environment as a device function: used by k_update
// (antsrl_update.hip) and, fused, at the tail of k_act (antsrl_act.hip).
#pragma once
#include "antsrl_util.h"

// ===================================================================================
// k_update — Environment.update minus the pheromone sweep (environment.py:42-47):
// Walls (walls.py:22-28), CircleObstacles (circle_obstacles.py:32-58), Ants.update
// (ants.py:123-130), Anthill.update (anthill.py:41-46, sparse form).
// one workgroup per environment; `out_buf` = pheromone buffer the sweep just wrote.
// ===================================================================================
__device__ __forceinline__ uint32_t block_excl_scan_flag(bool flag, uint32_t *wave_tot, int lane, int wave,
                                                         int nwaves, uint32_t *block_total)
{
    const unsigned long long m = __ballot(flag);
    const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = 0, tot = 0;
    for (int w = 0; w < nwaves; ++w) {
        const uint32_t t = wave_tot[w];
        if (w < wave) off += t;
        tot += t;
    }
    __syncthreads();
    *block_total = tot;
    return off + in_wave;
}

// The update phases of ONE environment, run by the whole workgroup.  `smem` is
// update_scratch_bytes() of LDS.  Called by k_update and, fused, at the tail of k_act.
// `phases` (UPD_*): which of the reference's update steps this call runs — Walls.update (step -1), CircleObstacles.update
// (0), Ants.update (999), Anthill.update (1000); all of them = one Environment.update.  antsrl_update_phase
// (include/antsrl.h) enqueues them one launch at a time so that HOST EnvObjects of the caller can run between them in
// update_step() order (environment.py:43-47); the pheromone's own step-0 update is the host's sweep launch / scaled-unit
// bookkeeping.  The environment's timestep advances with the LAST phase (the jitter of the first one is keyed on
// timestep + 1 either way).
#define UPD_WALLS 1
#define UPD_ROCKS 2
#define UPD_ANTS 4
#define UPD_ANTHILL 8
#define UPD_ALL 15
template <int C>
__device__ __forceinline__ void update_env(const KP &p, const int e, const double *__restrict__ wall_jitter,
                                           const int out_buf, unsigned char *smem, const int phases = UPD_ALL)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
    const int N = p.N, W = p.W, H = p.H, R = p.R;
    const size_t G = (size_t)W * H, eN = (size_t)e * N;
    uint32_t *hkeys = (uint32_t *)smem, *hvals = hkeys + p.HT;
    double *rk = (double *)(smem + align_up(8 * (size_t)p.HT, 16)); // [2R] new rock centres
    double *red = rk + 2 * (R > 0 ? R : 1);                          // [nwaves] reduction scratch
    uint32_t *wave_tot = (uint32_t *)(red + nwaves);                 // [nwaves]

    const uint32_t *walls = p.s.walls_bits + (size_t)e * p.words;
    const FoodView food{p.s.food + (size_t)e * G * p.fs, p.fs};
    const size_t PS = (size_t)p.ps; // floats per cell of the pheromone array
    float *out = p.s.phero[out_buf] + (size_t)e * G * PS;
    const int ts = p.s.timestep[e] + 1; // environment.py:45

    for (int h = tid; h < p.HT; h += T) {
        hkeys[h] = HASH_EMPTY;
        hvals[h] = 0u;
    }

    // ---- Walls.update, walls.py:25-28
    uint32_t carry = 0;
    if (p.scaled && (phases & UPD_WALLS)) {
        // A deposit that landed on a WALL cell in the previous update is visible to exactly one observation and is zeroed by
        // this update's Walls pass (walls.py:30), before this update's deposits (several barriers further down).
        for (int i = tid; i < N; i += T) {
            const int32_t wc = p.s.walldep_cell[eN + i];
            if (wc >= 0) {
                for (int c = 0; c < C; ++c) out[(size_t)wc * PS + c] = 0.0f;
                p.s.walldep_cell[eN + i] = -1;
            }
        }
    }
    for (int base = 0; (phases & UPD_WALLS) && base < N; base += T) {
        const int i = base + tid;
        bool hit = false;
        if (i < N) hit = test_bit(walls, (uint32_t)((int)p.s.x[eN + i] * H + (int)p.s.y[eN + i]));
        double u = 0.0;
        if (wall_jitter) { // k-th colliding ant (index order) takes the k-th draw
            uint32_t tot;
            const uint32_t rank = carry + block_excl_scan_flag(hit, wave_tot, lane, wave, nwaves, &tot);
            carry += tot;
            if (hit) u = wall_jitter[eN + rank];
        } else if (hit) {
            u = jitter_u01(p.rng_seed, p.env_id_base + (uint32_t)e, (uint32_t)ts, (uint32_t)i); // (the env's GLOBAL id)
        }
        if (hit) {
            p.s.x[eN + i] = p.s.prev_x[eN + i];
            p.s.y[eN + i] = p.s.prev_y[eN + i];
            p.s.theta[eN + i] += u - 0.5; // theta is NOT re-wrapped here
        }
    }
    __syncthreads();

    // ---- CircleObstacles.update, circle_obstacles.py:35-58
    if (R > 0 && (phases & UPD_ROCKS)) {
        // pass 1: centres -= sum_over_ants(push)/weight.  numpy sums the ants sequentially in
        // index order; non-colliding ants contribute exact zeros, so adding only the
        // colliding ones in index order reproduces the float64 result bit for bit.
        for (int q = wave; q < R; q += nwaves) {
            const double cx = p.s.rock_cx[(size_t)e * R + q], cy = p.s.rock_cy[(size_t)e * R + q];
            const double rad = p.s.rock_r[(size_t)e * R + q];
            const double rad2_hi = rad * rad * (1.0 + 1e-12) + 1e-300; // d2 above this: sqrt(d2) > rad for sure
            double sx = 0.0, sy = 0.0;
            for (int base = 0; base < N; base += 64) {
                const int i = base + lane;
                double px = 0.0, py = 0.0;
                bool col = false;
                if (i < N) {
                    const double vx = cx - p.s.x[eN + i], vy = cy - p.s.y[eN + i];
                    const double d2 = vx * vx + vy * vy;
                    // only colliding ants contribute: the sqrt and the division are spent on the few
                    // lanes a conservative squared-distance test lets through (the exact test
                    // `!(d > rad)` of the reference then decides)
                    if (!(d2 > rad2_hi)) {
                        const double d = sqrt(d2);
                        const double f = 1 - rad / (d + 0.001);
                        px = vx * f; py = vy * f;
                        col = !(d > rad);
                    }
                }
                unsigned long long m = __ballot(col);
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    sx += __shfl(px, l);
                    sy += __shfl(py, l);
                }
            }
            if (lane == 0) {
                const double wgt = p.s.rock_w[(size_t)e * R + q];
                rk[2 * q + 0] = cx - sx / wgt;
                rk[2 * q + 1] = cy - sy / wgt;
            }
        }
        __syncthreads();
        for (int q = tid; q < R; q += T) {
            p.s.rock_cx[(size_t)e * R + q] = rk[2 * q + 0];
            p.s.rock_cy[(size_t)e * R + q] = rk[2 * q + 1];
        }
        // pass 2 (:53-58): ants pushed out of the UPDATED rocks, then warp_xy
        for (int i = tid; i < N; i += T) {
            const double x = p.s.x[eN + i], y = p.s.y[eN + i];
            double sx = 0.0, sy = 0.0;
            for (int q = 0; q < R; ++q) {
                const double vx = rk[2 * q + 0] - x, vy = rk[2 * q + 1] - y;
                const double rad = p.s.rock_r[(size_t)e * R + q];
                const double d2 = vx * vx + vy * vy;
                if (d2 > rad * rad * (1.0 + 1e-12) + 1e-300) continue; // adds an exact +0.0: skip the sqrt and division
                const double d = sqrt(d2);
                const double f = 1 - rad / (d + 0.001);
                double px = vx * f, py = vy * f;
                if (d > rad) { px = 0.0; py = 0.0; }
                sx += px; sy += py;
            }
            p.s.x[eN + i] = warp_coord(x + sx, (double)W);
            p.s.y[eN + i] = warp_coord(y + sy, (double)H);
        }
        __syncthreads();
    }

    // ---- Ants.update, ants.py:123-130: prev := cur; deposit (pheromone.py:36-41)
    for (int i = tid; (phases & UPD_ANTS) && i < N; i += T) {
        const double x = p.s.x[eN + i], y = p.s.y[eN + i];
        p.s.prev_x[eN + i] = x;
        p.s.prev_y[eN + i] = y;
        lww_insert(hkeys, hvals, (uint32_t)p.HT - 1, prec_xy(p, (int)x, (int)y), (uint32_t)i); // key: the cell's pheromone record index
        p.s.reward_state[eN + i] = (uint8_t)((double)p.s.reward_state[eN + i] * 0.9); // :130
    }
    __syncthreads();
    for (int i = tid; (phases & UPD_ANTS) && i < N; i += T) {
        const uint32_t cell_id = (uint32_t)((int)p.s.x[eN + i] * H + (int)p.s.y[eN + i]); // row-major id: the wall bit map
        const uint32_t cell = prec_xy(p, (int)p.s.x[eN + i], (int)p.s.y[eN + i]);           // the cell's pheromone record
        if (lww_winner(hkeys, hvals, (uint32_t)p.HT - 1, cell) == (uint32_t)i) {
            if (!p.scaled) {
                for (int c = 0; c < C; ++c) {
                    const float a = p.s.activation[(eN + i) * C + c];
                    if (a != 0.0f) {
                        float v = out[(size_t)cell * PS + c] + a;
                        if (p.has_max_val) v = fminf(v, (float)p.max_val);
                        out[(size_t)cell * PS + c] = v;
                    }
                }
            } else {
                // grid holds u = v / f0^S_write.  Value after this update's (conceptual) sweep:
                // v = u * f0^(S+1), zero below the cut (the per-step cut is monotone, so testing the
                // current value equals testing every intermediate one); wall cells hold no
                // pheromone at deposit time (walls.py:30).
                const bool on_wall = test_bit(walls, cell_id);
                bool wrote = false;
                for (int c = 0; c < C; ++c) {
                    const float a = p.s.activation[(eN + i) * C + c];
                    if (a != 0.0f) {
                        double v = (double)out[(size_t)cell * PS + c] * p.g_dep;
                        if (v < p.threshold || on_wall) v = 0.0;
                        v += (double)a;
                        if (p.has_max_val) v = fmin(v, p.max_val);
                        out[(size_t)cell * PS + c] = (float)(v * p.inv_g_dep);
                        wrote = true;
                    } else if (on_wall) {
                        out[(size_t)cell * PS + c] = 0.0f;
                    }
                }
                if (on_wall && wrote) p.s.walldep_cell[eN + i] = (int32_t)cell;
            }
        }
    }
    // ---- Anthill.update (anthill.py:41-46), sparse: after the first full collect the only
    // non-zero food on the area is what this step's exchange winners wrote there.
    double gain = 0.0;
    for (int i = tid; (phases & UPD_ANTHILL) && i < N; i += T) {
        const int32_t dc = p.s.dirty_cell[eN + i];
        if (dc >= 0) {
            gain += (double)food[dc];
            food[dc] = 0.0f;
            p.s.dirty_cell[eN + i] = -1;
        }
    }
    // block sum of gain (integer-valued in every reference workload -> order-independent)
    for (int o = 32; o > 0; o >>= 1) gain += __shfl_down(gain, o);
    if (lane == 0) red[wave] = gain;
    __syncthreads();
    if (tid == 0 && (phases & UPD_ANTHILL)) {
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += red[w];
        if (s != 0.0) p.s.anthill_food[e] += s;
        p.s.timestep[e] = ts;
    }
}
