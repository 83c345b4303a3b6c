// antsrl_update_one.h — Environment.update minus the pheromone sweep for N <= 1024, ONE ant per thread, as a device
// function: the body of k_update_one (antsrl_update.hip) and the first half of k_update_move (antsrl_perceive.hip).
#pragma once
#include "antsrl_util.h"
#include "antsrl_update_env.h"

// LDS of k_update_one: hash [HT] keys + values, rock table [R][4] + new centres [R][2], reduction and
// scan scratch per wave, ants' x / y [N], collider flag per rock [R].
__host__ __device__ inline size_t update_one_lds_bytes(int HT, int R, int nwaves, int N)
{
    return align_up(8 * (size_t)HT, 16) + 48 * (size_t)(R > 0 ? R : 1) + 8 * (size_t)nwaves +
           align_up(4 * (size_t)nwaves, 8) + 16 * (size_t)N + align_up(4 * (size_t)R, 16) + 16;
}

// k_update for N <= 1024: ONE ant per thread.  Same phases and the same arithmetic as update_env,
// restructured around latency: every independent global load of the ant (position, previous position,
// reward tint, activation, the two sparse-update cells) is issued before anything waits; the ant's
// state then stays in registers across the phases, the rock pass reads all ants' positions from LDS,
// and each array is written back once.  update_env re-reads x / y from HBM in every phase (a dependent
// round trip behind each barrier) — it remains the path for N > 1024 and for the fused launch.
// `g_dep` / `inv_g_dep`: KP::g_dep / inv_g_dep as they stood at the update's own call (a deferred update runs after the
// host has advanced them for the next observation).
// `e`: the environment of this workgroup (blockIdx.x, or counted from the other end: env_of_block in antsrl_util.h).
// FWD_REC (k_update_move on interleaved records, p.ps == 4): the deposit cell's whole 16-byte record is loaded at once and
// its food / META words are handed to the move — a template parameter so that no run-time branch joins two load forms
// (the join's register copies would wait for the load on the spot).
template <int C, bool FWD_REC = false>
__device__ __forceinline__ void update_one_body(const KP &p, const int e, const double *__restrict__ wall_jitter, const int out_buf,
                                                unsigned char *smem, const double g_dep, const double inv_g_dep, UmFwd *fw = nullptr)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = T >> 6;
    const int N = p.N, W = p.W, H = p.H, R = p.R;
    const size_t G = (size_t)W * H, eN = (size_t)e * N;
    uint32_t *hkeys = (uint32_t *)smem, *hvals = hkeys + p.HT;
    double *rock = (double *)(smem + align_up(8 * (size_t)p.HT, 16)); // [R][4] cx, cy, radius, weight
    double *rk = rock + 4 * (R > 0 ? R : 1);                          // [R][2] new centres
    double *red = rk + 2 * (R > 0 ? R : 1);                           // [nwaves]
    uint32_t *wave_tot = (uint32_t *)(red + nwaves);                  // [nwaves]
    double *sx = (double *)((unsigned char *)wave_tot + align_up(4 * (size_t)nwaves, 8)), *sy = sx + N;
    uint32_t *rk_hit = (uint32_t *)(sy + N);                          // [R] pass 1 found a collider of this rock

    const uint32_t *walls = p.s.walls_bits + (size_t)e * p.words;
    const FoodView food{p.s.food + (size_t)e * G * p.fs, p.fs};
    const size_t PS = (size_t)p.ps; // floats per cell of the pheromone array
    float *out = p.s.phero[out_buf] + (size_t)e * G * PS;
    const bool on = tid < N;
    const size_t a = eN + (on ? tid : 0);

    // ---- every independent load first
    double x = p.s.x[a], y = p.s.y[a];
    const double px0 = STP_LD(p.s.prev_x[a]), py0 = STP_LD(p.s.prev_y[a]); // used on a wall hit only; prefetched all the same
    const uint8_t rstate = STP_LD(p.s.reward_state[a]);
    const double theta0 = p.s.theta[a]; // used on a wall hit only
    float act[C];
#pragma unroll
    for (int c = 0; c < C; ++c) act[c] = STP_LD(p.s.activation[a * C + c]);
    // (unconditional loads on the clamped index + selects: a load inside a per-lane branch is followed by
    // its own s_waitcnt vmcnt(0))
    const int32_t wc_l = STP_LD(p.s.walldep_cell[a]), dc_l = STP_LD(p.s.dirty_cell[a]);
    const int32_t wc = (p.scaled && on) ? wc_l : -1;
    const int32_t dc = on ? dc_l : -1;
    const int ts = p.s.timestep[e] + 1; // environment.py:45
    if (tid < R) {
        rock[4 * tid + 0] = p.s.rock_cx[(size_t)e * R + tid];
        rock[4 * tid + 1] = p.s.rock_cy[(size_t)e * R + tid];
        rock[4 * tid + 2] = p.s.rock_r[(size_t)e * R + tid];
        rock[4 * tid + 3] = p.s.rock_w[(size_t)e * R + tid];
    }
    for (int h = tid; h < p.HT; h += T) {
        hkeys[h] = HASH_EMPTY;
        hvals[h] = 0u;
    }
    const float food_dc_l = food[dc_l >= 0 ? dc_l : 0]; // nobody else touches a dirty cell during the update
    const float food_dc = dc >= 0 ? food_dc_l : 0.0f;

    // ---- Walls.update, walls.py:25-28
    // (the wall / anthill tests of this kernel read the 8 KB bit maps, not the META words of the cell records: with the META
    //  words c3 gains 0.9 % — 16 MB less competing for the Infinity Cache — and the latency-bound small batches lose 1-2 %, the
    //  first touch of the record line moving to the head of the dependency chain: profiles/r03/um_metabits_ab.txt)
    const bool hit = on && test_bit(walls, (uint32_t)((int)x * H + (int)y));
    UM_STAMP(1); // (issued; the wait sits in front of the first use below)
    double u = 0.0;
    if (wall_jitter) { // k-th colliding ant (index order) takes the k-th draw
        uint32_t tot;
        const uint32_t rank = block_excl_scan_flag(hit, wave_tot, lane, wave, nwaves, &tot);
        if (hit) u = wall_jitter[eN + rank];
    } else if (hit) {
        u = jitter_u01(p.rng_seed, p.env_id_base + (uint32_t)e, (uint32_t)ts, (uint32_t)tid); // (the env's GLOBAL id)
    }
    bool moved = hit;
    if (hit) {
        x = px0;
        y = py0;
        if (!fw) p.s.theta[a] = theta0 + (u - 0.5); // theta is NOT re-wrapped here
    }
    // (k_update_move: exactly what the move would load back; the move of the same launch overwrites x / y / theta, so the
    //  update's own stores of them would be dead)
    if (fw) fw->th = hit ? theta0 + (u - 0.5) : theta0;
    UM_STAMP_ON(2, (int)hit); // (the wall bits are back)

    // ---- CircleObstacles.update, circle_obstacles.py:35-58
    if (R > 0) {
        if (on) {
            sx[tid] = x;
            sy[tid] = y;
        }
        __syncthreads();
        UM_STAMP(3);
        // pass 1: centres -= sum_over_ants(push)/weight, ants summed in index order (colliding ones only:
        // the others contribute exact zeros)
        for (int q = wave; q < R; q += nwaves) {
            const double cx = rock[4 * q + 0], cy = rock[4 * q + 1], rad = rock[4 * q + 2];
            const double rad2_hi = rad * rad * (1.0 + 1e-12) + 1e-300; // d2 above this: sqrt(d2) > rad for sure
            double sumx = 0.0, sumy = 0.0;
            bool any = false;
            for (int base = 0; base < N; base += 64) {
                const int i = base + lane;
                double px = 0.0, py = 0.0;
                bool col = false;
                if (i < N) {
                    const double vx = cx - sx[i], vy = cy - sy[i];
                    const double d2 = vx * vx + vy * vy;
                    if (!(d2 > rad2_hi)) {
                        const double d = sqrt(d2);
                        const double f = 1 - rad / (d + 0.001);
                        px = vx * f; py = vy * f;
                        col = !(d > rad);
                    }
                }
                unsigned long long m = __ballot(col);
                any |= m != 0;
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    sumx += __shfl(px, l);
                    sumy += __shfl(py, l);
                }
            }
            if (lane == 0) {
                rk[2 * q + 0] = cx - sumx / rock[4 * q + 3];
                rk[2 * q + 1] = cy - sumy / rock[4 * q + 3];
                rk_hit[q] = any ? 1u : 0u;
            }
        }
        __syncthreads();
        UM_STAMP(4);
        if (tid < R) {
            p.s.rock_cx[(size_t)e * R + tid] = rk[2 * tid + 0];
            p.s.rock_cy[(size_t)e * R + tid] = rk[2 * tid + 1];
        }
        // pass 2 (:53-58): ants pushed out of the UPDATED rocks, then warp_xy
        double ax = 0.0, ay = 0.0;
        for (int q = 0; q < R; ++q) {
            // A rock no ant touched in pass 1 has not moved, and an ant is pushed in pass 2 exactly when it was a collider of
            // pass 1 (same centre, same d <= radius test): every push of this rock would be an exact zero
            if (!rk_hit[q]) continue;
            const double vx = rk[2 * q + 0] - x, vy = rk[2 * q + 1] - y;
            const double rad = rock[4 * q + 2];
            const double d2 = vx * vx + vy * vy;
            if (d2 > rad * rad * (1.0 + 1e-12) + 1e-300) continue; // adds an exact +0.0
            const double d = sqrt(d2);
            const double f = 1 - rad / (d + 0.001);
            double px = vx * f, py = vy * f;
            if (d > rad) { px = 0.0; py = 0.0; }
            ax += px; ay += py;
        }
        x = warp_coord(x + ax, (double)W);
        y = warp_coord(y + ay, (double)H);
        moved = true;
        UM_STAMP(5);
    }

    // ---- Ants.update, ants.py:123-130: prev := cur; deposit (pheromone.py:36-41)
    __syncthreads(); // the hash table is initialised (R == 0 and library jitter: no barrier so far)
    const uint32_t cell_id = (uint32_t)((int)x * H + (int)y); // row-major id: the wall bit map
    const uint32_t cell = prec_xy(p, (int)x, (int)y);           // the cell's PHEROMONE record: deposit, hash key, wall-deposit list
    // the deposit cell's old values, loaded by every ant ahead of the barriers (only the cell's winner uses
    // them; no other ant writes this cell in this update except the wall-deposit clear, and a deposit on a
    // wall cell ignores the old value)
    float pold[C];
    constexpr bool fw_rec = FWD_REC && C == 2 && !(UM_ABL & 4);
    if constexpr (fw_rec) {
        // interleaved records: the WHOLE record of the cell in one 16-byte load — the move's food value and area bit with it
        const stream_f4 rec = *reinterpret_cast<const stream_f4 *>(out + (size_t)cell * 4);
        pold[0] = rec.x;
        pold[C - 1] = rec.y;
        fw->food = rec.z;
        fw->meta = __float_as_uint(rec.w);
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) pold[c] = (UM_ABL & 4) ? 0.0f : out[(size_t)cell * PS + c]; // (UM_ABL 4: ablation, antsrl_device.h)
    }
    if (fw) {
        // The move's rotation and sincos (RL_api.py:190-196, ants.py:62-67), HERE: they need theta and the action only, and
        // the record load above is in flight — ~400 float64 VALU instructions under a memory round trip instead of behind it.
        double th = fw->th;
        if (fw->has_rot) th = np_mod_d(th + (double)fw->rot * p.max_rot_speed, 2 * PI_D);
        fw->th_new = th;
        sincos(th, &fw->sn, &fw->cs);
        fw->pre = 1;
        fw->ts = ts;
        // ... and the rotation of the NEXT observation's frame, cos / sin(theta + pi/2) (RL_api.py:100-108; k_perceive's
        // prologue would compute it behind a memory round trip of its own): parked in LDS until the move has the new position
        if (fw->frm_off) {
            double st2, ct2;
            sincos(th + PI_D * 0.5, &st2, &ct2);
            double *fs = reinterpret_cast<double *>(smem + fw->frm_off) + 2 * tid;
            fs[0] = ct2;
            fs[1] = st2;
        }
    }
    if (on) {
        if (moved && !fw) {
            p.s.x[a] = x;
            p.s.y[a] = y;
        }
        STP_ST(p.s.prev_x[a], x);
        STP_ST(p.s.prev_y[a], y);
        if (fw) {
            fw->x = x;
            fw->y = y;
        }
        lww_insert(hkeys, hvals, (uint32_t)p.HT - 1, cell, (uint32_t)tid);
        STP_ST(p.s.reward_state[a], (uint8_t)((double)rstate * 0.9)); // :130
    }
    __syncthreads();
    UM_STAMP(6);
    if (p.scaled) {
        // a deposit that landed on a WALL cell in the previous update is zeroed by this update's Walls
        // pass (walls.py:30), before this update's deposits
        if (wc >= 0) {
            for (int c = 0; c < C; ++c) out[(size_t)wc * PS + c] = 0.0f;
            p.s.walldep_cell[a] = -1;
        }
        __syncthreads();
        UM_STAMP(7);
    }
    if constexpr (fw_rec) {
        // The record's first use is HERE (and its food / META words in the move): without this the compiler extracts the area
        // bit right behind the load and waits for it there — the whole round trip in front of the hash inserts and two barriers.
        asm volatile("" : "+v"(pold[0]), "+v"(pold[C - 1]), "+v"(fw->food), "+v"(fw->meta));
    }
    double gain = 0.0;
    if (on) {
        const bool winner = lww_winner(hkeys, hvals, (uint32_t)p.HT - 1, cell) == (uint32_t)tid;
        if (fw && winner) fw->m |= UMFWD_WIN; // (the move's food exchange is decided over the same cells by the same rule)
        if (winner) {
            if (!p.scaled) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (act[c] != 0.0f) {
                        float v = pold[c] + act[c];
                        if (p.has_max_val) v = fminf(v, (float)p.max_val);
                        out[(size_t)cell * PS + c] = v;
                    }
                }
            } else { // scaled units, see update_env
                // (k_update_move: the cell's own record is in registers — its META word carries the wall bit the perception reads,
                //  antsrl_state.hip; the bit map would be one more dependent load inside this phase)
                bool on_wall;
                if constexpr (fw_rec) on_wall = (fw->meta & META_WALL) != 0;
                else on_wall = test_bit(walls, cell_id);
                bool wrote = false;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (act[c] != 0.0f) {
                        double v = (double)pold[c] * g_dep;
                        if (v < p.threshold || on_wall) v = 0.0;
                        v += (double)act[c];
                        if (p.has_max_val) v = fmin(v, p.max_val);
                        if (!(UM_ABL & 4) || v == 12345.678) out[(size_t)cell * PS + c] = (float)(v * inv_g_dep);
                        wrote = true;
                    } else if (on_wall) {
                        out[(size_t)cell * PS + c] = 0.0f;
                    }
                }
                if (on_wall && wrote) p.s.walldep_cell[a] = (int32_t)cell;
            }
        }
        // ---- Anthill.update (anthill.py:41-46), sparse form
        if (dc >= 0) {
            gain = (double)food_dc;
            food[dc] = 0.0f;
            p.s.dirty_cell[a] = -1;
        }
    }
    UM_STAMP(8); // (deposit and collect stores issued)
    for (int o = 32; o > 0; o >>= 1) gain += __shfl_down(gain, o);
    if (lane == 0) red[wave] = gain;
    __syncthreads();
    if (tid == 0) {
        double s2 = 0.0;
        for (int w = 0; w < nwaves; ++w) s2 += red[w];
        if (s2 != 0.0) p.s.anthill_food[e] += s2;
        p.s.timestep[e] = ts;
    }
}
