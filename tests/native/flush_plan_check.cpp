// Host-side exhaustive check of the observation copy-out plan (antsrl_amd/csrc/antsrl_flush.h), the code the
// device kernels run: for every row length the pipelined loops accept, every destination misalignment, every
// position of the run against a 128-byte line and both run lengths (one row = odd tail, two rows), every lane's
// 16-byte and edge stores must stay inside the run's own elements and together cover every element.
// Returns the number of violations (0 = pass); prints the first few.
#include <stdio.h>
#include <string.h>
#include "../../antsrl_amd/csrc/antsrl_flush.h"

extern "C" long flush_plan_violations(int verbose)
{
    long bad = 0;
    static unsigned char cov[2 * 368 + 16];
    for (uint32_t row = 8; row <= 368; ++row)
        for (uint32_t two = 0; two < 2; ++two)
            for (uint32_t mis = 0; mis < 4; ++mis)
                for (uint32_t phase = 0; phase < 8; ++phase) {
                    const uint32_t rowp = two ? 2 * row : row;
                    memset(cov, 0, sizeof(cov));
                    for (uint32_t lane = 0; lane < 64; ++lane) {
                        const FlushPlanF32 f = flush_plan_f32(lane, mis, rowp, phase);
                        const uint32_t js[3] = {f.j1, f.j2, f.j3};
                        for (int k = 0; k < 3; ++k)
                            for (uint32_t t = 0; t < 4; ++t) {
                                const uint32_t el = 4 * js[k] + t;
                                if (el < mis || el >= mis + rowp) {
                                    if (verbose && bad < 5) printf("f32 row=%u rowp=%u mis=%u phase=%u lane=%u: float4 %u outside\n", row, rowp, mis, phase, lane, js[k]);
                                    ++bad;
                                } else cov[el - mis] = 1;
                            }
                        if (f.fe < mis || f.fe >= mis + rowp) { ++bad; if (verbose && bad < 5) printf("f32 edge outside\n"); }
                        else cov[f.fe - mis] = 1;
                    }
                    for (uint32_t el = 0; el < rowp; ++el)
                        if (!cov[el]) { ++bad; if (verbose && bad < 5) printf("f32 row=%u rowp=%u mis=%u phase=%u: element %u not covered\n", row, rowp, mis, phase, el); }
                }
    for (uint32_t row = 16; row <= 368; ++row)
        for (uint32_t two = 0; two < 2; ++two)
            for (uint32_t mis = 0; mis < 8; ++mis)
                for (uint32_t phase = 0; phase < 8; ++phase) {
                    const uint32_t rowp = two ? 2 * row : row;
                    memset(cov, 0, sizeof(cov));
                    for (uint32_t lane = 0; lane < 64; ++lane) {
                        const FlushPlanB16 f = flush_plan_b16(lane, mis, rowp, phase);
                        const uint32_t gs[2] = {f.g1, f.g2};
                        for (int k = 0; k < 2; ++k)
                            for (uint32_t t = 0; t < 8; ++t) {
                                const uint32_t el = 8 * gs[k] + t;
                                if (el < mis || el >= mis + rowp) {
                                    if (verbose && bad < 5) printf("b16 row=%u rowp=%u mis=%u phase=%u lane=%u: piece %u outside\n", row, rowp, mis, phase, lane, gs[k]);
                                    ++bad;
                                } else cov[el - mis] = 1;
                            }
                        if (f.fe < mis || f.fe >= mis + rowp) { ++bad; if (verbose && bad < 5) printf("b16 edge outside\n"); }
                        else cov[f.fe - mis] = 1;
                    }
                    for (uint32_t el = 0; el < rowp; ++el)
                        if (!cov[el]) { ++bad; if (verbose && bad < 5) printf("b16 row=%u rowp=%u mis=%u phase=%u: element %u not covered\n", row, rowp, mis, phase, el); }
                }
    return bad;
}
