// Host-side check of the cell-record layout (antsrl_amd/csrc/antsrl_layout.h), compiled by tests/test_layout.py.
#include <cstddef>
#include <vector>
#include "../../antsrl_amd/csrc/antsrl_layout.h"

// number of violations over a list of grid shapes: the mapping must be a bijection onto [0, W*H), a 2 x 4 block must be one
// aligned run of 8 records, and cells that differ by one step in x or y must lie in the same or an adjacent-indexed block row
extern "C" long layout_violations(void)
{
    long bad = 0;
    const int shapes[][2] = {{2, 4}, {2, 8}, {4, 4}, {16, 12}, {64, 64}, {96, 48}, {40, 36}, {256, 256}, {254, 260}, {512, 512}, {6, 1020}};
    for (const auto &s : shapes) {
        const int W = s[0], H = s[1];
        std::vector<char> seen((std::size_t)W * H, 0);
        for (int x = 0; x < W; ++x)
            for (int y = 0; y < H; ++y) {
                const uint32_t r = tiled_slot(x, y, H);
                if (r >= (uint32_t)(W * H) || seen[r]) { ++bad; continue; }
                seen[r] = 1;
                // the block's first cell is (x & ~1, y & ~3); its records are 8 consecutive, 8-aligned indices
                const uint32_t r0 = tiled_slot(x & ~1, y & ~3, H);
                if ((r0 & 7u) != 0 || r < r0 || r >= r0 + 8 || r - r0 != (uint32_t)(((x & 1) << 2) + (y & 3))) ++bad;
            }
    }
    // the 8-byte {food, META} records (KP::ftile): blocks of 4 x 4 cells = 16 consecutive, 16-aligned record indices
    const int shapes44[][2] = {{4, 4}, {4, 8}, {8, 4}, {16, 12}, {64, 64}, {96, 48}, {40, 36}, {256, 256}, {252, 260}, {512, 512}, {8, 1020}};
    for (const auto &s : shapes44) {
        const int W = s[0], H = s[1];
        std::vector<char> seen((std::size_t)W * H, 0);
        for (int x = 0; x < W; ++x)
            for (int y = 0; y < H; ++y) {
                const uint32_t r = tiled44_slot(x, y, H);
                if (r >= (uint32_t)(W * H) || seen[r]) { ++bad; continue; }
                seen[r] = 1;
                const uint32_t r0 = tiled44_slot(x & ~3, y & ~3, H);
                if ((r0 & 15u) != 0 || r < r0 || r >= r0 + 16 || r - r0 != (uint32_t)(((x & 3) << 2) + (y & 3))) ++bad;
            }
    }
    return bad;
}
