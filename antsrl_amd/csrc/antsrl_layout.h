// antsrl_layout.h — where a cell's record lies in the cell-record arrays (KP::tiled / KP::ftile, antsrl_device.h), in ONE
// place: used by prec_xy / frec_xy (antsrl_util.h), by k_perceive's gather index (antsrl_perceive.hip) and by the host-side
// test (tests/test_layout.py compiles this header with g++ and checks that each mapping is a bijection onto [0, W * H) and
// that every block is one aligned run of records = one 128-byte line: 2 x 4 cells of 16-byte records, 4 x 4 cells of
// 8-byte ones).
#pragma once
#include <stdint.h>

#if !defined(__HIPCC__) && !defined(__host__)
#define __host__
#define __device__
#endif

// Record index of cell (x, y), 0 <= x < W (even), 0 <= y < H (a multiple of 4): blocks of 2 (x) by 4 (y) cells, the blocks
// row-major over (x / 2, y / 4), the eight cells of a block in (x & 1, y & 3) order.
__host__ __device__ inline uint32_t tiled_slot(const int x, const int y, const int H)
{
    return (uint32_t)((((x >> 1) * (H >> 2) + (y >> 2)) << 3) + ((x & 1) << 2) + (y & 3));
}

// The same for the 8-byte {food, META} records of the explicit-sweep layout (KP::ftile): blocks of 4 (x) by 4 (y) cells =
// sixteen records = one line; 0 <= x < W, 0 <= y < H, both multiples of 4.  A rotated 7 x 7 patch touches ~7.7 lines of
// this array instead of ~12.4 row-major ones (1 x 16 cells per line).
__host__ __device__ inline uint32_t tiled44_slot(const int x, const int y, const int H)
{
    return (uint32_t)((((x >> 2) * (H >> 2) + (y >> 2)) << 4) + ((x & 3) << 2) + (y & 3));
}
