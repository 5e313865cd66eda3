// Test helper, compiled for the HOST with hipcc (no GPU needed): the index's sort word of every point, by the product's own
// code (csrc/pcpx_curve.h: quantisation on the voxel grid, table-driven Hilbert key, index in the low bits, the all-ones
// word for a point outside the grid) -- so that CPU tests order points exactly as the GPU build does.
#include "pcpx_curve.h"

static const pcpx::u32 table_host[pcpx::HILBERT_TABLE_WORDS] = {PCPX_HILBERT_TABLE_INIT};

extern "C" void pcpx_test_sort_words(const float* xyz, unsigned long long n, const float* grid6, unsigned long long* out_words)
{
    const pcpx::CurveGrid g = pcpx::curve_grid(grid6[0], grid6[1], grid6[2], grid6[3], grid6[4], grid6[5]);
    const int idx_bits = pcpx::index_bits_for(n);
    for (unsigned long long i = 0; i < n; ++i) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const bool ok = (x >= grid6[0] && y >= grid6[1] && z >= grid6[2]) && (x <= grid6[3] && y <= grid6[4] && z <= grid6[5]);
        out_words[i] = ok ? pcpx::sort_word(pcpx::curve_key_inside(x, y, z, g, table_host, idx_bits), i, idx_bits) : pcpx::outside_word(i, idx_bits);
    }
}
