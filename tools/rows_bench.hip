// tools/rows_bench.hip -- the pre-smoothing pass kernel (glv_bars_rows_kernel) alone, as a stand-alone executable: timing of A/B builds
// (GLV_EXP_ROWS_* switches, window sizes) without rebuilding the product library.  Not part of the product; built by tools/rows_bench.sh.
//   rows_bench [n] [rows] [bins] [reps]
#include "../glava_amd/csrc/glv_misc.hip"
#include "../glava_amd/csrc/glv_tables.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace glv {
#define GLV_STUB(K)                                                                                                        \
    hipError_t launch_frame_##K(int, int, int, const FrameArgs&, int, hipStream_t) { return hipErrorUnknown; }            \
    int frame_variants_##K() { return 1; }                                                                                 \
    int frame_variant_ok_##K(int, int, int) { return 0; }                                                                  \
    FrameGeometry frame_geometry_##K(int) { return FrameGeometry{}; }
GLV_STUB(7) GLV_STUB(8) GLV_STUB(9) GLV_STUB(10) GLV_STUB(11) GLV_STUB(12) GLV_STUB(13) GLV_STUB(14)
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    using namespace glv;
    const uint32_t n = argc > 1 ? (uint32_t) atoi(argv[1]) : 4096u;
    const size_t rows = argc > 2 ? (size_t) atol(argv[2]) : (size_t) 32768 * 4096 / n;
    const uint32_t bins = argc > 3 ? (uint32_t) atoi(argv[3]) : (n >= 4096 ? 288u : 160u);
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    std::vector<BarDesc> desc; std::vector<float> w;
    make_bar_taps(desc, w, n, n, 0.025f, 0.5f);
    std::vector<BarMTile> mt; std::vector<BarTile> rounds; std::vector<float> wt, wsum;
    if (!make_bar_mtiles(mt, wt, wsum, rounds, desc, w, n, bins, 4u) || rounds.empty()) { fprintf(stderr, "no rounds for n=%u ring=%u\n", n, bins); return 2; }
    size_t steps = 0; for (auto& t : mt) steps += t.steps;
    std::vector<float> spec(rows * n);
    uint32_t lcg = 12345u;
    for (auto& v : spec) { lcg = lcg * 1664525u + 1013904223u; v = (float) (lcg >> 8) * (1.0f / 16777216.0f); }
    float *d_spec, *d_out, *d_wt, *d_wsum; BarTile* d_rounds; BarMTile* d_mt;
    CK(hipMalloc(&d_spec, sizeof(float) * rows * n)); CK(hipMalloc(&d_out, sizeof(float) * rows * n));
    CK(hipMalloc(&d_wt, sizeof(float) * wt.size())); CK(hipMalloc(&d_wsum, sizeof(float) * wsum.size()));
    CK(hipMalloc(&d_rounds, sizeof(BarTile) * rounds.size())); CK(hipMalloc(&d_mt, sizeof(BarMTile) * mt.size()));
    CK(hipMemcpy(d_spec, spec.data(), sizeof(float) * rows * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wt, wt.data(), sizeof(float) * wt.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_wsum, wsum.data(), sizeof(float) * wsum.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rounds, rounds.data(), sizeof(BarTile) * rounds.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_mt, mt.data(), sizeof(BarMTile) * mt.size(), hipMemcpyHostToDevice));
    const BarRowsTables rt{d_mt, (uint32_t) mt.size(), d_wt, d_wsum, d_rounds, (uint32_t) rounds.size(), bins};
    CK(prepare_bars_rows(n, &rt));
    {
        int nb = -1;
        const size_t lds = sizeof(float) * (size_t) GLV_ROWS_RB * bins;
        hipError_t e = bins == 288 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, glv_bars_rows_kernel<288, GLV_ROWS_RB>, 256, lds)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, glv_bars_rows_kernel<160, GLV_ROWS_RB>, 256, lds);
        printf("occupancy: %d workgroups of 256 per CU with %zu B of LDS (%s)\n", nb, lds, hipGetErrorString(e));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r16 = 0; r16 < 2; ++r16) {
        for (int i = 0; i < 5; ++i) CK(launch_bars(d_spec, d_out, rows, n, n, 0, nullptr, nullptr, nullptr, nullptr, r16 != 0, &rt));
        CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0;
        for (int k = 0; k < 3; ++k) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) CK(launch_bars(d_spec, d_out, rows, n, n, 0, nullptr, nullptr, nullptr, nullptr, r16 != 0, &rt));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            best = ms < best ? ms : best; sum += ms;
        }
        printf("n=%u rows=%zu ring=%u rounds=%zu steps/row-block=%zu %s: %.4f ms (best %.4f)  %.1f TFLOP/s on the matrix cores\n", n, rows, bins, rounds.size(), steps, r16 ? "r16" : "f32",
               sum / 3, best, (double) steps * 2.0 * 2048.0 * 2.0 * (double) ((rows + 63) / 64) / (sum / 3 * 1e-3) * 1e-12);      // (64-row blocks: two MFMAs per step)
    }
    return 0;
}
