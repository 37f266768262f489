// tools/rows_i8_bench.hip -- the integer pre-smoothing pass (glv_bars_rows_i8_kernel) alone, as a stand-alone executable: timing of A/B builds
// (GLV_EXP_I8_* switches) without rebuilding the product library, and a check of the plain build against the integer formula on the host.
// Not part of the product; built by tools/rows_i8_bench.sh.     rows_i8_bench [n] [rows] [reps]
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
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    std::vector<BarDesc> desc; std::vector<float> w;
    make_bar_taps(desc, w, n, n, 0.025f, 0.5f);
    std::vector<BarMTile> it; std::vector<BarTile> rounds; std::vector<int8_t> wq; std::vector<BarIFin> fin;
    uint32_t bins = 0;
    for (uint32_t b : {160u, 288u, 448u, 832u, 1600u}) { if (make_bar_itiles(it, wq, fin, rounds, desc, w, n, b, 4u) && !rounds.empty()) { bins = b; break; } }
    if (!bins) { fprintf(stderr, "no rounds for n=%u\n", n); return 2; }
    size_t steps = 0; for (auto& t : it) steps += t.steps;
    std::vector<uint16_t> tex(rows * n);
    uint32_t lcg = 12345u;
    for (auto& v : tex) { lcg = lcg * 1664525u + 1013904223u; v = (uint16_t) (lcg >> 16); }
    uint16_t *d_tex, *d_out; int8_t* d_wq; BarIFin* d_fin; BarTile* d_rounds; BarMTile* d_it;
    CK(hipMalloc(&d_tex, 2 * rows * n)); CK(hipMalloc(&d_out, 2 * rows * n));
    CK(hipMalloc(&d_wq, wq.size())); CK(hipMalloc(&d_fin, sizeof(BarIFin) * fin.size()));
    CK(hipMalloc(&d_rounds, sizeof(BarTile) * rounds.size())); CK(hipMalloc(&d_it, sizeof(BarMTile) * it.size()));
    CK(hipMemcpy(d_tex, tex.data(), 2 * rows * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wq, wq.data(), wq.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_fin, fin.data(), sizeof(BarIFin) * fin.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rounds, rounds.data(), sizeof(BarTile) * rounds.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_it, it.data(), sizeof(BarMTile) * it.size(), hipMemcpyHostToDevice));
    const BarIRowsTables rt{d_it, (uint32_t) it.size(), d_wq, d_fin, d_rounds, (uint32_t) rounds.size(), bins};
    CK(prepare_bars_i8(n, &rt));
    CK(hipMemset(d_out, 0xee, 2 * rows * n));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(launch_bars_i8(d_tex, false, d_out, rows, n, n, &rt, nullptr, true));
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int k = 0; k < 3; ++k) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) CK(launch_bars_i8(d_tex, false, d_out, rows, n, n, &rt, nullptr, true));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        best = ms < best ? ms : best; sum += ms;
    }
    // the integer formula on the host, for the first and last rows
    std::vector<uint16_t> got(rows * n);
    CK(hipMemcpy(got.data(), d_out, 2 * rows * n, hipMemcpyDeviceToHost));
    size_t bad = 0, checked = 0;
    for (size_t r : {(size_t) 0, rows / 2 + 1, rows - 1}) {
        for (uint32_t k = 0; k < n; ++k) {
            std::vector<int32_t> W;
            const int P = bar_int_weights(w.data() + desc[k].tap_offset, desc[k].count, W);
            uint16_t want = 0;
            if (P >= 0) {
                long long tot = 0;
                for (uint32_t i = 0; i < desc[k].count; ++i) tot += (long long) W[i] * tex[r * n + desc[k].first_bin + i];
                want = (uint16_t) ((tot + (1LL << (P - 1))) >> P);
            }
            bad += want != got[r * n + k]; ++checked;
        }
    }
    printf("n=%u rows=%zu ring=%u rounds=%zu steps/row-block=%zu wq=%zu KB: %.4f ms (best %.4f)  %zu of %zu checked texels differ\n", n, rows, bins, rounds.size(), steps,
           wq.size() / 1024, sum / 3, best, bad, checked);
    return 0;
}
