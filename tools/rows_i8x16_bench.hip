// tools/rows_i8x16_bench.hip -- PROTOTYPE (not part of the product): the integer pre-smoothing pass on v_mfma_i32_16x16x64_i8 --
// a wave takes 64 rows x 16 bars (four 16-row blocks x four digit planes = 64 accumulator registers instead of 128), steps of 64 bins, three workgroups
// per CU instead of two.  Same ring, rounds, weight stream and epilogue as glv_bars_rows_i8_kernel (the text below is that kernel with the tile
// geometry changed); own host tables (tiles of 16 bars, weight fragments in the 16x16x64 b-operand layout).  Checked against the integer formula on the
// host like rows_i8_bench.  Built by tools/rows_i8_bench.sh x16 (same flags).     rows_i8x16_bench [n] [rows] [reps]
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

template <int S, int RB, bool F32IN, bool R16>
__global__ void __launch_bounds__(64 * kRowsWaves, 3) glv_bars_rows_i8x16_kernel(const void* __restrict__ rows_in, void* __restrict__ bars_out, size_t nrows, uint32_t n,
                                                                              uint32_t bars, const BarTile* __restrict__ rounds, uint32_t nrounds, uint32_t rounds_per_wg,
                                                                              const BarMTile* __restrict__ tiles, const glv_i4v* __restrict__ wq,
                                                                              const BarIFin* __restrict__ fin) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(S % 32 == 0 && RB == 64, "prototype: 64 rows = four 16-row blocks");
    extern __shared__ __attribute__((aligned(16))) char i8_lds[];       // [2 planes][RB rows][S + 16 bytes]
    constexpr uint32_t PITCH = S + 16, S16 = S / 16, G = RB / 16, CPI = 64 * kRowsWaves / RB;       // CPI: columns of 8 bins one sweep of the workgroup fetches
    char* plane_h = i8_lds;
    char* plane_l = i8_lds + (size_t) RB * PITCH;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const size_t row0 = (size_t) blockIdx.x * RB;
    if (row0 >= nrows) return;
    const uint32_t R = (uint32_t) (nrows - row0 < RB ? nrows - row0 : RB);
    const uint32_t t_begin = blockIdx.y * rounds_per_wg, t_end = t_begin + rounds_per_wg < nrounds ? t_begin + rounds_per_wg : nrounds;
    if (t_begin >= t_end) return;
    const uint32_t frow = threadIdx.x % (uint32_t) RB, fcol = threadIdx.x / (uint32_t) RB;
    const size_t srow = row0 + (frow < R ? frow : R - 1);                 // (a partial row block repeats its last row; its stores are masked)
    const char* src = static_cast<const char*>(rows_in) + srow * (size_t) n * (F32IN ? 4u : 2u);
    struct Tex8 { uint32_t d[4]; };                                     // 8 texels, two per dword
    auto fetch = [&](uint32_t bin) -> Tex8 {
        Tex8 v;
        bin = bin + 8u <= n ? bin : n - 8u;                             // (a dummy request -- nothing new to park -- at the row's very end stays inside the row)
        if constexpr (F32IN) {                                          // rows of floats c / 65535 (the pass-by-pass chain): back to the texels, exactly
            const BarW4 a = ld<BarW4>(src, bin * 4u), b = ld<BarW4>(src, bin * 4u + 16u);
            v.d[0] = pack_unorm16(a.w[0], a.w[1]); v.d[1] = pack_unorm16(a.w[2], a.w[3]);
            v.d[2] = pack_unorm16(b.w[0], b.w[1]); v.d[3] = pack_unorm16(b.w[2], b.w[3]);
        } else {
            const glv_i4v a = *reinterpret_cast<const glv_i4v*>(src + bin * 2u);
            v.d[0] = (uint32_t) a.x; v.d[1] = (uint32_t) a.y; v.d[2] = (uint32_t) a.z; v.d[3] = (uint32_t) a.w;
        }
        return v;
    };
    auto park = [&](const Tex8& v, uint32_t bin) {                       // low bytes / high bytes of the 8 texels, biased to signed (c ^ 0x8080)
        const uint32_t l0 = __builtin_amdgcn_perm(v.d[1], v.d[0], 0x06040200u) ^ 0x80808080u, l1 = __builtin_amdgcn_perm(v.d[3], v.d[2], 0x06040200u) ^ 0x80808080u;
        const uint32_t h0 = __builtin_amdgcn_perm(v.d[1], v.d[0], 0x07050301u) ^ 0x80808080u, h1 = __builtin_amdgcn_perm(v.d[3], v.d[2], 0x07050301u) ^ 0x80808080u;
        const uint32_t at = frow * PITCH + bin % (uint32_t) S;
        *reinterpret_cast<uint2*>(plane_l + at) = make_uint2(l0, l1);
        *reinterpret_cast<uint2*>(plane_h + at) = make_uint2(h0, h1);
    };
    // the first round's whole window
    uint32_t filled_to;
    {
        const BarTile T = rounds[t_begin];
        const uint32_t ncol = (T.end - T.origin) / 8u;
        for (uint32_t c0 = fcol; c0 < ncol; c0 += 4u * CPI) {
            Tex8 v4[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) v4[q] = fetch(T.origin + 8u * (c0 + q * CPI < ncol ? c0 + q * CPI : 0u));
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q)
                if (c0 + q * CPI < ncol) park(v4[q], T.origin + 8u * (c0 + q * CPI));
        }
        filled_to = T.end;
    }
    __syncthreads();
    // a-operand: lane l reads 16 consecutive bins (one half of the step's 32) of row l % 32 of a group
    const uint32_t arow = (lane & 15u) * PITCH, ahalf = lane >> 4;
    // The weights: wave w takes tile k0 + w of every round and the host laid those tiles out one behind the other, so the wave reads ONE
    // stream of steps straight across tile boundaries, PF steps ahead: a step's three digit fragments sit in bank (step mod PF), and the bank
    // is reloaded as soon as its step has used it (the stream ends in PF steps of zeros).  Register banks mean an unrolled loop, and a tile may
    // end after any step: the loop below runs over the wave's STREAM, and what a tile's last step is followed by -- parking the next round's
    // bins, the epilogue, the round's barrier, the next tile's set-up -- hangs off each of the PF steps as a side block.
    constexpr int PF = 2;      // (prototype: two banks -- 168 registers for three waves per SIMD; steps are 64 bins, most tiles have one or two)
    static_assert((uint32_t) PF < kBarILookAhead, "the stream that lies last in memory is read PF steps past its last tile: the host pads kBarILookAhead steps");
    const glv_i4v* wp = nullptr;                                        // stream position of the NEXT load (lane-offset)
    glv_i4v wb[PF][3];
    glv_i4v acc[G][4];
    uint32_t t = t_begin, left = 0, ck = 0, next_end = filled_to, nnew = 0;
    BarMTile M = tiles[0];                                              // (overwritten before use)
    Tex8 pre[2];
    auto park_new = [&]() {
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q)
            if (fcol + q * CPI < nnew) park(pre[q], filled_to + 8u * (fcol + q * CPI));
        for (uint32_t c = fcol + 2u * CPI; c < nnew; c += CPI) park(fetch(filled_to + 8u * c), filled_to + 8u * c);
        filled_to = next_end > filled_to ? next_end : filled_to;
    };
    // opens round t: sets up the wave's tile, requests what the NEXT round adds to the ring (parked behind this round's arithmetic) and the
    // tile's epilogue constants.  false: the wave has no tile in this round
    // the epilogue constants {c, s} of the wave's tile: requested a ROUND ahead (the next tile's descriptor is known by then), behind the previous
    // tile's last step -- the full wait in front of that tile's stores covers them, so their first use never waits
    BarIFin fcur = BarIFin{0u, 0u}, fnext = BarIFin{0u, 0u};
    // (the descriptors of round t + 1 -- uniform scalar loads, the tile's dependent on the round's -- are requested while round t runs)
    BarTile Tn = rounds[t_begin];
    BarMTile Mn = tiles[Tn.k0 + wave < Tn.k1 ? Tn.k0 + wave : Tn.k0];
    auto open_round = [&]() -> bool {
        const BarTile T = Tn;
        const bool valid = T.k0 + wave < T.k1;
        if (valid) M = Mn;
        Tn = rounds[t + 1 < t_end ? t + 1 : t];
        Mn = tiles[Tn.k0 + wave < Tn.k1 ? Tn.k0 + wave : Tn.k0];
        next_end = t + 1 < t_end ? Tn.end : filled_to;
        nnew = next_end > filled_to ? (next_end - filled_to) / 8u : 0u;
        if (valid) {
            left = (uint32_t) __builtin_amdgcn_readfirstlane((int) M.steps);
            ck = ((uint32_t) __builtin_amdgcn_readfirstlane((int) (M.origin >> 4)) + ahalf) % S16;      // this lane's 16-bin chunk of the step, in the ring
#pragma unroll
            for (uint32_t g = 0; g < G; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[g][q] = glv_i4v{0, 0, 0, 0};
        }
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) pre[q] = fetch(filled_to + 8u * (fcol + q * CPI < nnew ? fcol + q * CPI : 0u));
        fcur = fnext;
        return valid;
    };
    // rounds without a tile for this wave: park, join the barrier, go on.  false: no round is left
    auto next_tile = [&]() -> bool {
        while (t < t_end) {
            if (open_round()) return true;
            fnext = fin[Mn.k0 + (lane & 15u)];
            // (awaited HERE: a compiler-visible load left in flight across the loop's back edge makes the backend guard every later reuse of its
            // register with a vmcnt(0) -- in front of every step's LDS reads, a drain of the result stores per step)
            asm volatile("" : "+v"(fnext.c), "+v"(fnext.s));
            park_new();
            __syncthreads();
            ++t;
        }
        return false;
    };
    // behind a tile's last step
    // Loads and stores retire on ONE counter, so behind a tile's result stores a wait for a weight fragment is in effect a wait for the stores as
    // well (vmcnt(6) in step() stays CORRECT -- loads retire in order among themselves -- it just lasts until the stores have left too).  Hence
    // every load is awaited BEFORE the stores are issued: the fragments of the next PF steps are then in their registers and those steps (most
    // tiles have no more) wait for nothing while the stores drain behind the round's barrier; `fresh` counts the steps that still need no wait.
    // That full wait includes the weight requests of the tile's last steps, a moment old.  Round 6: ALL of the epilogue's arithmetic (which needs no
    // load: its constants were requested a round ahead) sits between the last step and the wait, so that the L2 round trip of those requests runs
    // under ~1.5 K cycles of vector work instead of in front of them; parking the next round's texels follows the wait.
    uint32_t fresh = 0;
    auto close_tile = [&]() {
        fnext = fin[Mn.k0 + (lane & 15u)];                                      // (padded to whole tiles; Mn: the wave's tile of the next round, or a valid one)
        // a lane's 16 results of a group are one bar (k0 + lane % 32) of the rows 32 g + 8 (r / 4) + 4 (lane / 32) + r % 4
        const uint32_t kb = M.k0 + (lane & 15u);
        const BarIFin f = fcur;
        using OutT = std::conditional_t<R16, uint16_t, float>;
        auto result = [&](uint32_t g, int r) -> OutT {
            const int a0 = acc[g][0][r], a1 = acc[g][1][r], a2 = acc[g][2][r], a3 = acc[g][3][r];
            if constexpr (R16) {
                // (a bar whose weights sum to 0 has c = 0, s = 16 and all-zero digits: 0 >> 16 -- no branch)
                const uint32_t t16 = (uint32_t) ((a3 << 8) + a2 + ((a1 + (a0 >> 8)) >> 8));
                return (uint16_t) ((t16 + f.c) >> f.s);
            } else {
                const int P = (int) f.s + 16;
                const long long tot = ((long long) a3 << 24) + ((long long) a2 << 16) + ((long long) a1 << 8) + a0 + ((long long) 32896 << P);
                return f.s == kBarIFinNone ? __builtin_nanf("") : (float) (__builtin_ldexp((double) tot, -P) / 65535.0);
            }
        };
        auto out_of = [](uint32_t x) -> OutT { if constexpr (R16) return (uint16_t) x; else return __builtin_bit_cast(float, x); };
        // all results first (they take the place of the accumulators they come from) ...
        uint32_t res[G][4];                                                    // (a float's bits, or the texel)
#pragma unroll
        for (uint32_t g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (R16) res[g][r] = result(g, r);
                else res[g][r] = __builtin_bit_cast(uint32_t, result(g, r));
            }
        // ... pinned in front of the wait (sixteen at a time: an asm statement takes thirty operands) ...
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            uint32_t (&x)[4] = res[g];
            asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
        }
        // ... then every load this wave has in flight -- the weight stream's next PF steps ...
        __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0)
        asm volatile("" : "+v"(pre[0].d[0]), "+v"(pre[0].d[1]), "+v"(pre[0].d[2]), "+v"(pre[0].d[3]), "+v"(pre[1].d[0]), "+v"(pre[1].d[1]), "+v"(pre[1].d[2]), "+v"(pre[1].d[3]) : : "memory");
        fresh = PF;
        park_new();                                                             // (the ring's new texels: requested when the round opened)
        // ... then the stores, `global_store v_off, v_data, s[base:base+1]`: a uniform row base in scalar registers, walked from row to row by
        // scalar additions, + one 32-bit lane offset -- no vector address arithmetic per store.  (What it takes: the row OFFSET goes through an asm
        // statement so that the backend neither folds the lane offset into a 64-bit vector address it then walks with a v_lshl_add_u64 per store nor
        // keeps 64 hoisted row offsets in scalar registers -- the offset, not the pointer: a pointer that went through an asm statement loses its
        // address space and the stores become FLAT ones -- and the lane offset is re-defined opaquely in every basic block that stores: its
        // zero-extension must sit next to the store for the addressing mode to be matched.)
        // register r of a group is row 8 (r / 4) + r % 4 (+ 4 for the upper lanes: in the lane offset) of its 32
        uint32_t loff = (4u * (lane >> 4) * bars + kb) * (uint32_t) sizeof(OutT);
        const size_t one_row = (size_t) bars * sizeof(OutT);
        size_t ro = row0 * one_row;                                             // uniform: offset of the row the next store goes to
        // a partial last block: rows of the group this lane may store
        uint32_t rlim = R > 4u * (lane >> 4) ? R - 4u * (lane >> 4) : 0u;
        asm volatile("" : "+v"(rlim));
        if (kb < bars) {
            if (R == (uint32_t) RB) {                                           // whole block (uniform): no row checks, one basic block
                asm volatile("" : "+v"(loff));
#pragma unroll
                for (uint32_t g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        asm volatile("" : "+s"(ro));
                        st<OutT>(static_cast<char*>(bars_out) + ro, loff, out_of(res[g][i]));
                        ro += i == 3 ? 13 * one_row : one_row;
                    }
            } else {
#pragma unroll
                for (uint32_t g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        asm volatile("" : "+s"(ro));
                        if (16u * g + (uint32_t) i < rlim) {
                            uint32_t lo = loff;
                            asm volatile("" : "+v"(lo));
                            st<OutT>(static_cast<char*>(bars_out) + ro, lo, out_of(res[g][i]));
                        }
                        ro += i == 3 ? 13 * one_row : one_row;
                    }
            }
        }
        __syncthreads();
        ++t;
    };
    // The weight stream is requested and awaited by hand: behind a bank's three requests at least the other two banks' six have been issued
    // by the time the bank is used, and loads retire in order: vmcnt(6).  (Left to the compiler, every control-flow merge behind a tile's end
    // made the next wait a wait for everything, the fragments requested a moment ago included: an L2 round trip per tile.)  The compiler does
    // not know these loads are in flight: its own waits (for the ring's texels, the epilogue's constants) can only come out stricter than
    // necessary, never too lax.
    auto wload = [&](glv_i4v& d0, glv_i4v& d1, glv_i4v& d2) {
        asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:1024\n\tglobal_load_dwordx4 %2, %3, off offset:2048"
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2) : "v"(wp) : "memory");
        wp += 3 * 64;
    };
    auto step = [&](auto BC) {                                                  // one step of 32 bins on bank B
        constexpr int B = decltype(BC)::value;
        glv_i4v ah[G], al[G];
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            ah[g] = *reinterpret_cast<const glv_i4v*>(plane_h + g * 16u * PITCH + arow + ck * 16u);
            al[g] = *reinterpret_cast<const glv_i4v*>(plane_l + g * 16u * PITCH + arow + ck * 16u);
        }
        ck = ck + 4u >= S16 ? ck + 4u - S16 : ck + 4u;
        if (fresh != 0) --fresh;
        else asm volatile("s_waitcnt vmcnt(3)" : "+v"(wb[B][0]), "+v"(wb[B][1]), "+v"(wb[B][2]) : : "memory");
        const glv_i4v w0 = wb[B][0], w1 = wb[B][1], w2 = wb[B][2];
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            acc[g][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah[g], w2, acc[g][3], 0, 0, 0);
            acc[g][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah[g], w1, acc[g][2], 0, 0, 0);
            acc[g][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah[g], w0, acc[g][1], 0, 0, 0);
            acc[g][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(al[g], w0, acc[g][0], 0, 0, 0);
        }
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            acc[g][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(al[g], w2, acc[g][2], 0, 0, 0);
            acc[g][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(al[g], w1, acc[g][1], 0, 0, 0);
        }
        // the bank's next step, PF steps on: requested once its MFMAs have been issued (they read the registers when they issue)
        asm volatile("" : : "v"(acc[0][1]) : "memory");
        wload(wb[B][0], wb[B][1], wb[B][2]);
    };
    fnext = fin[Mn.k0 + (lane & 15u)];                                          // the first tile's epilogue constants
    asm volatile("" : "+v"(fnext.c), "+v"(fnext.s));
    if (!next_tile()) return;
    // the wave's first tile: fill the pipeline
    wp = wq + (uint32_t) __builtin_amdgcn_readfirstlane((int) M.w_off) + lane;
#pragma unroll
    for (int b = 0; b < PF; ++b) wload(wb[b][0], wb[b][1], wb[b][2]);
    for (;;) {
        step(std::integral_constant<int, 0>{});
        if (--left == 0) { close_tile(); if (!next_tile()) break; }
        step(std::integral_constant<int, 1>{});
        if (--left == 0) { close_tile(); if (!next_tile()) break; }
    }
#endif
}


constexpr uint32_t kTB = 16, kSB = 64;      // bars per tile, bins per step
static bool make_tables_x16(std::vector<BarMTile>& itiles, std::vector<int8_t>& wq, std::vector<BarIFin>& fin, std::vector<BarTile>& rounds,
                            const std::vector<BarDesc>& desc, const std::vector<float>& tap_w, uint32_t n, uint32_t bins, uint32_t tiles_per_round) {
    itiles.clear(); wq.clear(); fin.clear(); rounds.clear();
    const uint32_t bars = (uint32_t) desc.size(), nt = (bars + kTB - 1) / kTB;
    std::vector<std::vector<int32_t>> W(bars);
    fin.assign((size_t) nt * kTB, BarIFin{0u, kBarIFinNone});
    for (uint32_t k = 0; k < bars; ++k) {
        const int P = bar_int_weights(tap_w.data() + desc[k].tap_offset, desc[k].count, W[k]);
        if (P == -2) return false;
        if (P < 0) continue;
        const uint32_t s = (uint32_t) P - 16u;
        fin[k] = BarIFin{(32896u << s) + (1u << (s - 1u)), s};
    }
    auto tile_end = [&](uint32_t H) {
        uint32_t e = 0;
        const uint32_t k1 = (H + 1) * kTB < bars ? (H + 1) * kTB : bars;
        for (uint32_t k = H * kTB; k < k1; ++k) e = desc[k].first_bin + desc[k].count > e ? desc[k].first_bin + desc[k].count : e;
        return (e + 7u) & ~7u;
    };
    bool monotone = true;
    for (uint32_t T = 0; T < nt; ++T) {
        const uint32_t k0 = T * kTB, k1 = k0 + kTB < bars ? k0 + kTB : bars;
        uint32_t lo = 0xffffffffu;
        for (uint32_t k = k0; k < k1; ++k) lo = desc[k].first_bin < lo ? desc[k].first_bin : lo;
        BarMTile t{k0, lo & ~15u, 0u, 0u};
        t.steps = (tile_end(T) - t.origin + kSB - 1u) / kSB;
        if (T && t.origin < itiles[T - 1].origin) monotone = false;
        itiles.push_back(t);
    }
    // (a step reads 64 bins from the ring: the last step of a tile may reach up to 63 bins past the tile's own end -- zero weights there -- so the ring
    // must hold origin .. origin + 64 steps: the round cutter is given that as the tile's end)
    auto ring_end = [&](uint32_t H) { const uint32_t e = itiles[H].origin + itiles[H].steps * kSB; return e < tile_end(H) ? tile_end(H) : (e > n ? ((n + 7u) & ~7u) : e); };
    if (monotone) cut_bar_rounds(rounds, nt, [&](uint32_t T) { return itiles[T].origin; }, ring_end, n, bins, tiles_per_round);
    if (rounds.empty()) return false;
    std::vector<uint32_t> order;
    for (uint32_t wv = 0; wv < tiles_per_round; ++wv)
        for (const BarTile& r : rounds)
            if (r.k0 + wv < r.k1) order.push_back(r.k0 + wv);
    for (uint32_t Ti : order) {
        BarMTile& t = itiles[Ti];
        const uint32_t k0 = t.k0, k1 = k0 + kTB < bars ? k0 + kTB : bars;
        t.w_off = (uint32_t) (wq.size() / 16u);
        wq.resize(wq.size() + (size_t) t.steps * 3u * 64u * 16u, 0);
        int8_t* base = wq.data() + (size_t) t.w_off * 16u;
        for (uint32_t k = k0; k < k1; ++k)
            for (uint32_t i = 0; i < desc[k].count; ++i) {
                const uint32_t rel = desc[k].first_bin + i - t.origin, step = rel / kSB, quarter = (rel % kSB) / 16u, j = rel % 16u;
                int8_t d[3];
                bar_int_digits(W[k][i], d);
                for (uint32_t q = 0; q < 3; ++q) base[(((size_t) step * 3u + q) * 64u + quarter * 16u + (k - k0)) * 16u + j] = d[q];
            }
    }
    wq.resize(wq.size() + (size_t) kBarILookAhead * 3u * 64u * 16u, 0);
    return true;
}
}  // namespace glv

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int S> static hipError_t launch_x16(const uint16_t* rows_in, uint16_t* out, size_t nrows, uint32_t n, const glv::BarIRowsTables& rt) {
    using namespace glv;
    const size_t lds = (size_t) 2 * 64 * (S + 16);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(glv_bars_rows_i8x16_kernel<S, 64, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (e != hipSuccess) return e;
    const uint32_t xb = (uint32_t) ((nrows + 63) / 64);
    hipLaunchKernelGGL((glv_bars_rows_i8x16_kernel<S, 64, false, true>), dim3(xb, 1), dim3(64 * kRowsWaves), lds, nullptr, rows_in, out, nrows, n, n, rt.rounds, rt.nrounds, rt.nrounds,
                       rt.tiles, reinterpret_cast<const glv_i4v*>(rt.wq), rt.fin);
    return hipGetLastError();
}

int main(int argc, char** argv) {
    using namespace glv;
    const uint32_t n = argc > 1 ? (uint32_t) atoi(argv[1]) : 4096u;
    const size_t rows = argc > 2 ? (size_t) atol(argv[2]) : (size_t) 32768 * 4096 / n;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    std::vector<BarDesc> desc; std::vector<float> w;
    make_bar_taps(desc, w, n, n, 0.025f, 0.5f);
    std::vector<BarMTile> it; std::vector<BarTile> rounds; std::vector<int8_t> wq; std::vector<BarIFin> fin;
    uint32_t bins = 0;
    for (uint32_t b : {288u, 448u}) { if (make_tables_x16(it, wq, fin, rounds, desc, w, n, b, 4u)) { bins = b; break; } }
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
    CK(hipMemset(d_out, 0xee, 2 * rows * n));
    auto launch = [&]() { return bins == 288 ? launch_x16<288>(d_tex, d_out, rows, n, rt) : launch_x16<448>(d_tex, d_out, rows, n, rt); };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(launch());
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int k = 0; k < 3; ++k) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) CK(launch());
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        best = ms < best ? ms : best; sum += ms;
    }
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
    int nb = 0; (void) hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(glv_bars_rows_i8x16_kernel<288, 64, false, true>), 256, (size_t) 2 * 64 * (288 + 16));
    printf("x16: n=%u rows=%zu ring=%u rounds=%zu steps/row-block=%zu (of 64 bins x 16 bars) wq=%zu KB workgroups/CU=%d: %.4f ms (best %.4f)  %zu of %zu checked texels differ\n", n, rows, bins, rounds.size(), steps,
           wq.size() / 1024, nb, sum / 3, best, bad, checked);
    return 0;
}
