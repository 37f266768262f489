// glv_kernel_tmpl.h -- the gfx950 (CDNA4, wave64) frame kernel template and its launcher.
//
//   glv_frame_kernel   PCM (or planar f32) -> unpack -> window -> nn-point FFT -> abs/log/tilt
//                      -> gravity -> average in ONE launch; spectra never leave registers/LDS
//                      between stages.  Replaces transform_fft/gravity/average
//                      (glava/render.c:720-847) and the unpack loop of glava/fifo.c:94-110.
//
// Execution shape: T = nn/E lanes cooperate on one FFT ("slot"), SLOTS slots per workgroup;
// every lane owns E = 8/16/32 complex points and performs up to log2(E) radix-2 stages on them in registers
// between LDS exchanges (ds_write2_b64 / ds_read2_b64 / ds_read_b128; the pass-0 layout is padded by one
// point per E).  HBM traffic is the algorithmic minimum: PCM in (8 B per lane per load, lanes contiguous),
// spectra out (16 B per lane, lanes contiguous), state in/out.  Workgroups are persistent over a
// grid-stride list of frames so per-lane twiddles and tilt factors can stay in VGPRs.
//
// No MFMA: the path is a bandwidth/LDS/VALU problem, not a dense contraction.
//
// Tuning knobs (template parameters; production picks one set per size in glv_inst.hip,
// tools/tune.py sweeps them through glv_tune.hip):
//   LOG_E   log2 of the points a lane owns (4: E=16, T=nn/16 lanes per row; 3: E=8, T=nn/8; 5: E=32, T=nn/32)
//   SLOTS   FFT slots per workgroup
//   NBUF    1: one LDS exchange region per slot, two barriers per exchange
//           2: ping-pong regions, one barrier per exchange
//           0: SPLIT exchange -- one half-size region per slot, the row crosses it one float component at a time (real
//              parts, then imaginary parts: glv_frame.h exchange_write_half), four barriers per exchange.  Halves the
//              LDS a row in flight needs, which is what lets the sizes whose tables do not fit beside two full
//              regions (N=16384: 65 KiB of twiddles) keep them in LDS (TWREG 4) with two rows per CU
//   TWREG   1 (true): passes >= 1 keep their per-lane twiddles in VGPRs across frames
//           0 (false): gathered from the (L2-resident) table at every pass
//           2: middle passes gather from an LDS copy of their table range (entries E-1 .. L0_last-2: 8 KiB
//              for N=16384 at 32 points per lane), the last pass from the L2-resident table
//           3: middle passes from LDS, the last pass resident in VGPRs
//           4: every pass >= 1 from an LDS copy of the table (nn - E entries: 16 KiB at N=4096) -- no twiddle
//              VGPRs; allows 3 waves/SIMD at N=4096 (6 slots), measured slower than 2 (tools/tune.py)
//   WINLDS  true: window table staged once per workgroup into LDS; false: read through L1/L2
//   OCC     __launch_bounds__ minimum waves per SIMD (caps the VGPR budget: 512 / OCC)
//   TILTREG   0: tilt factors re-read from the table per row; 1: the 2E per-lane factors stay in VGPRs across
//             rows; 2: evaluated in registers with the reference's float operations (no memory, no registers)
//             3: log_mode 1: the folded factor tilt * ln2/3 from a per-lane base term and ONE fused multiply-add + max per value
//                (glv_core.h tilt_lin: <= 4.5e-7 from the reference's own rounded factor, contract 1e-5); other log modes as 2
//   PREFETCH  0: no software pipeline (load, transform, store per row)
//           1: in-place pipeline -- the next frame's (s16, interleaved f32) or row's (planar f32) samples
//              are requested before the current row's passes and unpacked/windowed after its epilogue,
//              straight into the point registers.  (Earlier variants that unpacked into a second point
//              set before the epilogue, per frame or per row, measured slower at every size and are gone;
//              profiles/tune_r01.txt still lists them as pf=1 / pf=2, this one as pf=3.)
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>

#include "glv_frame.h"

namespace glv {

// ---- slot-scoped synchronisation --------------------------------------------------------------------
// The slots of a workgroup transform independent rows; nothing but the read-only tables is shared between
// them, so an exchange only has to order the waves of ONE slot:
//   * T <= 64: the slot lives inside one wave and LDS executes a wave's instructions in order -- no
//     hardware barrier at all, only a compiler fence;
//   * one slot per workgroup: s_barrier is already slot-scoped;
//   * otherwise (N=4096: two waves per slot, two slots): a counter in LDS per slot.  Lane 0 of every wave
//     adds 1 (ordered behind the wave's own LDS writes), then the wave polls until all T/64 waves of the
//     slot arrived.  A workgroup-wide s_barrier here would keep the two slots in lockstep -- the same
//     coupling that makes 4 slots per workgroup slower than 2.
template <int T, int SLOTS>
struct SlotSync {
    static constexpr bool WAVE_LOCAL = T <= 64 && (64 % T) == 0;
    static constexpr bool WORKGROUP = !WAVE_LOCAL && SLOTS == 1;
    static constexpr bool COUNTER = !WAVE_LOCAL && !WORKGROUP;
    static constexpr uint32_t WAVES = T / 64 > 0 ? T / 64 : 1;
    uint32_t* counter = nullptr;     // LDS, one per slot (COUNTER mode)
    uint32_t expected = 0;
    __device__ __forceinline__ void sync() {
        if constexpr (WAVE_LOCAL) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (WORKGROUP) {
            __syncthreads();
        } else {
            expected += WAVES;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < expected) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
};

// UNIT_SHORTCUT: pass 0 evaluates its (1, +0) twiddles as a +- b (s16 input only, glv_core.h SubPass::run)
template <int LOG_NN, int LOG_E, int NBUF, int TWREG, bool UNIT_SHORTCUT = true>
struct Body {
    using FR = Frame<LOG_NN, LOG_E>;
    static constexpr int P = FR::P, NN = FR::NN, N = FR::N, T = FR::T;

    // where the per-lane twiddles of pass Q (>= 1) come from
    static constexpr bool TW_LDS_MODE = (TWREG == 4 && P >= 2) || (TWREG >= 2 && P >= 3);
    static constexpr bool is_mid(int q) { return q >= 1 && q < P - 1; }
    static constexpr bool from_lds(int q) { return TW_LDS_MODE && (is_mid(q) || (TWREG == 4 && q >= 1)); }
    static constexpr bool resident(int q) { return q >= 1 && (TWREG == 1 || (TWREG >= 2 && !TW_LDS_MODE) || (TWREG == 3 && q == P - 1)); }
    // LDS copy: table entries [LDS_BIAS, LDS_BIAS + LDS_ENTRIES) = the stages of passes 1..P-2 (TWREG 2, 3)
    // or of every pass >= 1 (TWREG 4: the whole table but pass 0's E-1 entries)
    static constexpr int LDS_BIAS = tw_offset(FR::E);
    static constexpr int LDS_ENTRIES = !TW_LDS_MODE ? 0 : TWREG == 4 ? NN - FR::E : (1 << FR::PL::log_l0(P - 1)) - FR::E;

    template <int PASS>
    static constexpr int tw_off() {
        if constexpr (PASS == 0) return 0;
        else return tw_off<PASS - 1>() + FR::template PassInfo<PASS - 1>::NTW;
    }
    static constexpr int TW_TOTAL = tw_off<P>();

    template <int PASS>
    static __device__ __forceinline__ cf (&tw_ref(cf* all))[FR::template PassInfo<PASS>::NTW] {
        return *reinterpret_cast<cf(*)[FR::template PassInfo<PASS>::NTW]>(all + tw_off<PASS>());
    }

    // pass-0 twiddles: wave-uniform by construction; readfirstlane pins them into SGPRs
    static __device__ __forceinline__ void gather_uniform_tw0(cf* all, const cf* __restrict__ table) {
        cf (&t0)[FR::template PassInfo<0>::NTW] = tw_ref<0>(all);
        FR::template gather_tw<0>(t0, table, 0);
#pragma unroll
        for (int i = 0; i < FR::template PassInfo<0>::NTW; ++i) {
            t0[i].x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t0[i].x)));
            t0[i].y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t0[i].y)));
        }
    }

    // once per kernel: the passes whose twiddles stay in VGPRs
    template <int PASS>
    static __device__ __forceinline__ void gather_resident(cf* all, const cf* __restrict__ table, int tid) {
        if constexpr (resident(PASS)) FR::template gather_tw<PASS>(tw_ref<PASS>(all), table, tid);
        if constexpr (PASS + 1 < P) gather_resident<PASS + 1>(all, table, tid);
    }
    // per row: the twiddles of a non-resident pass, from the LDS copy or the L2-resident table
    template <int PASS>
    static __device__ __forceinline__ void gather_transient(cf* all, const cf* __restrict__ table, const cf* lds_tw, int tid) {
        if constexpr (from_lds(PASS)) FR::template gather_tw<PASS, LDS_BIAS>(tw_ref<PASS>(all), lds_tw, tid);
        else if constexpr (!resident(PASS)) FR::template gather_tw<PASS>(tw_ref<PASS>(all), table, tid);
    }

    // passes PASS..P-1 on the 16 register-resident points; `xcount` counts exchanges so the
    // ping-pong region alternates consistently across channels and frames.  Scheduling fences
    // pin the phase order  compute | LDS write + next twiddle gather | barrier | LDS read | compute
    // so that the backend cannot pull the table loads of later passes (30 VGPRs each) up front.
    template <int PASS, typename SYNC>
    static __device__ __forceinline__ void run(cf (&v)[FR::E], cf* tw_all, const cf* __restrict__ table,
                                               char* xslot, int tid, unsigned& xcount, const cf* lds_tw, SYNC& sy) {
        // pass 0's twiddles are the same for every lane (k0 = 0); the kernel gathers them once, before
        // the row loop, into scalar registers (gather_uniform_tw0) -- a vector load here would sit in
        // front of every row's first butterfly AND, vmcnt being in-order, behind the PCM prefetch.
        FR::template compute<PASS, UNIT_SHORTCUT>(v, tw_ref<PASS>(tw_all));
        if constexpr (PASS + 1 < P) {
            char* xb = xslot + (NBUF == 2 ? (size_t) (xcount & 1u) * FR::XREGION * sizeof(cf) : 0);
            GLV_SCHED_FENCE();
            if constexpr (NBUF == 0) {
                // split exchange: real parts, then imaginary parts, through the one half-size region
                sy.sync();                                  // previous readers of the region are done
                FR::template exchange_write_half<PASS, 0>(xb, v, tid);
                gather_transient<PASS + 1>(tw_all, table, lds_tw, tid);
                sy.sync();
                FR::template exchange_read_half<PASS + 1, 0>(v, xb, tid);
                GLV_SCHED_FENCE();
                sy.sync();                                  // every real part has been read
                FR::template exchange_write_half<PASS, 1>(xb, v, tid);
                sy.sync();
                FR::template exchange_read_half<PASS + 1, 1>(v, xb, tid);
            } else {
            if constexpr (NBUF == 1) sy.sync();         // previous readers of the region are done
            FR::template exchange_write<PASS>(xb, v, tid);
            // the next pass's per-lane twiddles travel from L2 while the exchange settles
            gather_transient<PASS + 1>(tw_all, table, lds_tw, tid);
            sy.sync();
            FR::template exchange_read<PASS + 1>(v, xb, tid);
            }
            GLV_SCHED_FENCE();
            ++xcount;
            run<PASS + 1>(v, tw_all, table, xslot, tid, xcount, lds_tw, sy);
        }
    }
};

// dynamic LDS of one workgroup: exchange regions | window (WINLDS) | log table | middle-pass twiddles (TWREG >= 2) |
// one 16-byte cell per slot for the slot barrier counter
template <int LOG_NN, int LOG_E, int SLOTS, int NBUF, bool WINLDS, int TWREG = 0>
constexpr size_t frame_lds_bytes() {
    return (size_t) SLOTS * (NBUF == 0 ? sizeof(float) : NBUF * sizeof(cf)) * Frame<LOG_NN, LOG_E>::XREGION + (WINLDS ? (size_t) Frame<LOG_NN, LOG_E>::N * sizeof(double) : 0)
           + ((size_t) sizeof(LogEntry) << log_tab_bits_of(LOG_NN)) + (size_t) Body<LOG_NN, LOG_E, NBUF, TWREG>::LDS_ENTRIES * sizeof(cf) + (size_t) 16 * SLOTS;
}

// wave-uniform value -> SGPR (valid when all lanes of the wave hold the same value)
template <bool UNIFORM>
static __device__ __forceinline__ uint32_t maybe_scalar(uint32_t v) {
    if constexpr (UNIFORM) return (uint32_t) __builtin_amdgcn_readfirstlane((int) v);
    else return v;
}

template <int LOG_NN, int IN_MODE, int LOG_MODE, int SLOTS, int NBUF, int TWREG, bool WINLDS, int OCC, int PREFETCH, int TILTREG,
          int LOG_E, int STATEFUL, int WPRE = 0>
__global__ void __launch_bounds__((Frame<LOG_NN, LOG_E>::T * SLOTS), OCC)
glv_frame_kernel(const FrameArgs a) {
    using FR = Frame<LOG_NN, LOG_E>;
    constexpr int E = FR::E;
    constexpr int T = FR::T, N = FR::N;
    constexpr bool RING = IN_MODE == IN_S16_RING;
    constexpr bool S16 = IN_MODE == IN_S16_STEREO || RING;
    constexpr bool WSPLIT = S16 && win_split_of(LOG_NN, STATEFUL);     // s16 samples: the window product without fp64 (glv_core.h apply_window_split)
    // f32 rows may hold -0.0, Inf and NaN: no unit-twiddle shortcut, non-finite values through the bit-faithful log
    constexpr bool NF = !S16;
    using BD = Body<LOG_NN, LOG_E, NBUF, TWREG, S16>;
    constexpr bool WAVE_SLOT = (T % 64) == 0;      // a wave never straddles two slots
    constexpr size_t XBYTES = (size_t) FR::XREGION * (NBUF == 0 ? sizeof(float) : sizeof(cf));     // NBUF 0: split exchange, half-size region
    constexpr int NREG = NBUF == 0 ? 1 : NBUF;                                                         // regions per slot

    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHREADS = T * SLOTS;
    const uint32_t slot = maybe_scalar<WAVE_SLOT>(threadIdx.x / T);
    const int tid = (int) (threadIdx.x % T);
    char* xslot = smem + (size_t) slot * NREG * XBYTES;

    const void* gwin = WSPLIT ? a.win_split : static_cast<const void*>(a.win);     // 16 bytes per complex point either way
    const void* win = gwin;
    if constexpr (WINLDS) {
        char* lwin = smem + (size_t) SLOTS * NREG * XBYTES;
        for (int i = threadIdx.x; i < N / 2; i += NTHREADS) st<d2>(lwin, (uint32_t) i * 16u, ld<d2>(gwin, (uint32_t) i * 16u));
        __syncthreads();
        win = lwin;
    }

    // log_mode 0: the 256- or 512-entry (1/c, log(c)/3) table is gathered per value; LDS serves such random
    // 16-byte reads without touching the vector-memory path the PCM/spectrum streams use.
    const LogEntry* logtab = a.logtab;
    if constexpr (LOG_MODE == 0) {
        char* llog = smem + (size_t) SLOTS * NREG * XBYTES + (WINLDS ? (size_t) N * sizeof(double) : 0);
        constexpr int LB = log_tab_bits_of(LOG_NN);             // entries this size uses: every 2^(9 - LB)-th of the table in HBM
        for (int i = threadIdx.x; i < (1 << LB); i += NTHREADS) st<d2>(llog, (uint32_t) i * 16u, ld<d2>(a.logtab, (uint32_t) (i << (kLogTabMaxBits - LB)) * 16u));
        __syncthreads();
        logtab = reinterpret_cast<const LogEntry*>(llog);
    }

    // TWREG >= 2: the twiddle table range of the middle passes is staged into LDS once per workgroup
    const cf* lds_tw = nullptr;
    if constexpr (BD::TW_LDS_MODE) {
        char* ltw = smem + (size_t) SLOTS * NREG * XBYTES + (WINLDS ? (size_t) N * sizeof(double) : 0) + ((size_t) sizeof(LogEntry) << log_tab_bits_of(LOG_NN));
        for (int i = threadIdx.x; i < BD::LDS_ENTRIES; i += NTHREADS) st<cf>(ltw, (uint32_t) i * 8u, ld<cf>(a.tw, (uint32_t) (BD::LDS_BIAS + i) * 8u));
        __syncthreads();
        lds_tw = reinterpret_cast<const cf*>(ltw);
    }

    // slot-scoped synchronisation state (a counter per slot at the end of the LDS allocation)
    SlotSync<T, SLOTS> sy;
    if constexpr (SlotSync<T, SLOTS>::COUNTER) {
        uint32_t* ctr = reinterpret_cast<uint32_t*>(smem + frame_lds_bytes<LOG_NN, LOG_E, SLOTS, NBUF, WINLDS, TWREG>() - 16 * SLOTS) + 4 * slot;
        if (tid == 0) { ctr[0] = 0; ctr[1] = 0; ctr[2] = 0; }
        __syncthreads();
        sy.counter = ctr;
    }

    cf tw_all[BD::TW_TOTAL];
    BD::gather_uniform_tw0(tw_all, a.tw);
    if constexpr (FR::P > 1) BD::template gather_resident<1>(tw_all, a.tw, tid);
    cf tilt_reg[TILTREG == 1 ? E : 1];
    if constexpr (TILTREG == 1) FR::gather_tilt(tilt_reg, a.tilt, tid);
    if constexpr (TILTREG == 3) {
        // the lane part of every output index of the last pass: out_index(tid, gi, r) - out_index(0, gi, r) (the same for all gi, r)
        const int lane_q = FR::template out_index<FR::P - 1>(tid, 0, 0) - FR::template out_index<FR::P - 1>(0, 0, 0);
        tilt_reg[0].x = tilt_lin_base(tilt_lin(a.inv_n, a.fft_scale, a.one_minus_cutoff), 2 * lane_q);
        tilt_reg[0].y = 0.0f;
    }

    // operator chain, uniform for the launch.  Stateless and stateful chains are separate kernels
    // (STATEFUL): the history loads of gravity/average need ~60 more VGPRs in the epilogue, and having
    // them in the same kernel costs the plain FFT+magnitude pass 15-30 % (register allocation is per
    // kernel, not per path).
    const bool raw_out = (a.ops & OP_RAW) != 0;
    // STATEFUL 0: no state (FFT + magnitude only)   1: gravity / average   2: gravity / average with the
    // bars computed in the kernel (fused GLV_OP_BARS): the finished row is written to the slot's LDS
    // exchange region (idle between a row's last exchange and the next row's first) instead of HBM.
    // 3: no state, output as GL_R16 texels (GLV_OP_R16: uint16 [units][n], 8N instead of 12N bytes per frame)
    // 4: gravity / average with the output as GL_R16 texels (the state stays f32).  Separate kernels, not a run-time
    // branch: a second copy of the epilogue in the stateful kernel cost N=16384 176 more bytes of scratch per lane.
    // 5: the GL_R16 chain (glv_params.gl_storage == 1): upload quantisation, GL_MAX + gravity pass, ring, average pass on uint16
    // state (glv_frame.h epilogue_gl16); output GL_R16 texels or their floats (a.ops & OP_R16, uniform).  6: the same with the
    // bars computed in the kernel (the finished row's floats go to the slot's LDS region), bars as floats or texels (a.bars_r16)
    // 7: class 5 for GLV_OP_BARS_ONLY batches (a.live_points != 0): state, average and output for the row's live blocks only (glv_frame.h epilogue_gl16 LIVE)
    // 8 / 9: classes 2 / 6 (bars fused) for GLV_OP_BARS_ONLY batches: magnitude, state and the row in LDS for the live blocks only
    constexpr bool FUSED_BARS = STATEFUL == 2 || STATEFUL == 6 || STATEFUL == 8 || STATEFUL == 9;
    constexpr bool GL16 = STATEFUL == 5 || STATEFUL == 6 || STATEFUL == 7 || STATEFUL == 9;
    constexpr bool GL16_LIVE = STATEFUL == 7 || STATEFUL == 9;
    constexpr bool F32_LIVE = STATEFUL == 8;
    constexpr bool HAS_STATE = STATEFUL == 1 || STATEFUL == 2 || STATEFUL == 4 || STATEFUL == 8 || GL16;
    static_assert(!FUSED_BARS || WAVE_SLOT, "fused bars need whole waves per row");
    static_assert(!GL16 || LOG_MODE != 2, "the GL_R16 chain is built for log modes 0 and 1");
    static_assert(!FUSED_BARS || NBUF == 1, "fused bars park the finished row in exchange region 0: needs the full-size, two-barrier region");
    // class 7 in the s16 frame pipeline: the row's old state is requested before its transform (glv_frame.h gl16_state_prefetch)
    constexpr bool LIVE_PRE = GL16_LIVE && FR::LIVE_PREFETCH && S16 && PREFETCH == 1;
    typename FR::LivePre live_pre;
    const typename FR::LivePre* live_pre_ptr = nullptr;
    auto finish = [&](const cf (&v)[E], size_t row, int tid) {
        // a.out == nullptr (gravity without average only): the spectra ARE the gravity state
        // (render.c:733-734 stores the same value to both), so the second copy is not written
        float* out_row = FUSED_BARS ? reinterpret_cast<float*>(xslot)
                                               : (HAS_STATE && a.out == nullptr ? nullptr : a.out + row * N);
        if constexpr (GL16) {
            float* o = FUSED_BARS ? reinterpret_cast<float*>(xslot)
                                  : ((a.ops & OP_R16) ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(a.out) + row * N) : a.out + row * N);
            FR::template epilogue_gl16<LOG_MODE, TILTREG, NF, FUSED_BARS, GL16_LIVE>(v, o, row, tid, a, logtab, tilt_reg, live_pre_ptr);
        } else if constexpr (STATEFUL == 4) {
            float* out16 = reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(a.out) + row * N);
            if (raw_out) FR::template epilogue<LOG_MODE, EPI_RAW_STATE, 0, true, NF>(v, out16, row, tid, a, logtab);
            else FR::template epilogue<LOG_MODE, EPI_MAG_STATE, TILTREG, true, NF>(v, out16, row, tid, a, logtab, tilt_reg);
        } else if constexpr (F32_LIVE) {
            FR::template epilogue<LOG_MODE, EPI_MAG_STATE, TILTREG, false, NF, true>(v, out_row, row, tid, a, logtab, tilt_reg);
        } else if constexpr (HAS_STATE) {
            if (raw_out) FR::template epilogue<LOG_MODE, EPI_RAW_STATE, 0, false, NF>(v, out_row, row, tid, a, logtab);
            else FR::template epilogue<LOG_MODE, EPI_MAG_STATE, TILTREG, false, NF>(v, out_row, row, tid, a, logtab, tilt_reg);
        } else if constexpr (STATEFUL == 3) {
            float* out16 = reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(a.out) + row * N);
            if (raw_out) FR::template epilogue<LOG_MODE, EPI_RAW, 0, true, NF>(v, out16, row, tid, a, logtab);
            else FR::template epilogue<LOG_MODE, EPI_MAG, TILTREG, true, NF>(v, out16, row, tid, a, logtab, tilt_reg);
        } else {
            if (raw_out) FR::template epilogue<LOG_MODE, EPI_RAW, 0, false, NF>(v, out_row, row, tid, a, logtab);
            else FR::template epilogue<LOG_MODE, EPI_MAG, TILTREG, false, NF>(v, out_row, row, tid, a, logtab, tilt_reg);
        }
    };
    // FUSED_BARS: lane k of a slot stores bar k (and k + T, ... when bars > T): the first one's weight sum stays in a register
    float bar_wsum = 1.0f;
    if constexpr (FUSED_BARS) { if ((uint32_t) tid < a.bars) bar_wsum = a.bar_desc[tid].weight_sum; }
    // the epilogue of one row; every lane of the workgroup calls it (barriers inside when FUSED_BARS)
    auto finish_row = [&](const cf (&v)[E], size_t row, int tid, bool active) {
        if constexpr (FUSED_BARS) {
            sy.sync();                                     // every reader of the row's last exchange is done
            if (active) finish(v, row, tid);               // finished row -> LDS, natural order
            sy.sync();
            // T/GL groups of GL = 2 / 4 / 8 lanes work through the row's bar chunks (glv_frame.h "GLV_OP_BARS
            // arithmetic"; work lists from make_bar_items).  BB = bar_batch_of(LOG_NN) steps at a time: their loads (LDS
            // row + L2-resident weights) are issued together and the next batch's items are fetched
            // while the current one is reduced.  No global store inside the loop (vmcnt is one in-order
            // counter): every step stores its running total to the slack behind the row -- slot `res` is the
            // bar a chunk completes, or the dump slot lres[bars] -- and after a barrier the slot's lanes
            // divide by the weight sums and store the bars coalesced.
            constexpr int GL = bar_lanes_of((uint32_t) N);                    // lanes per group: 2 / 4 / 8
            constexpr uint32_t G = T / GL;
            constexpr int BB = bar_batch_of(LOG_NN);
            float* lrow = reinterpret_cast<float*>(xslot);
            float* lres = lrow + N;                                       // XREGION has NN/E points (2T floats) of slack: bars + 1 <= 2T (glv_api.cpp bar_fusable)
            static_assert(2 * T >= 64, "the chunk reads of bar_item_load stay inside the slot's region");
            if (active) {
                const int sub = tid & (GL - 1);
                const uint32_t g = (uint32_t) tid / GL;
                const BarItem* items = a.bar_items + g;
                BarItem it[BB];
#pragma unroll
                for (int b = 0; b < BB; ++b) it[b] = items[(size_t) b * G];
                float total = 0.0f;
                for (uint32_t s0 = 0; s0 < a.bar_nsteps; s0 += BB) {
                    BarTaps tp[BB];
                    BarItem nx[BB];
                    // (taps read from LDS step by step, only the weights held for the batch, would let ten steps fit the
                    // registers -- and ran 0.746 instead of 0.672 ms at six: ten LDS round trips in a row)
#pragma unroll
                    for (int b = 0; b < BB; ++b) tp[b] = bar_item_load<false>(lrow, a.bar_w, it[b], sub);   // slack: 2T >= 63 floats
#pragma unroll
                    for (int b = 0; b < BB; ++b) nx[b] = items[(size_t) (s0 + BB + b) * G];   // table has one batch of padding
                    // every load of the batch is ISSUED before the first sum: left to itself the backend sinks one step's four LDS reads to
                    // their uses -- four exposed LDS round trips (read, lgkmcnt(0), multiply) per batch of TWO steps (N = 4096 GL chain + 80 bars
                    // 1.645 -> 1.481 ms, N = 1024 1.318 -> 1.277, same box alternating: profiles/r05/uniform_branches.txt).  The six-step batches of
                    // N >= 16384 keep the backend's order: there one sunk step of six costs less than 96 registers of loads held at once
                    // (configs[2] 0.626 -> 0.643 ms with the fence on the same box)
                    if constexpr (BB <= kBarBatch) GLV_SCHED_FENCE();
#pragma unroll
                    for (int b = 0; b < BB; ++b) {
                        total = __builtin_fmaf(total, it[b].keep, group_sum<GL>(bar_item_lane_sum(tp[b])));
                        if (sub == 0) lres[it[b].res] = total;
                    }
#pragma unroll
                    for (int b = 0; b < BB; ++b) it[b] = nx[b];
                }
            }
            sy.sync();
            if (active) {
                if (GL16 && a.bars_r16) {                  // uniform: the bars as GL_R16 texels (the smooth pass's render target, render.c:2277-2303)
                    uint16_t* bo = reinterpret_cast<uint16_t*>(a.bars_out) + row * a.bars;
                    if ((uint32_t) tid < a.bars) bo[tid] = (uint16_t) unorm16(lres[tid] / bar_wsum);
                    for (uint32_t k = (uint32_t) tid + T; k < a.bars; k += T) bo[k] = (uint16_t) unorm16(lres[k] / a.bar_desc[k].weight_sum);
                } else {
                if ((uint32_t) tid < a.bars) a.bars_out[row * a.bars + (uint32_t) tid] = lres[tid] / bar_wsum;
                // more bars than lanes (N=1024: 64 lanes, 80 bars): a second trip, its weight sums from L2
                for (uint32_t k = (uint32_t) tid + T; k < a.bars; k += T) a.bars_out[row * a.bars + k] = lres[k] / a.bar_desc[k].weight_sum;
                }
            }
            // the next row's first exchange write is preceded by a barrier (NBUF == 1): the bars readers are safe
        } else {
            if (active) finish(v, row, tid);
        }
    };
    // row handled by this slot in the iteration that starts at `base` (idle slots clamp to the last
    // row and still take part in the barriers; they just do not store).  For s16 input row r is
    // channel r&1 of frame r>>1: the two channels of a frame sit in neighbouring slots.
    auto row_of = [&](uint32_t base) -> uint32_t {
        const uint32_t r = base + slot;
        return r < a.units ? r : a.units - 1;
    };

    unsigned xcount = 0;
    // A workgroup with a single slot takes both channel rows of a frame back to back (SEQ = 2), so
    // the frame's PCM is still fetched from HBM once and re-read from this CU's L1/L2.
    constexpr uint32_t SEQ = SLOTS == 1 ? 2 : 1, RPI = SLOTS * SEQ;      // rows per workgroup iteration
    const uint32_t stride = gridDim.x * RPI;
    const uint32_t nsteps = a.units == 0 ? 0 : ((a.units - 1) / RPI / gridDim.x + 1) * SEQ;   // upper bound, uniform
    auto step_base = [&](uint32_t step) -> uint32_t {            // first row of the workgroup's step-th row group
        return blockIdx.x * RPI + (step / SEQ) * stride + (step % SEQ) * SLOTS;
    };
    const int tid_outer = tid;
    if constexpr (S16 && PREFETCH == 1) {
        // In-place frame pipeline: one slot = one FRAME at a time, its two channel rows back to back, so
        // every frame's PCM is loaded exactly once by exactly one slot.  Per row r = 2*m + ch:
        //   A  ch == 1 only: issue the PCM loads of the slot's next frame into `raw` (both channels of
        //      the current frame were unpacked out of it already)        <- HBM latency starts here
        //   B  all FFT passes of row r                     (registers/LDS only: ~2 us of cover)
        //   W  s_waitcnt vmcnt(0): A's loads are a whole transform old, the previous row's stores older
        //   D  epilogue of row r: log/tilt/state + spectrum stores
        //   C  unpack + window row r+1 from `raw` straight into v (register-only: no vector-memory
        //      wait can land behind D's stores, W retired every load)
        // gfx9-class targets count loads and stores on one in-order counter (vmcnt) and a wait with
        // both kinds pending drains everything -- so a row must never need fresh load data right
        // after its predecessor's stores were issued; hence W before D and nothing but registers in C.
        const uint32_t nframes = a.units / 2;
        const uint32_t fstride = gridDim.x * SLOTS;
        const uint32_t nfs = nframes == 0 ? 0 : (nframes - 1) / fstride + 1;
        auto frame_of = [&](uint32_t m) -> uint32_t {
            const uint32_t f = blockIdx.x * SLOTS + m * fstride + slot;
            return f < nframes ? f : nframes - 1;
        };
        auto frame_ptr = [&](uint32_t f) -> const void* { return static_cast<const char*>(a.in) + (size_t) f * ((size_t) N * 4); };
        cf v[E];
        typename FR::Raw raw;
        if (blockIdx.x * SLOTS < nframes) {
            int tid = tid_outer;
            asm volatile("" : "+v"(tid));
            FR::template load_pcm<RING>(raw, frame_ptr(frame_of(0)), tid, a.rot);
            FR::template unpack_window<0, WSPLIT>(v, raw, win, tid, 0u, a.mono != 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (uint32_t r = 0; r < 2 * nfs; ++r) {
            const uint32_t m = r >> 1, ch = r & 1u;
            if (blockIdx.x * SLOTS + m * fstride >= nframes) break;                     // uniform for the workgroup
            int tid = tid_outer;
            asm volatile("" : "+v"(tid));
            const uint32_t fraw = blockIdx.x * SLOTS + m * fstride + slot;
            const bool active = fraw < nframes;
            const uint32_t f = frame_of(m);
            if (ch) FR::template load_pcm<RING>(raw, frame_ptr(frame_of(m + 1)), tid, a.rot);   // A
            if constexpr (LIVE_PRE) { FR::live_prefetch(live_pre, (size_t) f * 2 + ch, tid, a); live_pre_ptr = &live_pre; }   // A': the row's old state (W awaits it)
            GLV_SCHED_FENCE();
            BD::template run<0>(v, tw_all, a.tw, xslot, tid, xcount, lds_tw, sy);                            // B
            GLV_SCHED_FENCE();
            __builtin_amdgcn_s_waitcnt(0x0F70);                                                  // W
            // WPRE: the first window values of row r+1 are requested ahead of D's stores (glv_frame.h window_prefetch)
            typename FR::template WinPre<WPRE> wp;
            if constexpr (WPRE > 0) { FR::template window_prefetch<WPRE>(wp, win, tid); GLV_SCHED_FENCE(); }
            finish_row(v, (size_t) f * 2 + ch, tid, active);                                     // D
            GLV_SCHED_FENCE();
            FR::template unpack_window<WPRE, WSPLIT>(v, raw, win, tid, ch ^ 1u, a.mono != 0, wp.w);      // C
        }
        return;
    }
    if constexpr (IN_MODE == IN_F32_PLANAR && PREFETCH == 1 && LOG_E <= 4) {
        // planar f32 rows (the lb/rb snapshot): the in-place pipeline over ROWS -- A issue the loads of the
        // slot's next row, B transform, W, D epilogue, C window the fetched samples into the point registers
        auto row_ptr = [&](uint32_t row) -> const void* { return static_cast<const char*>(a.in) + (size_t) row * ((size_t) N * 4); };
        cf v[E];
        typename FR::RawF raw;
        if (step_base(0) < a.units) {
            int tid = tid_outer;
            asm volatile("" : "+v"(tid));
            FR::load_f32_raw(raw, row_ptr(row_of(step_base(0))), tid);
            FR::window_f32_raw(v, raw, win, tid);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (uint32_t step = 0; step < nsteps; ++step) {
            const uint32_t base = step_base(step);
            if (base >= a.units) break;                          // uniform for the workgroup
            int tid = tid_outer;
            asm volatile("" : "+v"(tid));
            const bool active = base + slot < a.units;
            const uint32_t row = row_of(base);
            const uint32_t nb = step_base(step + 1);
            const bool has_next = step + 1 < nsteps && nb < a.units;
            FR::load_f32_raw(raw, row_ptr(row_of(has_next ? nb : base)), tid);                   // A (unconditional)
            GLV_SCHED_FENCE();
            BD::template run<0>(v, tw_all, a.tw, xslot, tid, xcount, lds_tw, sy);                    // B
            GLV_SCHED_FENCE();
            __builtin_amdgcn_s_waitcnt(0x0F70);                                                  // W
            finish_row(v, (size_t) row, tid, active);                                            // D
            GLV_SCHED_FENCE();
            FR::window_f32_raw(v, raw, win, tid);                                                // C
        }
        return;
    }
    constexpr bool F32S = IN_MODE == IN_F32_STEREO || IN_MODE == IN_F32_RING;
    constexpr bool RINGF = IN_MODE == IN_F32_RING;
    if constexpr (F32S && PREFETCH == 1 && LOG_E <= 4) {
        // interleaved stereo f32 (PulseAudio), stereo only (the rare mono mix takes the generic loop below): one
        // slot = one frame, rows back to back, every row fetches its successor row's channel -- the sibling
        // channel of the same frame, then channel 0 of the slot's next frame
        if (a.mono == 0) {
            const uint32_t nframes = a.units / 2;
            const uint32_t fstride = gridDim.x * SLOTS;
            const uint32_t nfs = nframes == 0 ? 0 : (nframes - 1) / fstride + 1;
            auto frame_of = [&](uint32_t m) -> uint32_t {
                const uint32_t f = blockIdx.x * SLOTS + m * fstride + slot;
                return f < nframes ? f : nframes - 1;
            };
            auto frame_ptr = [&](uint32_t f) -> const void* { return static_cast<const char*>(a.in) + (size_t) f * ((size_t) N * 8); };
            cf v[E];
            typename FR::RawF raw;
            if (blockIdx.x * SLOTS < nframes) {
                int tid = tid_outer;
                asm volatile("" : "+v"(tid));
                FR::template load_f32s_raw<RINGF>(raw, frame_ptr(frame_of(0)), tid, 0u, a.rot);
                FR::window_f32_raw(v, raw, win, tid);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
            for (uint32_t r = 0; r < 2 * nfs; ++r) {
                const uint32_t m = r >> 1, ch = r & 1u;
                if (blockIdx.x * SLOTS + m * fstride >= nframes) break;                 // uniform for the workgroup
                int tid = tid_outer;
                asm volatile("" : "+v"(tid));
                const bool active = blockIdx.x * SLOTS + m * fstride + slot < nframes;
                const uint32_t f = frame_of(m);
                FR::template load_f32s_raw<RINGF>(raw, frame_ptr(frame_of(m + ch)), tid, ch ^ 1u, a.rot);   // A (unconditional)
                GLV_SCHED_FENCE();
                BD::template run<0>(v, tw_all, a.tw, xslot, tid, xcount, lds_tw, sy);                // B
                GLV_SCHED_FENCE();
                __builtin_amdgcn_s_waitcnt(0x0F70);                                              // W
                finish_row(v, (size_t) f * 2 + ch, tid, active);                                 // D
                GLV_SCHED_FENCE();
                FR::window_f32_raw(v, raw, win, tid);                                            // C
            }
            return;
        }
    }
    for (uint32_t step = 0; step < nsteps; ++step) {
        const uint32_t base = step_base(step);
        if (base >= a.units) break;                              // uniform for the workgroup
        // Re-define the lane id opaquely every iteration: window / twiddle table reads are loop
        // invariant, and LLVM would otherwise hoist ~120 VGPRs worth of them out of the frame loop
        // (and then spill them).  TWREG is the explicit, budgeted way to keep twiddles resident.
        int tid = tid_outer;
        asm volatile("" : "+v"(tid));
        const bool active = base + slot < a.units;
        const uint32_t row = row_of(base);
        cf v[E];
        if constexpr (S16) {
            // not pipelined: one slot = one channel row; the two channels of a frame sit in neighbouring
            // slots (the second reader of the frame's PCM hits L1/L2).  Measured faster than keeping the
            // samples in registers across two sequential rows of one slot.
            typename FR::Raw raw;
            FR::template load_pcm<RING>(raw, static_cast<const char*>(a.in) + (size_t) (row >> 1) * ((size_t) N * 4), tid, a.rot);
            GLV_SCHED_FENCE();
            FR::template unpack_window<0, WSPLIT>(v, raw, win, tid, row & 1u, a.mono != 0);
        } else if constexpr (F32S) {
            FR::template load_f32_stereo_window<RINGF>(v, static_cast<const char*>(a.in) + (size_t) (row >> 1) * ((size_t) N * 8), win, tid,
                                                       row & 1u, a.mono != 0, a.rot);
        } else {
            FR::load_f32_window(v, static_cast<const char*>(a.in) + (size_t) row * ((size_t) N * 4), win, tid);
        }
        BD::template run<0>(v, tw_all, a.tw, xslot, tid, xcount, lds_tw, sy);
        finish_row(v, (size_t) row, tid, active);
    }
}



template <int LOG_NN, int IN_MODE, int LOG_MODE, int SLOTS, int NBUF, int TWREG, bool WINLDS, int OCC, int PREFETCH, int TILTREG,
          int LOG_E = 4, int WPRE = 0, int WPRE_S = 0>
hipError_t launch_variant(const FrameArgs& a, int grid, hipStream_t st) {
    using FR = Frame<LOG_NN, LOG_E>;
    constexpr size_t lds = frame_lds_bytes<LOG_NN, LOG_E, SLOTS, NBUF, WINLDS, TWREG>();
    static_assert(lds <= 160 * 1024, "exchange regions + window exceed the 160 KiB LDS of a gfx950 CU");
    // the >64 KiB dynamic-LDS opt-in is a per-device function attribute: remember it per device
    // (one process per GPU is the deployment, but a host that drives several devices must work too)
    // (several host threads may drive several devices through the same instantiation -- glv_multi_*: the flags are
    // atomics; two threads racing on one device at worst both set the attribute, which is idempotent)
    struct AttrDone { std::atomic<bool> dev[64] = {}; };
    auto launch = [&](auto k, AttrDone& done, int threads = FR::T * SLOTS) -> hipError_t {
        if (lds > 64 * 1024) {
            int dev = 0;
            (void) hipGetDevice(&dev);
            if (dev < 0 || dev >= 64 || !done.dev[dev].load(std::memory_order_acquire)) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
                if (e != hipSuccess) return e;
                if (dev >= 0 && dev < 64) done.dev[dev].store(true, std::memory_order_release);
            }
        }
        if (grid <= 0) return hipSuccess;             // attribute only: glv_api.cpp batch_prepare readies the kernels a batch may launch, so
                                                      // that no process call changes a function attribute (a first call can be captured)
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, st, a);
        return hipGetLastError();
    };
    static AttrDone done_plain, done_state, done_r16, done_state_r16;   // per instantiation
    // the stateful epilogue needs the registers a resident last pass (TWREG 3) would occupy
    constexpr int TW_STATEFUL = TWREG == 3 ? 2 : TWREG;
    // the GL_R16 chain (gl_storage == 1: uint16 state): built for log modes 0 and 1
    if (a.gl_storage == 1 && (a.ops & (OP_GRAVITY | OP_AVERAGE))) {
        if constexpr (LOG_MODE != 2) {
            static AttrDone done_gl16, done_gl16_bars, done_gl16_live, done_gl16_bars_live;
            if (a.live_points != 0 && a.bars_out == nullptr)
                return launch(glv_frame_kernel<LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TW_STATEFUL, WINLDS, OCC, PREFETCH, TILTREG, LOG_E, 7, WPRE_S>, done_gl16_live);
            if (a.bars_out != nullptr) {
                if constexpr (FR::T % 64 == 0 && NBUF == 1) {
                    if (a.live_points != 0)
                        return launch(glv_frame_kernel<LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TW_STATEFUL, WINLDS, OCC, PREFETCH, TILTREG, LOG_E, 9, WPRE_S>, done_gl16_bars_live);
                    return launch(glv_frame_kernel<LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TW_STATEFUL, WINLDS, OCC, PREFETCH, TILTREG, LOG_E, 6, WPRE_S>, done_gl16_bars);
                } else return hipErrorInvalidValue;
            }
            return launch(glv_frame_kernel<LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TW_STATEFUL, WINLDS, OCC, PREFETCH, TILTREG, LOG_E, 5, WPRE_S>, done_gl16);
        } else return hipErrorInvalidValue;
    }
    if (a.bars_out != nullptr) {
        if constexpr (FR::T % 64 == 0 && NBUF == 1) {
            static AttrDone done_bars, done_bars_live;
            if (!(a.ops & (OP_GRAVITY | OP_AVERAGE))) return hipErrorInvalidValue;
            if (a.live_points != 0)
                return launch(glv_frame_kernel<LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TW_STATEFUL, WINLDS, OCC, PREFETCH, TILTREG, LOG_E, 8, WPRE_S>, done_bars_live);
            return launch(glv_frame_kernel<LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TW_STATEFUL, WINLDS, OCC, PREFETCH, TILTREG, LOG_E, 2, WPRE_S>, done_bars);
        } else return hipErrorInvalidValue;
    }
    if ((a.ops & (OP_GRAVITY | OP_AVERAGE)) && (a.ops & OP_R16))
        return launch(glv_frame_kernel<LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TW_STATEFUL, WINLDS, OCC, PREFETCH, TILTREG, LOG_E, 4, WPRE_S>, done_state_r16);
    if (a.ops & (OP_GRAVITY | OP_AVERAGE))
        return launch(glv_frame_kernel<LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TW_STATEFUL, WINLDS, OCC, PREFETCH, TILTREG, LOG_E, 1, WPRE_S>, done_state);
    if (a.ops & OP_R16)
        return launch(glv_frame_kernel<LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TWREG, WINLDS, OCC, PREFETCH, TILTREG, LOG_E, 3, WPRE>, done_r16);
    return launch(glv_frame_kernel<LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TWREG, WINLDS, OCC, PREFETCH, TILTREG, LOG_E, 0, WPRE>, done_plain);
}

}  // namespace glv
