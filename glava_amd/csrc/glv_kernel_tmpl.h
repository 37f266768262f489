// glv_kernel_tmpl.h -- the gfx950 (CDNA4, wave64) frame kernel template and its launcher.
//
//   glv_frame_kernel   PCM (or planar f32) -> unpack -> window -> nn-point FFT -> abs/log/tilt
//                      -> gravity -> average in ONE launch; spectra never leave registers/LDS
//                      between stages.  Replaces transform_fft/gravity/average
//                      (glava/render.c:720-847) and the unpack loop of glava/fifo.c:94-110.
//
// Execution shape: T = nn/16 lanes cooperate on one FFT ("slot"), SLOTS slots per
// workgroup; every lane owns 16 complex points and performs up to four radix-2 stages on them
// in registers between LDS exchanges (ds_write_b64 / ds_read_b64, XOR swizzle after the first
// pass).  HBM traffic is the algorithmic minimum: PCM in (8 B per lane per load, lanes
// contiguous), spectra out (8 B per lane, lanes contiguous), state in/out.  Workgroups are
// persistent over a grid-stride list of frames so per-lane twiddles can stay in VGPRs.
//
// No MFMA: the path is a bandwidth/LDS/VALU problem, not a dense contraction.
//
// Tuning knobs (template parameters; production picks one set per size in glv_inst.hip,
// tools/tune.py sweeps them through glv_tune.hip):
//   SLOTS   FFT slots per workgroup
//   NBUF    1: one LDS exchange region per slot, two barriers per exchange
//           2: ping-pong regions, one barrier per exchange
//   TWREG   true: passes >= 1 keep their per-lane twiddles in VGPRs across frames
//           false: gathered from the (L2-resident) table at every pass
//   WINLDS  true: window table staged once per workgroup into LDS; false: read through L1/L2
//   OCC     __launch_bounds__ minimum waves per SIMD (caps the VGPR budget: 512 / OCC)
#pragma once

#include <hip/hip_runtime.h>

#include "glv_frame.h"

namespace glv {

template <int LOG_NN, int NBUF, bool TWREG>
struct Body {
    using FR = Frame<LOG_NN>;
    static constexpr int P = FR::P, NN = FR::NN, N = FR::N, T = FR::T;

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

    template <int PASS>
    static __device__ __forceinline__ void gather_from(cf* all, const cf* __restrict__ table, int tid) {
        FR::template gather_tw<PASS>(tw_ref<PASS>(all), table, tid);
        if constexpr (PASS + 1 < P) gather_from<PASS + 1>(all, table, tid);
    }

    // passes PASS..P-1 on the 16 register-resident points; `xcount` counts exchanges so the
    // ping-pong region alternates consistently across channels and frames.
    template <int PASS>
    static __device__ __forceinline__ void run(cf (&v)[16], cf* tw_all, const cf* __restrict__ table,
                                               cf* xslot, int tid, unsigned& xcount) {
        // pass 0's twiddles are the same for every lane (k0 = 0): compile-time table offsets,
        // scalar loads; they are (re)gathered here so they never occupy VGPRs across frames.
        if constexpr (!TWREG || PASS == 0) FR::template gather_tw<PASS>(tw_ref<PASS>(tw_all), table, tid);
        FR::template compute<PASS>(v, tw_ref<PASS>(tw_all));
        if constexpr (PASS + 1 < P) {
            cf* xb = xslot + (NBUF == 2 ? (xcount & 1u) * NN : 0);
            if constexpr (NBUF == 1) __syncthreads();   // previous readers of the region are done
            FR::template exchange_write<PASS>(xb, v, tid);
            __syncthreads();
            FR::template exchange_read<PASS + 1>(v, xb, tid);
            ++xcount;
            run<PASS + 1>(v, tw_all, table, xslot, tid, xcount);
        }
    }
};

template <int LOG_NN, int IN_MODE, int LOG_MODE, int SLOTS, int NBUF, bool TWREG, bool WINLDS, int OCC>
__global__ void __launch_bounds__(Frame<LOG_NN>::T * SLOTS, OCC)
glv_frame_kernel(const FrameArgs a) {
    using FR = Frame<LOG_NN>;
    using BD = Body<LOG_NN, NBUF, TWREG>;
    constexpr int T = FR::T, N = FR::N, NN = FR::NN;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int slot = threadIdx.x / T;
    const int tid = threadIdx.x % T;
    cf* xslot = reinterpret_cast<cf*>(smem) + (size_t) slot * NBUF * NN;

    const double* win = a.win;
    if constexpr (WINLDS) {
        double* lwin = reinterpret_cast<double*>(smem + (size_t) SLOTS * NBUF * NN * sizeof(cf));
        for (int i = threadIdx.x; i < N / 2; i += T * SLOTS)
            reinterpret_cast<d2*>(lwin)[i] = reinterpret_cast<const d2*>(a.win)[i];
        __syncthreads();
        win = lwin;
    }

    cf tw_all[BD::TW_TOTAL];
    if constexpr (TWREG && FR::P > 1) BD::template gather_from<1>(tw_all, a.tw, tid);

    // operator chain, uniform for the launch
    const int epi = (a.ops & (OP_GRAVITY | OP_AVERAGE)) ? ((a.ops & OP_RAW) ? EPI_RAW_STATE : EPI_MAG_STATE)
                                                        : ((a.ops & OP_RAW) ? EPI_RAW : EPI_MAG);
    auto finish = [&](const cf (&v)[16], size_t row) {
        float* out_row = a.out + row * N;
        switch (epi) {
            case EPI_MAG:       FR::template epilogue<LOG_MODE, EPI_MAG>(v, out_row, row, tid, a); break;
            case EPI_MAG_STATE: FR::template epilogue<LOG_MODE, EPI_MAG_STATE>(v, out_row, row, tid, a); break;
            case EPI_RAW:       FR::template epilogue<LOG_MODE, EPI_RAW>(v, out_row, row, tid, a); break;
            default:            FR::template epilogue<LOG_MODE, EPI_RAW_STATE>(v, out_row, row, tid, a); break;
        }
    };

    unsigned xcount = 0;
    for (uint32_t base = blockIdx.x * SLOTS; base < a.units; base += gridDim.x * SLOTS) {
        const uint32_t unit = base + slot;
        const bool active = unit < a.units;      // idle slots still take part in the barriers
        const uint32_t u = active ? unit : a.units - 1;
        cf v[16];
        if constexpr (IN_MODE == IN_S16_STEREO) {
            typename FR::Pcm pcm;
            FR::load_pcm(pcm, static_cast<const int16_t*>(a.in) + (size_t) u * 2 * N, tid, a.rot);
            if (a.mono) FR::mono_mix(pcm);
            // channel 0 (left), then channel 1 (right), from the same 8-byte loads
            FR::template unpack_window<0>(v, pcm, win, tid);
            BD::template run<0>(v, tw_all, a.tw, xslot, tid, xcount);
            if (active) finish(v, (size_t) u * 2);
            FR::template unpack_window<1>(v, pcm, win, tid);
            BD::template run<0>(v, tw_all, a.tw, xslot, tid, xcount);
            if (active) finish(v, (size_t) u * 2 + 1);
        } else {
            FR::load_f32_window(v, static_cast<const float*>(a.in) + (size_t) u * N, win, tid);
            BD::template run<0>(v, tw_all, a.tw, xslot, tid, xcount);
            if (active) finish(v, (size_t) u);
        }
    }
}

template <int LOG_NN, int SLOTS, int NBUF, bool WINLDS>
constexpr size_t frame_lds_bytes() {
    return (size_t) SLOTS * NBUF * Frame<LOG_NN>::NN * sizeof(cf) + (WINLDS ? (size_t) Frame<LOG_NN>::N * sizeof(double) : 0);
}

template <int LOG_NN, int IN_MODE, int LOG_MODE, int SLOTS, int NBUF, bool TWREG, bool WINLDS, int OCC>
hipError_t launch_variant(const FrameArgs& a, int grid, hipStream_t st) {
    using FR = Frame<LOG_NN>;
    auto k = glv_frame_kernel<LOG_NN, IN_MODE, LOG_MODE, SLOTS, NBUF, TWREG, WINLDS, OCC>;
    constexpr size_t lds = frame_lds_bytes<LOG_NN, SLOTS, NBUF, WINLDS>();
    static_assert(lds <= 160 * 1024, "exchange regions + window exceed the 160 KiB LDS of a gfx950 CU");
    if (lds > 64 * 1024) {
        static bool attr_done = false;   // per instantiation
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(FR::T * SLOTS), lds, st, a);
    return hipGetLastError();
}

}  // namespace glv
