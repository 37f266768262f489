// glava_amd/csrc/glv_winsplit.h -- the device-side search for, and check of, the float-pair form of the window table
// (glv_core.h WinSplit / apply_window_split).  Included by glv_misc.hip (the product) and glv_tune.hip (the knob-sweep harness,
// so that its outputs stay bit-comparable with the product's).
#pragma once
#include <hip/hip_runtime.h>
#include "glv_core.h"

namespace glv {

// ---- the s16 window as float pairs (glv_core.h WinSplit / apply_window_split) -------------------------------------------
// One workgroup per window position i: hi = (float) w, lo = (float) (w - hi), then the identity
//   fma(x, hi, x * lo) == (float) ((double) x * w)   for x = k / 65535 (glv_core.h div_65535), k = 1 .. 32768
// (both sides are odd in x; x = 0 is exact) is checked for every k; where it fails, lo moves by +1, -1, +2, ... ulps until it
// holds for all k.  *fail is raised if no shift up to +-16 ulps works (never observed: 14 positions of all seven sizes take
// one ulp, the rest none; the search exists so that a host libm with a different last bit in cos() cannot break the
// contract silently).  shifted: number of positions whose lo was moved (diagnostics).
static __global__ void __launch_bounds__(256) glv_window_split_kernel(const double* __restrict__ w_tab, float* __restrict__ split,
                                                               uint32_t n, int* __restrict__ fail, int* __restrict__ shifted) {
    __shared__ int bad;
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const double w = w_tab[i];
    const float hi = (float) w, lo0 = (float) (w - (double) hi);
    bool found = false;
    for (int t = 0; t < 33; ++t) {
        const int d = (t + 1) / 2 * ((t & 1) ? 1 : -1);                       // 0, +1, -1, +2, -2, ...
        // lo0 shifted by d ulps (lo0 != 0 and far from the ends of the binade ladder: plain integer steps on the encoding)
        const int enc = __builtin_bit_cast(int, lo0);
        const float lo = lo0 == 0.0f ? (float) d * 0x1p-149f : __builtin_bit_cast(float, enc >= 0 ? enc + d : enc - d);
        if (threadIdx.x == 0) bad = 0;
        __syncthreads();
        int mine = 0;
        for (int k = 1 + (int) threadIdx.x; k <= 32768; k += 256) {
            const float x = div_65535((float) k);
            mine |= apply_window_split(x, hi, lo) != apply_window(x, w);
        }
        if (mine) bad = 1;
        __syncthreads();
        const int b = bad;
        __syncthreads();
        if (!b) {
            if (threadIdx.x == 0) {
                split[(i >> 1) * 4 + (i & 1)] = hi;
                split[(i >> 1) * 4 + 2 + (i & 1)] = lo;
                if (d != 0) atomicAdd(shifted, 1);
            }
            found = true;
            break;
        }
    }
    if (!found && threadIdx.x == 0) { *fail = 1; split[(i >> 1) * 4 + (i & 1)] = hi; split[(i >> 1) * 4 + 2 + (i & 1)] = lo0; }
}

// every (sample value, window position) pair of the table: mismatches of the split product against the fp64 product
static __global__ void __launch_bounds__(256) glv_window_split_check_kernel(const double* __restrict__ w_tab, const float* __restrict__ split,
                                                                     uint32_t n, unsigned long long* __restrict__ mismatches) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const double w = w_tab[i];
    const float hi = split[(i >> 1) * 4 + (i & 1)], lo = split[(i >> 1) * 4 + 2 + (i & 1)];
    unsigned long long mine = 0;
    for (int k = -32768 + (int) threadIdx.x; k <= 32767; k += 256) {
        const float x = div_65535((float) k);
        const float a = apply_window_split(x, hi, lo), r = apply_window(x, w);
        mine += __builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, r);
    }
    if (mine) atomicAdd(mismatches, mine);
}

inline hipError_t launch_window_split_impl(const double* w_tab, float* split, uint32_t n, int* d_fail_shifted, hipStream_t st) {
    hipLaunchKernelGGL(glv_window_split_kernel, dim3(n), dim3(256), 0, st, w_tab, split, n, d_fail_shifted, d_fail_shifted + 1);
    return hipGetLastError();
}
inline hipError_t launch_window_split_check_impl(const double* w_tab, const float* split, uint32_t n, unsigned long long* d_mismatches, hipStream_t st) {
    hipLaunchKernelGGL(glv_window_split_check_kernel, dim3(n), dim3(256), 0, st, w_tab, split, n, d_mismatches);
    return hipGetLastError();
}


}  // namespace glv
