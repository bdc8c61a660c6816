// Internal helpers shared by the gfx950 kernels of libpicons.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <mutex>
#include "../../include/picons.h"

void pc_set_error(const char* fmt, ...);

// Set by pc_run_ops_timed around a conv op: the launch then carries the event pair in its own dispatch packet
// (hipExtLaunchKernelGGL), so timing a kernel costs no extra barrier packets on the stream.  Null otherwise.
extern thread_local hipEvent_t pc_tl_ev_start, pc_tl_ev_stop;

#define PC_CHECK_ARG(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) { pc_set_error(__VA_ARGS__); return PC_E_ARG; } \
    } while (0)

#define PC_CHECK_LAUNCH(what)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            pc_set_error("%s: %s", what, hipGetErrorString(e_));                     \
            return PC_E_LAUNCH;                                                      \
        }                                                                            \
    } while (0)

// One-time hipFuncSetAttribute(MaxDynamicSharedMemorySize) per kernel instantiation.  The C-ABI is entered from Python's main thread and
// from the autograd engine's worker thread: std::call_once instead of a plain static flag (a data race, however benign), and a refusal
// surfaces through pc_last_error() instead of as a generic launch failure.  `fn` in parentheses if it carries template commas.
#define PC_SET_LDS_ONCE(fn, bytes, what)                                                                                             \
    do {                                                                                                                             \
        static std::once_flag once_;                                                                                                 \
        static hipError_t rc_ = hipSuccess;                                                                                          \
        const size_t b_ = (size_t)(bytes);                                                                                           \
        std::call_once(once_, [&] { rc_ = hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b_); }); \
        if (rc_ != hipSuccess) { pc_set_error("%s: %zu bytes of dynamic LDS refused: %s", what, b_, hipGetErrorString(rc_)); return PC_E_LAUNCH; } \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// XCD-aware bijective block remap (cdna guide T1): consecutive logical ids share an XCD/L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Barrier that must also publish the LDS-DMA (buffer_load ... lds) tiles the block's waves have issued: every wave's own transfers have landed
// (vmcnt(0)) before it arrives.  __syncthreads() alone is NOT a contract for that: hipcc (ROCm 7.2) adds the vmcnt(0) only when its waitcnt pass
// knows of a pending LDS-DMA, and in round 5 it lost that knowledge across a loop back edge (conv_x6.hip: a barrier of the K loop was emitted
// with lgkmcnt(0) alone and waves read tiles that had not landed).  The wait is written as inline asm, which the pass cannot drop.
#define PC_SYNC_DMA() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)

