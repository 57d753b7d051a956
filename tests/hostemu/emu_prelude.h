/* emu_prelude.h -- TEST INFRASTRUCTURE: lets g++ compile the translation unit libblinkyhip generates for hiprtc
 * (bk_emit.cpp + bkm.h + bk_device_rt.h + bk_build_kernels.h) as plain host C++, one "thread" per block, so that
 * CPU tests and tools/flag_probe.py can inspect what the device code computes and which pixels it flags
 * (bk_device_rt.h's exactness bookkeeping) without a GPU.  Never linked into the product. */
#pragma once
#include <cstring>
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(x)
#define __shared__ static
#define __HIP_MEMORY_SCOPE_AGENT 0
struct EmuDim3 { unsigned x, y, z; };
static EmuDim3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0}, blockDim = {1, 1, 1}, gridDim = {1, 1, 1};
static inline void __syncthreads() {}
template <typename T> static inline T __hip_atomic_load(const T *p, int, int) { return *p; }
template <typename T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
/* one "thread" per block: a wave of one lane */
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
static inline int __popcll(unsigned long long m) { return __builtin_popcountll(m); }
static inline int __ffsll(long long m) { return __builtin_ffsll(m); }
static inline int __shfl(int v, int) { return v; }
static inline int __shfl_down(int v, unsigned int) { return v; }   /* (a wave of one lane: no neighbour) */
static inline int __shfl_xor(int v, int) { return v; }
template <typename T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
