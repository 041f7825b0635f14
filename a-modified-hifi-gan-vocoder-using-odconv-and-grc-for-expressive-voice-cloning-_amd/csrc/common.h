// Shared device helpers for the MI355X (gfx950) vocoder kernels.
// Storage types: float, bf16, fp16; arithmetic/accumulation is always fp32.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/mi355x_vocoder.h"

namespace mv {

using bf16 = __hip_bfloat16;
using f16 = _Float16;

constexpr int WAVE = 64;  // CDNA wavefront

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16>(const bf16* p) {
  return __uint_as_float(((uint32_t)(*reinterpret_cast<const uint16_t*>(p))) << 16);
}
template <> __device__ __forceinline__ float ld<f16>(const f16* p) { return (float)(*p); }
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a release/acquire fence pair around s_barrier, and on gfx9 loads and
// stores share one counter: hipcc emits `s_waitcnt vmcnt(0)` in front of every barrier, so a global prefetch issued before the barrier
// (the next tile's rows, the next k-steps' weight fragments) is waited for right there and overlaps nothing - in-kernel marks of the
// multi-tile ODConv kernel: 14.7 k ticks per tile "staging" a tile that had been requested a whole tile earlier.  Use this one where the
// barrier only separates LDS writers from LDS readers; global loads stay in flight across it (their own uses are still waited for by
// the compiler), global stores are not made visible to the other waves by it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A load whose conversion - its first use, i.e. the wait for it - is deliberately placed later: the raw bits, zero-extended to one
// 32-bit register (16-bit values kept in a `T` array get packed in pairs right behind the loads, which is a use as well).
template <typename T> __device__ __forceinline__ uint32_t ldraw(const T* p);
template <> __device__ __forceinline__ uint32_t ldraw<float>(const float* p) { return *reinterpret_cast<const uint32_t*>(p); }
template <> __device__ __forceinline__ uint32_t ldraw<bf16>(const bf16* p) { return *reinterpret_cast<const uint16_t*>(p); }
template <> __device__ __forceinline__ uint32_t ldraw<f16>(const f16* p) { return *reinterpret_cast<const uint16_t*>(p); }
template <typename T> __device__ __forceinline__ float rawtofl(uint32_t r);
template <> __device__ __forceinline__ float rawtofl<float>(uint32_t r) { return __uint_as_float(r); }
template <> __device__ __forceinline__ float rawtofl<bf16>(uint32_t r) { return __uint_as_float(r << 16); }
template <> __device__ __forceinline__ float rawtofl<f16>(uint32_t r) { return (float)__builtin_bit_cast(f16, (uint16_t)r); }

template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16>(bf16* p, float v) { *p = __float2bfloat16(v); }  // RNE, NaN-safe
template <> __device__ __forceinline__ void st<f16>(f16* p, float v) { *p = (f16)v; }

// round-trip through the storage type (used where the reference order has a stored intermediate)
template <typename T> __device__ __forceinline__ float rnd(float v) {
  T t; st<T>(&t, v); return ld<T>(&t);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// Four wave sums at once on the DPP path (no LDS round trips: __shfl_xor is ds_bpermute, ~100 cycles a step): quad butterfly, half-row
// and row mirrors give every lane its 16-lane row total, the four row totals are read with v_readlane and added in a fixed order.
// Every lane returns the same bits.
__device__ __forceinline__ void wave_sum4_dpp(float (&v)[4]) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] += dpp(v[i], std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] += dpp(v[i], std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] += dpp(v[i], std::integral_constant<int, 0x141>{});     // row_half_mirror
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] += dpp(v[i], std::integral_constant<int, 0x140>{});     // row_mirror
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = __builtin_bit_cast(int, v[i]);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    v[i] = (r0 + r1) + (r2 + r3);
  }
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = (threadIdx.x < nw) ? red[threadIdx.x] : 0.f;
  if (wid == 0) {
    r = wave_sum(r);
    if (lane == 0) red[0] = r;
  }
  __syncthreads();
  return red[0];
}

enum Act { ACT_NONE = 0, ACT_LRELU = 1, ACT_TANH = 2, ACT_SILU = 3 };

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case ACT_LRELU: return v >= 0.f ? v : v * slope;
    case ACT_TANH: return tanhf(v);
    case ACT_SILU: return v / (1.f + __expf(-v));
    default: return v;
  }
}

// 8 consecutive 16-bit elements <-> 8 floats through one 16-byte access (p must be 16-byte aligned; T is bf16 or f16)
template <typename T> __device__ __forceinline__ void load8f(const T* p, float* o) {
  alignas(16) T tmp[8];
  *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = ld<T>(tmp + e);
}
template <typename T> __device__ __forceinline__ void store8f(T* p, const float* v) {
  alignas(16) T tmp[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) st<T>(tmp + e, v[e]);
  *reinterpret_cast<uint4*>(p) = *reinterpret_cast<uint4*>(tmp);
}
__host__ __device__ inline bool all_mult8(long a, long b = 0, long c2 = 0, long d = 0, long e = 0, long f = 0, long g = 0) {
  return ((a | b | c2 | d | e | f | g) & 7) == 0;
}

// Activation functors for epilogue loops.  `apply_act(v, act, slope)` with a run-time `act` inside an unrolled element loop
// compiles to a scalar branch tree PER ELEMENT (with the tanh / SiLU bodies inlined each time): kernels pick one functor
// per launch instead - ActLrelu covers "none" (negative slope 1) and LeakyReLU branch-free, ActAny is the general case.
struct ActLrelu {
  float ns;
  __device__ __forceinline__ float operator()(float v) const { return v >= 0.f ? v : v * ns; }
};
struct ActAny {
  int act; float slope;
  __device__ __forceinline__ float operator()(float v) const { return apply_act(v, act, slope); }
};

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace mv

// dtype dispatch: BODY sees `T`
#define MV_DISPATCH(dtype, ...)                                         \
  switch (dtype) {                                                      \
    case MV_F32: { using T = float; __VA_ARGS__; break; }               \
    case MV_BF16: { using T = mv::bf16; __VA_ARGS__; break; }           \
    case MV_F16: { using T = mv::f16; __VA_ARGS__; break; }             \
    default: return MV_ERR_DTYPE;                                       \
  }

// internal cross-file entry (conv_out.hip), not part of the C ABI
// (part8 / tab8 / nwg / eps: the fp32 MFMA form forms the deferred GroupNorm affine itself from the last block's partial sums and
//  `ab` is not read - mvi_conv_out_affine_takes_partials says whether that form will run)
bool mvi_conv_out_affine_takes_partials(int dtype, int ks);
int mvi_conv_out_affine(const void* f, const void* x, const float* ab, const float* wt, float bias, void* y, int B, int T_, int C, int ks,
                        int pad, int act, int dtype, hipStream_t stream, const float* part8 = nullptr, const float* tab8 = nullptr,
                        int nwg = 0, float eps = 0.f, int x_pair = 0);
// internal cross-file entry (mrf_stream.hip): the MRF chain in its streaming form (MV_F32_W16, dilations (1, 3, 5) only)
int mvi_mrf_chain_stream(const void* x, void* out, const void* const* packed, int nblocks, char* ws, size_t act_bytes, int B, int Tn,
                         float eps, hipStream_t stream, const void** f_last, const void** x_last, int* x_last_pair,
                         const float** part8_last, const float** tab_last, int* nwg_last, int x_pair_in = 0);

// Zero fill as a KERNEL launch.  hipMemsetAsync becomes a memset node when a stream is being captured, and on this ROCm build a memset
// node is not reliably ordered in front of the kernel node that follows it in a replayed graph: a captured training step whose
// weight-gradient kernels accumulate with atomics onto a freshly zeroed buffer produced inf gradients from its second replay on as
// soon as the replays were not issued back to back (tests/test_gpu_train_graph.py).  A kernel node has no such problem.
hipError_t mvi_zero_async(void* p, size_t bytes, hipStream_t stream);

#define MV_CHECK_ARG(cond) do { if (!(cond)) return MV_ERR_ARG; } while (0)
// runtime calls that are not kernel launches: propagate the hipError_t as the entry point's (positive) return code
#define MV_HIP(call) do { const hipError_t mv_e_ = (call); if (mv_e_ != hipSuccess) return (int)mv_e_; } while (0)
#define MV_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
