// Channels-last ODConv1d / ODConvTranspose1d on MFMA: attention -> kernel aggregation -> convolution
// (+ bias, FiLM, LeakyReLU, next-layer pooling) in ONE launch.
//   reference arithmetic: hifigan_modified/odconv.py:73-108 (ODConv1d) and :172-205 (ODConvTranspose1d);
//                         FiLM epilogue: grc_lora.py:111-129; LeakyReLU(0.1): SURVEY.md §A item 2.
//
// Formulation.  Per sample b the layer is a GEMM  D[row][q] = sum_k Wb[row][k] X[k][q]  with the per-sample kernel
// Wb = sum_kb alpha[b,kb] W[kb] (K banks) formed ON THE FLY in registers while the A operand is loaded:
//   regular conv (stride 1):   row = o,        k = (j, c),   X[(j,c)][t] = x[t - pad + j*dil][c],  y[t][o]
//   transposed conv:           row = (r, o),   k = (m, c),   X[(m,c)][q] = x[q - m][c],            y[s*q + r - pad][o]
//     with kernel tap j = r + m*s  (r = output phase, m = 0..ks/s-1): every output sample is produced exactly once,
//     there is no overlap-add and the output rows of one q are s*Cout CONTIGUOUS channels-last elements.
// alpha itself is computed in the prologue from the pooled channel sums that the PRODUCING kernel accumulated in its
// epilogue (pooled_out), so the "mean over all T" dependency of odconv.py:37,85 costs no extra pass over x.
//
// Tiling: workgroup = 4 waves; wave w owns MW 16-row M-tiles; the workgroup covers S samples x NB 16-column N-tiles.
// The x tiles of the S samples live in LDS (row stride padded by 16 B); bank weights stream from L2 in packed
// A-fragment order (one coalesced 1 KB load per bank per fragment) and are reused for the S samples.
#include "mfma.h"
#include <type_traits>
#include <cstdlib>
#include <cstdio>

namespace mv {

constexpr int OD_MAXK = 8;

struct OdP {
  int B, Cin, Tin, Cout, Tout, ks, stride, pad, dil, transposed, K, act;
  float slope;
  int M;          // GEMM rows: Cout (conv) or stride*Cout (transposed)
  int ntaps;      // ks (conv) or ks/stride (transposed)
  int nchunks;    // ntaps*Cin/8 valid 8-channel chunks along k
  int ksteps;     // ceil(nchunks/4)
  int nq;         // GEMM columns per sample: Tout (conv) or Tin + ntaps - 1 (transposed)
  int shift_lo;   // lowest input-row shift: -pad (conv) or -(ntaps-1) (transposed)
  int nrows;      // LDS rows per sample tile
  int film_F;
  int pool_n;     // floats per sample in pooled_in: slots * rows partial sums of the producing launch (Cin = dense sums)
  int in_f16;     // fp32 launch whose INPUT x is fp16 (mixed storage: the first fp32 stage reads the fp16 stream; multi-tile kernel only)
  int out_pair;   // fp32 launch whose OUTPUT rows are written in the MRF chain's pair-row format (streaming kernel, 64 output channels)
#ifdef MV_OD_TIMING
  int dbg;        // ablation switches of the timing build (MV_KL_DBG)
#endif
};

template <typename T> struct WLoad;   // this lane's 8 packed weights -> fp32
template <> struct WLoad<bf16> {
  static constexpr int BYTES = 16;
  struct R { u32x4 u; };
  static __device__ __forceinline__ R load(const char* p) { R r; r.u = *reinterpret_cast<const u32x4*>(p); return r; }
  static __device__ __forceinline__ void fma8(const R& r, float a, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] += a * bf16_lo(r.u[i]); f[2 * i + 1] += a * bf16_hi(r.u[i]); }
  }
};
template <> struct WLoad<f16> {
  static constexpr int BYTES = 16;
  struct R { f16x8_t h; };
  static __device__ __forceinline__ R load(const char* p) { R r; r.h = *reinterpret_cast<const f16x8_t*>(p); return r; }
  static __device__ __forceinline__ void fma8(const R& r, float a, float* f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] += a * (float)r.h[i];
  }
};
template <> struct WLoad<float> {
  static constexpr int BYTES = 32;
  struct R { f32x4 a, b; };
  static __device__ __forceinline__ R load(const char* p) {
    R r; r.a = reinterpret_cast<const f32x4*>(p)[0]; r.b = reinterpret_cast<const f32x4*>(p)[1]; return r;
  }
  static __device__ __forceinline__ void fma8(const R& r, float a, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] += a * r.a[i]; f[4 + i] += a * r.b[i]; }
  }
};

template <typename T> __device__ __forceinline__ typename Mma<T>::V make_a(const float* f);
template <> __device__ __forceinline__ Mma<bf16>::V make_a<bf16>(const float* f) {
  u32x4 u = {pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]), pack_bf16(f[4], f[5]), pack_bf16(f[6], f[7])};
  Mma<bf16>::V r; r.v = __builtin_bit_cast(bf16x8_t, u); return r;
}
template <> __device__ __forceinline__ Mma<f16>::V make_a<f16>(const float* f) {
  u32x4 u = {pack_f16(f[0], f[1]), pack_f16(f[2], f[3]), pack_f16(f[4], f[5]), pack_f16(f[6], f[7])};
  Mma<f16>::V r; r.v = __builtin_bit_cast(f16x8_t, u); return r;
}
template <> __device__ __forceinline__ Mma<float>::V make_a<float>(const float* f) { return Mma<float>::split(f); }

// ------------------------------------------------------------------------------------------------ pack
// packed[kb][mtile][kstep][lane][8] : element j of lane = W row (16*mtile + lane&15), k-chunk (4*kstep + lane>>4), k = 8*chunk + j
template <typename T, typename P>
__global__ __launch_bounds__(256) void odconv_pack_kernel(const P* __restrict__ w, T* __restrict__ out, OdP p) {
  const long per_bank = (long)(p.M / 16) * p.ksteps * 512;
  const long total = per_bank * p.K;
  const int cpc = p.Cin / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = idx % 8, lane = (idx / 8) % 64;
    const long fr = idx / 512;
    const int kstep = fr % p.ksteps;
    const int mt = (fr / p.ksteps) % (p.M / 16);
    const int kb = fr / ((long)p.ksteps * (p.M / 16));
    const int row = 16 * mt + (lane & 15), chunk = 4 * kstep + (lane >> 4);
    float v = 0.f;
    if (chunk < p.nchunks) {
      const int tap = chunk / cpc, c = 8 * (chunk % cpc) + j;
      if (p.transposed) {
        const int r = row / p.Cout, o = row % p.Cout, jj = r + tap * p.stride;
        v = ld<P>(w + (((long)kb * p.Cin + c) * p.Cout + o) * p.ks + jj);       // [K][Cin][Cout][ks]
      } else {
        v = ld<P>(w + (((long)kb * p.Cout + row) * p.Cin + c) * p.ks + tap);    // [K][Cout][Cin][ks]
      }
    }
    st<T>(out + idx, v);
  }
}

#ifdef MV_OD_TIMING
__device__ long long* od_dbg = nullptr;
#define OD_TM() do { if (ntm < 8) tmk[ntm++] = clock64(); } while (0)
#else
#define OD_TM() do {} while (0)
#endif
// logits z[s][k] = Wa[k] . mean[s] + ba[k] from the channel sums in `scratch` (sample s at scratch + s * sstride): a wave takes two
// (sample, bank) pairs at a time and issues ALL their attention-weight loads before the first multiply - a `for (c = lane; c < Cin;
// c += 64) acc += w[c] * m[c]` loop is one L2 round trip per 64 channels (8 in a row for the first upsampler's 512: 12 k of its 70 k
// ticks went here, before anything else could start).
constexpr int OD_LU = 8;                                   // 64-channel slices of attention weights a lane holds per (sample, bank) pair
// The attention weights (and bias) of this wave's first two (sample, bank) pairs, requested before anything else in the prologue - they
// depend on nothing, so the partial sums, their reduction and these loads are ONE memory round trip instead of three in a row (in-kernel
// marks: partial sums 1.5 k, logits 3-6 k - a weight round trip, then the bias loaded after the wave tree - and softmax 2.5 k ticks).
template <typename T>
__device__ __forceinline__ bool od_logits_preload(uint32_t (&wv)[2][OD_LU], uint32_t (&bv)[2], int S, const OdP& p,
                                                  const T* __restrict__ att_w, const T* __restrict__ att_b) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int Cin = p.Cin, npair = S * p.K;
  if (npair > 8 || Cin > 64 * OD_LU) return false;
  // every load unconditional at a clamped address and kept RAW (storage type): `cc < Cin ? ld(...) : 0` compiles to a branch per load
  // with an s_waitcnt vmcnt(0) inside - 18 memory round trips in a row (15 k of the first upsampler's 25 k prologue ticks) - and a
  // conversion here would be the loads' first use, i.e. a wait for them before the caller's own loads are issued
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int pr = wid * 2 + q < npair ? wid * 2 + q : npair - 1, kb = pr % p.K;
#pragma unroll
    for (int u = 0; u < OD_LU; ++u) {
      const int cc = 64 * u + lane;
      wv[q][u] = ldraw<T>(att_w + (long)kb * Cin + (cc < Cin ? cc : Cin - 1));
    }
    bv[q] = ldraw<T>(att_b ? att_b + kb : att_w);
  }
  return true;
}

template <typename T>
__device__ __forceinline__ void od_logits(float* alds, const float* scratch, int sstride, int S, const OdP& p,
                                          const T* __restrict__ att_w, const T* __restrict__ att_b,
                                          bool have, const uint32_t (&pre)[2][OD_LU], const uint32_t (&preb)[2]) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int Cin = p.Cin, npair = S * p.K;
  constexpr int U = OD_LU;                                 // 64-channel slices per batch
  if (have) {                                              // npair <= 8, Cin <= 512: one batch, weights already in registers
    const int pr0 = wid * 2;
    float acc[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pr = pr0 + q < npair ? pr0 + q : npair - 1, s = pr / p.K;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cc = 64 * u + lane;
        const float m = scratch[s * sstride + (cc < Cin ? cc : Cin - 1)];     // (the weights of cc >= Cin are zeros)
        acc[q] += cc < Cin ? rawtofl<T>(pre[q][u]) * m : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float v = wave_sum(acc[q]);
      const int pr = pr0 + q;
      uint32_t rb = preb[q];
      asm volatile("" : "+v"(rb));                         // (keeps the conversion - the wait for the preloads - down here)
      if (lane == 0 && pr < npair) {
        const int s = pr / p.K, kb = pr % p.K;
        alds[s * OD_MAXK + kb] = v / (float)p.Tin + (att_b ? rawtofl<T>(rb) : 0.f);
      }
    }
    return;
  }
  for (int pr0 = wid * 2; pr0 < npair; pr0 += 8) {
    float acc[2] = {0.f, 0.f};
    for (int c0 = 0; c0 < Cin; c0 += 64 * U) {
      float wv[2][U];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int pr = pr0 + q < npair ? pr0 + q : npair - 1, kb = pr % p.K;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int cc = c0 + 64 * u + lane;
          wv[q][u] = ld<T>(att_w + (long)kb * Cin + (cc < Cin ? cc : Cin - 1));      // (unconditional: see od_logits_preload)
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int pr = pr0 + q < npair ? pr0 + q : npair - 1, s = pr / p.K;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int cc = c0 + 64 * u + lane;
          const float m = scratch[s * sstride + (cc < Cin ? cc : Cin - 1)];
          acc[q] += cc < Cin ? wv[q][u] * m : 0.f;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float v = wave_sum(acc[q]);
      const int pr = pr0 + q;
      if (lane == 0 && pr < npair) {
        const int s = pr / p.K, kb = pr % p.K;
        alds[s * OD_MAXK + kb] = v / (float)p.Tin + (att_b ? ld<T>(att_b + kb) : 0.f);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ attention prologue
// alpha[s][:] = softmax_k( Wa . mean_t x + ba ) (odconv.py:36-40,85) for the S samples of a workgroup, from the PRODUCER's partial
// channel sums (pooled_in: pool_n = slots x rows floats per sample, row % Cin = channel).  All 256 threads take part: thread t owns
// channel t % Cin and every G-th partial of it (G = 256 / Cin thread groups), all its loads are independent and issued together;
// the G group sums and then the K logits are added in a fixed order, so the result does not depend on timing (no atomics).
// (One wave per (sample, bank) walking pool_n elements with a dependent load per step cost 7-12 us per layer: a chain of
// pool_n / 64 L2 round trips in front of everything else.)  `scratch` = >= S * 256 floats of LDS that nothing else uses yet.
template <typename T, int NTHR = 256>                     // NTHR: threads of the workgroup (those past 256 only take part in the logits)
__device__ __forceinline__ void od_alpha_from_partials(float* alds, float* scratch, int S, int b0, const OdP& p,
                                                       const float* __restrict__ pooled_in, const T* __restrict__ att_w,
                                                       const T* __restrict__ att_b, long long* tm = nullptr) {
  const int tid = threadIdx.x;
  const int Cin = p.Cin;
  const int npc = p.pool_n / Cin;                       // partials per channel
  // Fast tail (K <= 4 banks, Cin <= 512, a wave per sample): wave s holds ALL of sample s's attention weights (raw, requested here -
  // they depend on nothing), forms the K logits with one DPP reduction and the softmax in registers.  The general tail below goes
  // through (sample, bank) pairs, an LDS tree per pair, a barrier and a softmax thread: 6 k ticks behind the partial sums at the first
  // upsampler, against ~1.5 k here.
  const int wid_ = tid >> 6, lane_ = tid & 63;
  const bool fast = p.K <= 4 && Cin <= 64 * OD_LU && S <= NTHR / 64;
  uint32_t wq[4][OD_LU], bq[4];
  if (fast && wid_ < S) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int kb = k < p.K ? k : 0;
#pragma unroll
      for (int u = 0; u < OD_LU; ++u) {
        const int cc = 64 * u + lane_;
        wq[k][u] = ldraw<T>(att_w + (long)kb * Cin + (cc < Cin ? cc : Cin - 1));
      }
      bq[k] = ldraw<T>(att_b ? att_b + kb : att_w);
    }
  }
  auto fast_tail = [&](int sstride) {
    if (wid_ < S) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < OD_LU; ++u) {
        const int cc = 64 * u + lane_;
        const float m = scratch[wid_ * sstride + (cc < Cin ? cc : Cin - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += cc < Cin ? rawtofl<T>(wq[k][u]) * m : 0.f;
      }
      wave_sum4_dpp(acc);
      float z[4], mx = -INFINITY, den = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        z[k] = k < p.K ? acc[k] / (float)p.Tin + (att_b ? rawtofl<T>(bq[k]) : 0.f) : -INFINITY;
        mx = fmaxf(mx, z[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { z[k] = k < p.K ? expf(z[k] - mx) : 0.f; den += z[k]; }
      if (lane_ == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < p.K) alds[wid_ * OD_MAXK + k] = z[k] / den;
      }
    }
  };
  uint32_t wpre[2][OD_LU] = {}, bpre[2] = {};
  bool havew = false;
  if (!fast) havew = od_logits_preload<T>(wpre, bpre, S, p, att_w, att_b);
  if (Cin <= 256) {
    const int G = 256 / Cin, c = tid % Cin, gq = tid / Cin;          // Cin is a multiple of 8; threads beyond G * Cin idle
    // two samples per sweep: their loads are independent and issued together (a sample after the other is a memory round trip each)
    for (int s0 = 0; s0 < S; s0 += 2) {
      float a[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      bool ok[2];
      const float* src[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        ok[q] = gq < G && s0 + q < S && b0 + s0 + q < p.B;
        src[q] = pooled_in + (long)(ok[q] ? b0 + s0 + q : b0) * p.pool_n + c;
      }
      int j = gq;
      for (; j + 3 * G < npc; j += 4 * G) {
        float v[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[q][i] = ok[q] ? src[q][(long)(j + i * G) * Cin] : 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) a[q][i] += v[q][i];
      }
      for (; j < npc; j += G) {
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) v[q] = ok[q] ? src[q][(long)j * Cin] : 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) a[q][0] += v[q];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (gq < G && s0 + q < S) scratch[((s0 + q) * G + gq) * Cin + c] = (a[q][0] + a[q][1]) + (a[q][2] + a[q][3]);
    }
    __syncthreads();
    for (int i = tid; i < S * Cin; i += NTHR) {           // fixed-order sum over the G groups -> scratch[s][0][c]
      const int s = i / Cin, cc = i - s * Cin;
      float a = 0.f;
      for (int q = 0; q < G; ++q) a += scratch[(s * G + q) * Cin + cc];
      scratch[(s * G) * Cin + cc] = a;
    }
    __syncthreads();
    if (fast) { fast_tail(G * Cin); return; }
    od_logits<T>(alds, scratch, G * Cin, S, p, att_w, att_b, havew, wpre, bpre);
  } else {
    // wide inputs (Cin > 256, e.g. the first upsampler's 512): few partials per channel.  Thread t owns the (sample, channel) elements
    // t, t + 256, ... of [S][Cin], four at a time with their loads issued together: partial j of all four, added in index order.
    // (A `for (s) for (c)` loop with the load inside is one dependent memory round trip per element and thread: 4 x ~4.5 k ticks at
    // the first upsampler, where this prologue was 25 k of the kernel's 80 k ticks.)
    const int E = S * Cin;
    for (int u0 = 0; u0 < E; u0 += 1024) {
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      const float* src[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = u0 + tid + 256 * u;
        const int s = idx / Cin, cc = idx - s * Cin;
        ok[u] = idx < E && b0 + s < p.B && tid < 256;
        src[u] = pooled_in + (long)(ok[u] ? b0 + s : b0) * p.pool_n + (ok[u] ? cc : 0);
      }
      for (int j = 0; j < npc; ++j) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ok[u] ? src[u][(long)j * Cin] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += v[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u0 + tid + 256 * u < E && tid < 256) scratch[u0 + tid + 256 * u] = a[u];       // [s][Cin]: needs S * Cin floats of scratch
    }
    if (tm) tm[0] = clock64();
    __syncthreads();
    if (tm) tm[1] = clock64();
    if (fast) { fast_tail(Cin); if (tm) tm[2] = clock64(); return; }
    od_logits<T>(alds, scratch, Cin, S, p, att_w, att_b, havew, wpre, bpre);
  }
  if (tm) tm[2] = clock64();
  __syncthreads();
  if (tid < S) {                                          // softmax over the banks, values in registers (one LDS round trip, not 3 K)
    float z[OD_MAXK], m = -INFINITY, den = 0.f;
#pragma unroll
    for (int kb = 0; kb < OD_MAXK; ++kb) { z[kb] = kb < p.K ? alds[tid * OD_MAXK + kb] : -INFINITY; m = fmaxf(m, z[kb]); }
#pragma unroll
    for (int kb = 0; kb < OD_MAXK; ++kb) { z[kb] = kb < p.K ? expf(z[kb] - m) : 0.f; den += z[kb]; }
#pragma unroll
    for (int kb = 0; kb < OD_MAXK; ++kb) if (kb < p.K) alds[tid * OD_MAXK + kb] = z[kb] / den;
  }
}

// The same attention for ONE sample on ONE wave, no LDS and no barrier (K <= 4, Cin <= 512): lane l owns the channels l, l + 64, ..;
// all of its loads - attention weights, bias, the producer's partial sums - are requested together and are the wave's first memory
// traffic, so they are one round trip AHEAD of whatever the other waves of the workgroup stage meanwhile (the workgroup-wide form
// above is two round trips and three barriers, and behind bulk loads its small requests wait for them: 9-12 k ticks in the
// upsamplers' prologues).  The caller picks the wave and separates the write of alds from its readers with a barrier.
template <typename T>
__device__ __forceinline__ void od_alpha_wave(float* alds_s, int b, const OdP& p, const float* __restrict__ pooled_in,
                                              const T* __restrict__ att_w, const T* __restrict__ att_b) {
  const int lane = threadIdx.x & 63;
  const int Cin = p.Cin, npc = p.pool_n / Cin;
  uint32_t wq[4][OD_LU], bq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int kb = k < p.K ? k : 0;
#pragma unroll
    for (int u = 0; u < OD_LU; ++u) {
      const int cc = 64 * u + lane;
      wq[k][u] = ldraw<T>(att_w + (long)kb * Cin + (cc < Cin ? cc : Cin - 1));
    }
    bq[k] = ldraw<T>(att_b ? att_b + kb : att_w);
  }
  float a[OD_LU];
#pragma unroll
  for (int u = 0; u < OD_LU; ++u) a[u] = 0.f;
  const float* src = pooled_in + (long)b * p.pool_n;
  int off[OD_LU];
#pragma unroll
  for (int u = 0; u < OD_LU; ++u) off[u] = 64 * u + lane < Cin ? 64 * u + lane : Cin - 1;
  int j = 0;
  for (; j + 1 < npc; j += 2) {                           // two partials of every owned channel in flight, added in index order
    float v0[OD_LU], v1[OD_LU];
#pragma unroll
    for (int u = 0; u < OD_LU; ++u) { v0[u] = src[(long)j * Cin + off[u]]; v1[u] = src[(long)(j + 1) * Cin + off[u]]; }
#pragma unroll
    for (int u = 0; u < OD_LU; ++u) { a[u] += v0[u]; a[u] += v1[u]; }
  }
  if (j < npc) {
#pragma unroll
    for (int u = 0; u < OD_LU; ++u) a[u] += src[(long)j * Cin + off[u]];
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < OD_LU; ++u)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += 64 * u + lane < Cin ? rawtofl<T>(wq[k][u]) * a[u] : 0.f;
  wave_sum4_dpp(acc);
  float z[4], mx = -INFINITY, den = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    z[k] = k < p.K ? acc[k] / (float)p.Tin + (att_b ? rawtofl<T>(bq[k]) : 0.f) : -INFINITY;
    mx = fmaxf(mx, z[k]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { z[k] = k < p.K ? expf(z[k] - mx) : 0.f; den += z[k]; }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < p.K) alds_s[k] = z[k] / den;
  }
}

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int S, int MW, int NB, bool PF, int KB>
__global__ __launch_bounds__(256) void odconv_cl_kernel(const T* __restrict__ x, const T* __restrict__ wp,
                                                        const T* __restrict__ bias, const float* __restrict__ alpha_in,
                                                        const float* __restrict__ pooled_in, const T* __restrict__ att_w,
                                                        const T* __restrict__ att_b, const T* __restrict__ film,
                                                        T* __restrict__ y, float* __restrict__ pooled_out, OdP p) {
  using M = Mma<T>;
  using V = typename M::V;
  using WL = WLoad<T>;
  constexpr int ES = M::ES;
  // fp32 storage: the x tiles are PRE-SPLIT into a hi and a lo bf16 plane per row when they are staged (one split per element);
  // splitting in load_b() cost ~30 VALU instructions per operand read, NB reads per k-step - more than the k-step's MFMAs
  constexpr bool SPLIT = (ES == 4);
  constexpr int LES = 2;                                       // operand element size in LDS
  extern __shared__ __align__(16) char lds[];
  float* alds = reinterpret_cast<float*>(lds);                 // [S][OD_MAXK] alpha (KB <= OD_MAXK banks used)
  float* bias_l = alds + S * OD_MAXK;                          // [S][4*MW*16] alpha-mixed bias of this workgroup's rows
  char* xl = reinterpret_cast<char*>(bias_l + S * 4 * MW * 16);
  const int PLANE = p.Cin * LES;                               // byte offset of the lo plane inside a row (SPLIT)
  const int RS = lds_row_stride(SPLIT ? 2 * PLANE : p.Cin * ES, LES);
#ifdef MV_OD_TIMING
  long long tmk[8]; int ntm = 0;
#endif
  OD_TM();

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * NB * 16;
  const int mt0 = (blockIdx.y * 4 + wid) * MW;                 // first M-tile of this wave
  const int b0 = blockIdx.z * S;
  const int n_mt = p.M / 16;

  // ---- alpha for the S samples: given, or softmax(Wa . pooled/Tin + ba) (odconv.py:36-40)
  if (alpha_in) {
    if (tid < S * p.K) {
      const int s = tid / p.K, kb = tid % p.K;
      alds[s * OD_MAXK + kb] = (b0 + s < p.B) ? alpha_in[(long)(b0 + s) * p.K + kb] : 0.f;
    }
  } else {
    od_alpha_from_partials<T>(alds, reinterpret_cast<float*>(xl), S, b0, p, pooled_in, att_w, att_b);   // xl: not staged yet
    __syncthreads();                                   // the scratch is dead before the x tiles land in it
  }
  OD_TM();
  // ---- stage x tiles: input rows q0+shift_lo .. +nrows-1 of every sample, zero outside [0,Tin)
  {
    const int cpr = p.Cin * ES / 16;
    const int per = p.nrows * cpr;
    if constexpr (SPLIT) {
      constexpr int UB = 4;
      for (int i0 = tid; i0 < S * per; i0 += 256 * UB) {
        f32x4 v[UB];
        int d[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int i = i0 + u * 256;
          v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          d[u] = -1;
          if (i < S * per) {
            const int s = i / per, rem = i - s * per;
            const int r = rem / cpr, ch = rem - r * cpr;
            const int tin = q0 + p.shift_lo + r;
            d[u] = (s * p.nrows + r) * RS + ch * 8;
            if (b0 + s < p.B && tin >= 0 && tin < p.Tin)
              v[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(x + ((long)(b0 + s) * p.Tin + tin) * p.Cin) + ch * 16);
          }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
          if (d[u] >= 0) {
            u32x2 hi, lo;
            Mma<float>::split4(v[u], hi, lo);
            *reinterpret_cast<u32x2*>(xl + d[u]) = hi;
            *reinterpret_cast<u32x2*>(xl + d[u] + PLANE) = lo;
          }
      }
    } else {
      stage_batched<4, 256>(tid, S * per, xl, [&](int i, const void*& src, int& dst) {
        const int s = i / per, rem = i - s * per;
        const int r = rem / cpr, ch = rem - r * cpr;
        const int tin = q0 + p.shift_lo + r;
        if (b0 + s < p.B && tin >= 0 && tin < p.Tin)
          src = reinterpret_cast<const char*>(x + ((long)(b0 + s) * p.Tin + tin) * p.Cin) + ch * 16;
        dst = (s * p.nrows + r) * RS + ch * 16;
      });
    }
  }
  __syncthreads();
  OD_TM();
  // alpha-mixed bias of this workgroup's rows, once, into LDS: the K bank loads of a row are independent and overlap the first
  // weight fetch (read one at a time in the epilogue they were serialized L2 round trips)
  if (bias) {
    constexpr int RWp = 4 * MW * 16;
    for (int i = tid; i < S * RWp; i += 256) {
      const int s = i / RWp, rr = i - s * RWp;
      const int row = blockIdx.y * RWp + rr;
      float v[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int o = row < p.M ? (p.transposed ? row % p.Cout : row) : 0;
        v[kb] = ld<T>(bias + (long)(kb < p.K ? kb : 0) * p.Cout + o);
      }
      float a = 0.f;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) a += (kb < p.K ? alds[s * OD_MAXK + kb] : 0.f) * v[kb];
      bias_l[i] = a;
    }
  }

  float al[S][KB];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) al[s][kb] = kb < p.K ? alds[s * OD_MAXK + kb] : 0.f;

  f32x4 acc[S][MW][NB];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int mw = 0; mw < MW; ++mw)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[s][mw][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cpc = p.Cin / 8;
  const long bank_stride = (long)n_mt * p.ksteps * 512 * ES;     // bytes between banks
  const char* wlane = reinterpret_cast<const char*>(wp) + (long)lane * 8 * ES;
  auto wload = [&](int kstep, typename WL::R (&dst)[MW][KB]) {
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) {
      const int mt = (mt0 + mw) < n_mt ? (mt0 + mw) : (n_mt - 1);   // clamp: out-of-range tiles compute garbage that is never stored
      const char* wbase = wlane + ((long)mt * p.ksteps + kstep) * 512 * ES;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        dst[mw][kb] = WL::load(wbase + (kb < p.K ? kb : 0) * bank_stride);   // unconditional: alpha of a missing bank is 0
    }
  };
  typename WL::R wr[MW][KB];
  wload(0, wr);
  int tap = 0, c8 = g;                                           // this lane's chunk (4*kstep + g) as (tap, c8)
  while (c8 >= cpc) { c8 -= cpc; ++tap; }
  for (int kstep = 0; kstep < p.ksteps; ++kstep) {
    typename WL::R wn[MW][KB];
    // next fragments travel L2 -> registers under this step's math; the load is unconditional (clamped to the last k-step):
    // a branch around it makes the compiler wait for vmcnt(0), i.e. for the fragments it has just requested
    if (PF) wload(kstep + 1 < p.ksteps ? kstep + 1 : p.ksteps - 1, wn);
    // per-sample kernels for this k-step: sum_kb alpha[s,kb] * W[kb]  (fp32, then one rounding to the operand type)
    V afr[S][MW];
#pragma unroll
    for (int mw = 0; mw < MW; ++mw)
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
          WL::fma8(wr[mw][kb], al[s][kb], f);
        afr[s][mw] = make_a<T>(f);
      }
    // B operands: x[q + shift(tap)][8*c8 ..], loaded just in time
    const bool kvalid = (4 * kstep + g) < p.nchunks;
    const int shift = p.transposed ? -tap : (tap * p.dil - p.pad);
    const int rbase = kvalid ? (shift - p.shift_lo + col) : col;
    const int coff = kvalid ? c8 * 8 * LES : 0;
    const int bbase = rbase * RS + coff;             // 32-bit LDS offsets
#pragma unroll
    for (int s = 0; s < S; ++s) {
      // all B fragments of this sample first (independent LDS reads in flight together), then the MFMAs
      V bfr[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) bfr[n] = M::load_bp(xl + (s * p.nrows * RS + bbase + n * 16 * RS), PLANE);
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int mw = 0; mw < MW; ++mw) acc[s][mw][n] = M::mma(afr[s][mw], bfr[n], acc[s][mw][n]);
    }
    if (PF) {
#pragma unroll
      for (int mw = 0; mw < MW; ++mw)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
          wr[mw][kb] = wn[mw][kb];
    } else {
      wload(kstep + 1 < p.ksteps ? kstep + 1 : p.ksteps - 1, wr);
    }
    c8 += 4;
    while (c8 >= cpc) { c8 -= cpc; ++tap; }
  }

  OD_TM();
  // ---- epilogue: bias, FiLM, activation -> LDS tile [s][q][rows of this workgroup] -> whole-row 16-byte stores
  __syncthreads();                                   // every wave is done with the x tiles: reuse the region
  constexpr int RW = 4 * MW * 16;                    // GEMM rows covered by this workgroup
  constexpr int ORS = RW * ES + 16;                  // staged row stride (bytes)
  char* ol = xl;
  const int R0 = blockIdx.y * RW;
  // one epilogue body per (activation kind, FiLM, pooling) combination, selected once per launch: a run-time `act` / pointer
  // test inside the unrolled element loops compiles to scalar branch trees per element (8000 instructions before this)
  auto epilogue = [&](auto actf, auto film_c, auto pool_c) {
    constexpr bool HAS_FILM = decltype(film_c)::value, HAS_POOL = decltype(pool_c)::value;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int b = b0 + s;
#pragma unroll
      for (int mw = 0; mw < MW; ++mw) {
        const int mt = mt0 + mw;
        float rowsum[4] = {0.f, 0.f, 0.f, 0.f};
        if (mt < n_mt && b < p.B) {
          const int row = 16 * mt + 4 * g;
          const int r = p.transposed ? row / p.Cout : 0;
          const int o = p.transposed ? row % p.Cout : row;
          float bv[4] = {0.f, 0.f, 0.f, 0.f}, gam[4] = {1.f, 1.f, 1.f, 1.f}, bet[4] = {0.f, 0.f, 0.f, 0.f};
          if (bias) {
            const f32x4 bl = *reinterpret_cast<const f32x4*>(bias_l + s * RW + (row - R0));
            bv[0] = bl[0]; bv[1] = bl[1]; bv[2] = bl[2]; bv[3] = bl[3];
          }
          if (HAS_FILM)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (o + i < p.film_F) {
                gam[i] = ld<T>(film + (long)b * 2 * p.film_F + o + i);
                bet[i] = ld<T>(film + (long)b * 2 * p.film_F + p.film_F + o + i);
              }
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const int q = q0 + n * 16 + col;
            const int u = p.transposed ? q * p.stride + r - p.pad : q;
            const bool ok = q < p.nq && u >= 0 && u < p.Tout;
            float ov[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float vv = (acc[s][mw][n][i] + bv[i]);
              if (HAS_FILM) vv = M::round_store(vv) * gam[i] + bet[i];   // the reference stores the conv output before FiLM
              vv = actf(vv);
              ov[i] = vv;
              if (HAS_POOL && ok) rowsum[i] += M::round_store(vv);
            }
            M::store4(ol + ((long)(s * NB * 16 + n * 16 + col)) * ORS + (row - R0) * ES, ov);
          }
        }
        if (HAS_POOL) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = rowsum[i];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
            rowsum[i] = v;
          }
          if (pooled_out && col == 0 && mt < n_mt && b < p.B) {
            // partial sums of this workgroup's column block, one slot per blockIdx.x: written once, summed by the consumer
            const int row = 16 * mt + 4 * g;
            *reinterpret_cast<f32x4*>(pooled_out + ((long)b * gridDim.x + blockIdx.x) * p.M + row) =
                f32x4{rowsum[0], rowsum[1], rowsum[2], rowsum[3]};
          }
        }
      }
    }
  };
  {
    using TT = std::true_type; using FF = std::false_type;
    const bool simple = p.act <= ACT_LRELU;
    const ActLrelu al_{p.act == ACT_NONE ? 1.f : p.slope};
    const ActAny aa_{p.act, p.slope};
    if (film) { if (simple) epilogue(al_, TT{}, TT{}); else epilogue(aa_, TT{}, TT{}); }   // FiLM only rides with pooling (input_proj)
    else if (pooled_out) { if (simple) epilogue(al_, FF{}, TT{}); else epilogue(aa_, FF{}, TT{}); }
    else { if (simple) epilogue(al_, FF{}, FF{}); else epilogue(aa_, FF{}, FF{}); }
  }
  OD_TM();
  __syncthreads();
  {
    constexpr int EPC = 16 / ES;                     // elements per 16-byte chunk
    constexpr int CPR = RW / EPC;                    // chunks per staged row
    // a thread keeps its 16-byte column of the staged rows (256 % CPR == 0): the phase / channel split of that column - two
    // integer divisions by the run-time Cout - is done once, not once per row
    static_assert(256 % CPR == 0, "store loop: fixed column per thread");
    const int ch = tid % CPR, row = R0 + ch * EPC;
    const int r = p.transposed ? row / p.Cout : 0;
    const int o = p.transposed ? row % p.Cout : row;
    if (row < p.M)
      for (int e = tid / CPR; e < S * NB * 16; e += 256 / CPR) {
        const int s = e / (NB * 16), qi = e - s * (NB * 16);
        const int b = b0 + s, q = q0 + qi;
        const int u = p.transposed ? q * p.stride + r - p.pad : q;
        if (b >= p.B || q >= p.nq || u < 0 || u >= p.Tout) continue;
        const u32x4 val = *reinterpret_cast<const u32x4*>(ol + ((long)e) * ORS + ch * 16);
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(y + ((long)b * p.Tout + u) * p.Cout + o)) = val;
      }
  }
#ifdef MV_OD_TIMING
  OD_TM();
  if (tid == 0 && od_dbg) {
    const long wgid = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wgid < 8192) for (int i = 0; i < 8; ++i) od_dbg[wgid * 8 + i] = i < ntm ? tmk[i] - tmk[0] : -1;
  }
#endif
}


// ------------------------------------------------------------------------------------------------ multi-tile variant
// The last upsamplers (ODConvTranspose1d with ks = 2*stride, 64/128 input channels, small banks, long sequences): one workgroup
// walks TL consecutive 128-column tiles of ONE sample.  What the one-tile kernel redoes per tile is done once per workgroup: the
// attention softmax, the alpha-mixed bias and - the expensive part - the alpha mix of the K bank fragments (all KST k-steps of a
// wave's rows stay in registers as ready A operands).  The x tile of the NEXT step travels global -> registers under the MFMAs of
// the current one (unconditional, row-clamped loads) and is committed to LDS after the current tile's stores, so no global-load
// latency sits on the per-tile path.  16-bit storage, LeakyReLU / none, no FiLM.
template <typename T, int MW, int NB, int CIN, int KB, typename TI = T>
__global__ __launch_bounds__(256, 2) void odconv_cl_mt_kernel(const TI* __restrict__ x, const T* __restrict__ wp, const T* __restrict__ bias,
                                                           const float* __restrict__ alpha_in, const float* __restrict__ pooled_in,
                                                           const T* __restrict__ att_w, const T* __restrict__ att_b, T* __restrict__ y,
                                                           float* __restrict__ pooled_out, OdP p, int TL) {
  using M = Mma<T>;
  using V = typename M::V;
  using WL = WLoad<T>;
  // fp32 storage: x tiles PRE-SPLIT into hi / lo bf16 planes at commit (split operands, mfma.h); ES = element size in HBM, LES in LDS
  // TI: element type of x in HBM (T, or fp16 feeding an fp32 launch: widened and split as the tile is committed)
  constexpr int ES = M::ES, LES = 2, NTAPS = 2, ESI = sizeof(TI), CPR = CIN * ESI / 16, CPC = CIN / 8, KST = NTAPS * CIN / 32;
  static_assert(ESI == ES || (ES == 4 && ESI == 2), "input type: the storage type, or fp16 into fp32");
  constexpr bool SPLIT = (ES == 4);
  constexpr int PLANE = CIN * LES;
  constexpr int NROWS = NB * 16 + NTAPS - 1, PER = NROWS * CPR, XP = (PER + 255) / 256;
  constexpr int RW = 4 * MW * 16, ORS = RW * ES + 16;
  extern __shared__ __align__(16) char lds[];
  float* alds = reinterpret_cast<float*>(lds);                 // [OD_MAXK]
  float* bias_l = alds + OD_MAXK;                              // [RW]
  char* xl = reinterpret_cast<char*>(bias_l + RW);
  const int RS = lds_row_stride(SPLIT ? 2 * PLANE : CIN * ES, LES);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, col = lane & 15, g = lane >> 4;
  const int mt0 = (blockIdx.y * 4 + wid) * MW, b = blockIdx.z, n_mt = p.M / 16, R0 = blockIdx.y * RW;

  if (alpha_in) {
    if (tid < p.K) alds[tid] = alpha_in[(long)b * p.K + tid];
  } else {
    od_alpha_from_partials<T>(alds, reinterpret_cast<float*>(xl), 1, b, p, pooled_in, att_w, att_b);   // xl: nothing committed yet
  }
  // x tile prefetch registers: piece i = tid + 256 j of the [NROWS][CPR] tile
  uint4 xreg[XP];
  const char* xb = reinterpret_cast<const char*>(x + (long)b * p.Tin * CIN);
  auto xfetch = [&](int q0) {
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      int i = tid + j * 256; if (i >= PER) i = PER - 1;
      const int r = i / CPR, ch = i % CPR;
      int tin = q0 + p.shift_lo + r; tin = tin < 0 ? 0 : (tin >= p.Tin ? p.Tin - 1 : tin);
      xreg[j] = *reinterpret_cast<const uint4*>(xb + ((long)tin * CIN) * ESI + ch * 16);
    }
  };
  auto xcommit = [&](int q0) {
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int i = tid + j * 256;
      if (i < PER) {
        const int r = i / CPR, ch = i % CPR, tin = q0 + p.shift_lo + r;
        const uint4 v = (tin >= 0 && tin < p.Tin) ? xreg[j] : make_uint4(0, 0, 0, 0);
        if constexpr (SPLIT && ESI == 2) {                    // 8 fp16 channels -> 8 hi + 8 lo bf16
          alignas(16) f16 hv[8];
          *reinterpret_cast<uint4*>(hv) = v;
          const f32x4 f0 = {(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]}, f1 = {(float)hv[4], (float)hv[5], (float)hv[6], (float)hv[7]};
          u32x2 h0, l0, h1, l1;
          Mma<float>::split4(f0, h0, l0);
          Mma<float>::split4(f1, h1, l1);
          *reinterpret_cast<u32x4*>(xl + r * RS + ch * 16) = u32x4{h0[0], h0[1], h1[0], h1[1]};
          *reinterpret_cast<u32x4*>(xl + r * RS + PLANE + ch * 16) = u32x4{l0[0], l0[1], l1[0], l1[1]};
        } else if constexpr (SPLIT) {
          u32x2 hi, lo;
          Mma<float>::split4(__builtin_bit_cast(f32x4, v), hi, lo);
          *reinterpret_cast<u32x2*>(xl + r * RS + ch * 8) = hi;
          *reinterpret_cast<u32x2*>(xl + r * RS + PLANE + ch * 8) = lo;
        } else {
          *reinterpret_cast<uint4*>(xl + r * RS + ch * 16) = v;
        }
      }
    }
  };
  const int tile0 = blockIdx.x * TL;
  xfetch(tile0 * NB * 16);
  __syncthreads();                                             // alpha visible
  if (bias) {
    for (int i = tid; i < RW; i += 256) {
      const int row = R0 + i;
      const int o = row < p.M ? row % p.Cout : 0;
      float v[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) v[kb] = ld<T>(bias + (long)(kb < p.K ? kb : 0) * p.Cout + o);
      float a = 0.f;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) a += (kb < p.K ? alds[kb] : 0.f) * v[kb];
      bias_l[i] = a;
    }
  }
  // this sample's kernel, once: A[mw][kstep] = sum_kb alpha[kb] * W[kb] fragments
  V afr[MW][KST];
  {
    float al[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) al[kb] = kb < p.K ? alds[kb] : 0.f;
    const long bank_stride = (long)n_mt * KST * 512 * ES;
    const char* wlane = reinterpret_cast<const char*>(wp) + (long)lane * 8 * ES;
    // The KB bank fragments of an operand are requested TOGETHER (fp32: those of operand i+1 while operand i is mixed) and the
    // scheduling barriers keep loads and multiplies apart: left alone, the scheduler (minimising live registers next to the resident
    // afr[][]) put every load right in front of its multiply - `global_load; s_waitcnt vmcnt(0)` 64 times in a row, one L2 round trip
    // each: ~40 k cycles, half of the fp32 kernel.  (The 16-bit variants sit at the two-waves-per-SIMD register cliff, where any pinning of
    // the schedule costs the second wave: they keep the compiler's order - their loads do overlap in part.)
    if constexpr (ES == 4) {
      typename WL::R wr[2][KB];
      auto ldfrag = [&](int idx, int set) {
        const int mw = idx / KST, ks = idx - mw * KST;
        const int mt = (mt0 + mw) < n_mt ? (mt0 + mw) : (n_mt - 1);
        const char* wbase = wlane + ((long)mt * KST + ks) * 512 * ES;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) wr[set][kb] = WL::load(wbase + (kb < p.K ? kb : 0) * bank_stride);
      };
      ldfrag(0, 0);
#pragma unroll
      for (int idx = 0; idx < MW * KST; ++idx) {
        if (idx + 1 < MW * KST) ldfrag(idx + 1, (idx + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) WL::fma8(wr[idx & 1][kb], al[kb], f);
        afr[idx / KST][idx % KST] = make_a<T>(f);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int mw = 0; mw < MW; ++mw) {
        const int mt = (mt0 + mw) < n_mt ? (mt0 + mw) : (n_mt - 1);
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
          typename WL::R wr[KB];
          const char* wbase = wlane + ((long)mt * KST + ks) * 512 * ES;
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) wr[kb] = WL::load(wbase + (kb < p.K ? kb : 0) * bank_stride);
          float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) WL::fma8(wr[kb], al[kb], f);
          afr[mw][ks] = make_a<T>(f);
        }
      }
    }
  }
  const ActLrelu actf{p.act == ACT_NONE ? 1.f : p.slope};
  float psum[MW][4];
#pragma unroll
  for (int mw = 0; mw < MW; ++mw)
#pragma unroll
    for (int i = 0; i < 4; ++i) psum[mw][i] = 0.f;

  for (int t = 0; t < TL; ++t) {
    const int q0 = (tile0 + t) * NB * 16;
    if (q0 >= p.nq) break;
    xcommit(q0);
    {
      const int qn = q0 + NB * 16;
      xfetch(qn < p.nq && t + 1 < TL ? qn : q0);               // always issued: a branch around it would drain the queue first
    }
    __syncthreads();
    f32x4 acc[MW][NB];
#pragma unroll
    for (int mw = 0; mw < MW; ++mw)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[mw][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // operand double buffer: the B fragments of k-step ks+1 are read from LDS while the MFMAs of k-step ks run; the scheduling
    // barriers keep it that way (the scheduler otherwise sinks each read to just in front of its MFMA: r r wait M M r r wait ...)
    // (not at 256 input channels: with 16 resident A fragments the second operand set costs the second wave per SIMD - 30 -> 40 us)
    constexpr bool DBUF = (CIN <= 128);
    V bfr[DBUF ? 2 : 1][NB];
    auto ldb = [&](int ks, int set) {
      const int chunk = 4 * ks + g, tap = chunk / CPC, c8 = chunk % CPC;
      const int bbase = (-tap - p.shift_lo + col) * RS + c8 * 8 * LES;
#pragma unroll
      for (int n = 0; n < NB; ++n) bfr[set][n] = M::load_bp(xl + bbase + n * 16 * RS, PLANE);
    };
    if constexpr (DBUF) ldb(0, 0);
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
      if constexpr (DBUF) {
        if (ks + 1 < KST) ldb(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        ldb(ks, 0);
      }
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int mw = 0; mw < MW; ++mw) acc[mw][n] = M::mma(afr[mw][ks], bfr[DBUF ? (ks & 1) : 0][n], acc[mw][n]);
      if constexpr (DBUF) __builtin_amdgcn_sched_barrier(0);
    }
    // fp32 storage: a lane's 4 accumulator rows are 4 consecutive output channels = one 16-byte store - straight to HBM, no output
    // tile in LDS and two barriers per tile instead of four (16-bit storage needs the LDS tile to form 16-byte rows)
    constexpr bool DIRECT = (ES == 4);
    if constexpr (!DIRECT) __syncthreads();                    // x tile consumed: reuse the region as the output tile
    char* ol = xl;
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) {
      const int mt = mt0 + mw;
      if (mt < n_mt) {
        const int row = 16 * mt + 4 * g;
        const int r = row / p.Cout;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) { const f32x4 bl = *reinterpret_cast<const f32x4*>(bias_l + (row - R0)); bv[0] = bl[0]; bv[1] = bl[1]; bv[2] = bl[2]; bv[3] = bl[3]; }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const int q = q0 + n * 16 + col;
          const int u = q * p.stride + r - p.pad;
          const bool ok = q < p.nq && u >= 0 && u < p.Tout;
          float ov[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            ov[i] = actf(acc[mw][n][i] + bv[i]);
            if (ok) psum[mw][i] += M::round_store(ov[i]);
          }
          if constexpr (DIRECT) {
            if (ok && row < p.M) M::store4(y + ((long)b * p.Tout + u) * p.Cout + (row - r * p.Cout), ov);
          } else {
            M::store4(ol + ((long)(n * 16 + col)) * ORS + (row - R0) * ES, ov);
          }
        }
      }
    }
    __syncthreads();                                           // (DIRECT: x tile consumed before the next one lands)
    if constexpr (!DIRECT) {
      constexpr int EPC = 16 / ES, CPRO = RW / EPC;
      static_assert(256 % CPRO == 0, "store loop: fixed column per thread");
      const int ch = tid % CPRO, row = R0 + ch * EPC;
      const int r = row / p.Cout, o = row % p.Cout;
      if (row < p.M)
        for (int e = tid / CPRO; e < NB * 16; e += 256 / CPRO) {
          const int q = q0 + e;
          const int u = q * p.stride + r - p.pad;
          if (q >= p.nq || u < 0 || u >= p.Tout) continue;
          const u32x4 val = *reinterpret_cast<const u32x4*>(ol + ((long)e) * ORS + ch * 16);
          *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(y + ((long)b * p.Tout + u) * p.Cout + o)) = val;
        }
      __syncthreads();                                         // output tile drained before the next x tile lands
    }
  }
  if (pooled_out) {                                            // one partial per (row, workgroup): slot blockIdx.x of this sample
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) {
      const int mt = mt0 + mw;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = psum[mw][i];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
        if (col == 0 && mt < n_mt) pooled_out[((long)b * gridDim.x + blockIdx.x) * p.M + 16 * mt + 4 * g + i] = v;
      }
    }
  }
}

template <typename T, int MW, int NB, int CIN, typename TI = T>
static int od_mt_launch(const void* x, const void* wp, const void* bias, const float* alpha, const float* pooled_in, const void* att_w,
                        const void* att_b, void* y, float* pooled_out, OdP p, hipStream_t stream, int* slots_out) {
  constexpr int ES = Mma<T>::ES;
  const int nrows = NB * 16 + 1;
  const size_t xbytes = (size_t)nrows * lds_row_stride(ES == 4 ? 4 * CIN : CIN * ES, 2);
  const size_t obytes = (size_t)NB * 16 * (4 * MW * 16 * ES + 16);
  const size_t lds = sizeof(float) * (OD_MAXK + 4 * MW * 16) + (xbytes > obytes ? xbytes : obytes);
  if (lds > 80 * 1024) return MV_ERR_UNSUPPORTED;
  auto kern = odconv_cl_mt_kernel<T, MW, NB, CIN, 4, TI>;
  static size_t lds_set = 0;
  if (lds > lds_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); lds_set = lds; }
  const int ntl = cdiv(p.nq, NB * 16);
  const int gy = cdiv(p.M / 16, 4 * MW);
  // about two workgroups per CU: enough to overlap each other's barriers, few enough to amortise the per-workgroup weight mix
  static int tgt = -1;
  if (tgt < 0) { const char* e = getenv("MV_OD_MT_WGS"); tgt = e ? atoi(e) : 512; }
  int TL = (int)(((long)ntl * gy * p.B + tgt - 1) / tgt);
  if (TL < 1) TL = 1;
  if (TL > 16) TL = 16;
  dim3 grid(cdiv(ntl, TL), gy, p.B);
  if (grid.y > 65535 || grid.z > 65535) return MV_ERR_UNSUPPORTED;
  if (slots_out) { *slots_out = (int)grid.x; return MV_OK; }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const TI*)x, (const T*)wp, (const T*)bias, alpha,
                     pooled_in, (const T*)att_w, (const T*)att_b, (T*)y, pooled_out, p, TL);
  return MV_OK;
}

// ------------------------------------------------------------------------------------------------ sample-resident variant
// Second upsampler (ODConvTranspose1d 256 -> 128, ks = 2 * stride, 257 columns per sample, 4 x 1 MB of banks), 16-bit storage.
// The multi-tile kernel above keeps a wave's 16 MIXED fragments resident and walks 96-column tiles: per tile it re-stages a 50 KB x
// tile through registers, crosses four barriers and reads every B fragment from LDS once per 16 rows; ablations put its 33 us at
// 16 us of prologue (attention chain + bank mix) + 17 us for three tiles whose matrix work is 1.5 k cycles per wave each.
// Here the WHOLE sample is resident instead: one workgroup = one sample x 128 GEMM rows (8 waves x 16 rows; with 128 output
// channels that is one phase r of the stride, i.e. complete 256-byte output rows), the sample's 258 input rows (t = -1 .. Tin) go
// to LDS once by LDS-DMA while the attention chain runs, and a wave then streams its 16 k-steps: mix the k-step's four bank
// fragments (ring of RD k-steps in flight, started before anything else) into ONE A operand, use it against all 17 column tiles,
// drop it.  No barrier after the prologue, no resident mixed fragments (the registers hold 17 accumulator tiles and the ring).
// The loop is LDS-bound by construction - every B fragment is read once per 16 rows: 8 waves x 17 KB per k-step.
template <typename T, int CIN, int NT>
__global__ __launch_bounds__(512) void odconv_sample_kernel(const T* __restrict__ x, const T* __restrict__ wp, const T* __restrict__ bias,
                                                            const float* __restrict__ alpha_in, const float* __restrict__ pooled_in,
                                                            const T* __restrict__ att_w, const T* __restrict__ att_b, T* __restrict__ y,
                                                            float* __restrict__ pooled_out, OdP p) {
  using M = Mma<T>;
  using V = typename M::V;
  using WL = WLoad<T>;
  constexpr int ES = 2, KB = 4, KST = 2 * CIN / 32, KPT = CIN / 32, RB = CIN * ES, RD = 4;
  static_assert(M::ES == 2 && KST % RD == 0 && RB <= 1024, "16-bit storage, static ring slots, one DMA instruction per row");
  extern __shared__ __align__(16) char lds[];
  float* alds = reinterpret_cast<float*>(lds);                 // [OD_MAXK]
  float* ascr = alds + OD_MAXK;                                // scratch of the attention chain: [max(Cin, 256)]
  char* xl = reinterpret_cast<char*>(ascr + (CIN > 256 ? CIN : 256));
  const int RS = lds_row_stride(RB, 2);
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, g = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, n_mt = p.M / 16;
  const int mt = blockIdx.x * 8 + wid;                         // this wave's M-tile (host: M % 128 == 0)
  const int nrows = p.nq + 1;                                  // input steps shift_lo .. Tin (the first and the last are zeros)
#ifdef MV_OD_TIMING
  long long tmk[8]; int ntm = 0;
#endif
  OD_TM();

  // ---- bank fragments of the first RD k-steps: they depend on nothing
  const long bank_stride = (long)n_mt * KST * 512 * ES;
  const char* wlane = reinterpret_cast<const char*>(wp) + ((long)mt * KST * 512 + lane * 8) * ES;
  typename WL::R ring[RD][KB];
  auto ldring = [&](int ks, int slot) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) ring[slot][kb] = WL::load(wlane + (long)ks * 512 * ES + (kb < p.K ? kb : 0) * bank_stride);
  };
  // ---- wave 7: alpha (odconv.py:36-40), its loads ahead of everything else it requests; waves 0-6: the sample's rows -> LDS
  // (rows outside the sample fail the buffer range check and arrive as zeros)
  const bool fastalpha = !alpha_in && p.K <= 4 && CIN <= 64 * OD_LU;
  if (wid == 7 && fastalpha) {
    od_alpha_wave<T>(alds, b, p, pooled_in, att_w, att_b);
#pragma unroll
    for (int i = 0; i < RD; ++i) ldring(i, i);
  } else {
    typedef __attribute__((address_space(3))) void lds_void;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(x)) + (long)b * p.Tin * RB, 0, p.Tin * RB, 0x00020000);
    const int nst = fastalpha ? 7 : 8;
    for (int i = wid; i < nrows; i += nst) {
      const int tin = p.shift_lo + i;
      const unsigned gbase = (tin >= 0 && tin < p.Tin) ? (unsigned)(tin * RB) : 0x7ffffff0u;
      if (lane * 16 < RB) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(xl + i * RS), 16, gbase + lane * 16, 0, 0, 0);
    }
    // the ring BEHIND the rows: the barrier below waits for everything but these RD * KB youngest loads
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RD; ++i) ldring(i, i);
    __builtin_amdgcn_sched_barrier(0);
  }
  OD_TM();
  if (!fastalpha) {
    if (alpha_in) {
      if (tid < p.K) alds[tid] = alpha_in[(long)b * p.K + tid];
    } else {
      od_alpha_from_partials<T, 512>(alds, ascr, 1, b, p, pooled_in, att_w, att_b);
    }
  }
  OD_TM();
  static_assert(RD * KB == 16, "the counted wait below");
  if (fastalpha) { asm volatile("s_waitcnt vmcnt(16)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // rows landed (and alds written); ring in flight
  else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  OD_TM();
  float al[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) al[kb] = kb < p.K ? alds[kb] : 0.f;

  const int row = 16 * mt + 4 * g;
  const int r = row / p.Cout, o = row - r * p.Cout;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};                          // alpha-mixed bias of this lane's 4 channels (requested before the loop)
  if (bias) {
    float braw[KB][4];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) M::load4(bias + (long)(kb < p.K ? kb : 0) * p.Cout + o, braw[kb]);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) bv[i] += al[kb] * braw[kb][i];
  }
  // per-lane LDS byte offset of column tile n, tap 1 (input step q - 1 = row q); tap 0 is one row further.  Columns past nq are
  // computed on clamped rows and never stored.
  unsigned lb[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    int r = n * 16 + col;
    r = r < nrows - 2 ? r : nrows - 2;
    lb[n] = (unsigned)(r * RS + g * 16);
  }
  f32x4 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KST; ++ks) {
    float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) WL::fma8(ring[ks % RD][kb], al[kb], f);
    const V a = make_a<T>(f);
    if (ks + RD < KST) ldring(ks + RD, ks % RD);
    const int tap = ks / KPT;                                  // k-chunk 4 ks + g = tap * Cin/8 + c8
    const int koff = (tap ? 0 : RS) + (ks % KPT) * 64;
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = M::mma(a, M::load_b(xl + lb[n] + koff), acc[n]);
  }

  OD_TM();
  // ---- epilogue: bias, activation, 8-byte stores (a lane's 4 rows are 4 consecutive channels), channel sums for the next layer
  const ActLrelu actf{p.act == ACT_NONE ? 1.f : p.slope};
  float psum[4] = {0.f, 0.f, 0.f, 0.f};
  T* yb = y + (long)b * p.Tout * p.Cout + o;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int q = n * 16 + col;
    const int u = q * p.stride + r - p.pad;
    const bool ok = q < p.nq && u >= 0 && u < p.Tout;
    float ov[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ov[i] = actf(acc[n][i] + bv[i]);
      psum[i] += ok ? M::round_store(ov[i]) : 0.f;
    }
    if (ok) M::store4(yb + (long)u * p.Cout, ov);
  }
  if (pooled_out) {                                            // one workgroup covers the sample's columns: one slot
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = psum[i];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
      psum[i] = v;
    }
    if (col == 0) *reinterpret_cast<f32x4*>(pooled_out + (long)b * p.M + row) = f32x4{psum[0], psum[1], psum[2], psum[3]};
  }
#ifdef MV_OD_TIMING
  OD_TM();
  if (tid == 0 && od_dbg) {
    const long wgid = (long)blockIdx.y * gridDim.x + blockIdx.x;
    if (wgid < 8192) for (int i = 0; i < 8; ++i) od_dbg[wgid * 8 + i] = i < ntm ? tmk[i] - tmk[0] : -1;
  }
#endif
}

template <typename T, int CIN, int NT>
static int od_sample_launch(const void* x, const void* wp, const void* bias, const float* alpha, const float* pooled_in, const void* att_w,
                            const void* att_b, void* y, float* pooled_out, OdP p, hipStream_t stream, int* slots_out) {
  if (!p.transposed || p.ntaps != 2 || p.K > 4 || p.Cin != CIN || p.M % 128 || p.nq > NT * 16 || p.nq <= (NT - 1) * 16 ||
      p.shift_lo != -1 || p.act > ACT_LRELU || (long)p.Tin * CIN * 2 > 0x7fff0000L)
    return MV_ERR_UNSUPPORTED;
  const size_t lds = sizeof(float) * (OD_MAXK + (CIN > 256 ? CIN : 256)) + (size_t)(p.nq + 1) * lds_row_stride(CIN * 2, 2);
  if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
  auto kern = odconv_sample_kernel<T, CIN, NT>;
  static size_t lds_set = 0;
  if (lds > lds_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); lds_set = lds; }
  dim3 grid(p.M / 128, p.B);
  if (grid.y > 65535) return MV_ERR_UNSUPPORTED;
  if (slots_out) { *slots_out = 1; return MV_OK; }
#ifdef MV_OD_TIMING
  static long long* dbg = nullptr;
  static int calls = 0;
  if (!dbg) { hipMalloc(&dbg, 8192 * 8 * 8); hipMemcpyToSymbol(HIP_SYMBOL(od_dbg), &dbg, sizeof(dbg)); }
  hipMemsetAsync(dbg, 0xff, 8192 * 8 * 8, stream);
#endif
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, (const T*)x, (const T*)wp, (const T*)bias, alpha, pooled_in,
                     (const T*)att_w, (const T*)att_b, (T*)y, pooled_out, p);
#ifdef MV_OD_TIMING
  if (++calls == 20) {
    hipStreamSynchronize(stream);
    static long long hbuf[8192 * 8];
    hipMemcpy(hbuf, dbg, sizeof(hbuf), hipMemcpyDeviceToHost);
    const long nwg = (long)grid.x * grid.y;
    double avg[8] = {0}; int cnt[8] = {0};
    for (long w = 0; w < nwg; ++w) for (int i = 0; i < 8; ++i) { long long v = hbuf[w * 8 + i]; if (v >= 0) { avg[i] += (double)v; cnt[i]++; } }
    fprintf(stderr, "[sample timing] grid %u x %u marks (start, issued, alpha, staged, loop, end):", grid.x, grid.y);
    for (int i = 0; i < 8; ++i) if (cnt[i]) fprintf(stderr, " %.0f", avg[i] / cnt[i]);
    fprintf(stderr, "\n");
  }
#endif
  return MV_OK;
}

// ------------------------------------------------------------------------------------------------ K-loop variant
// Short sequences with big kernel banks (the first upsampler: 33 columns per sample, 16.8 MB of banks): aggregating the
// per-sample kernel costs more than the convolution itself.  Here the banks are used AS STORED - exactly the reference's
// loop over k (odconv.py:187-204): D_k = W_k * x accumulates in a scratch accumulator, out += alpha[b,k] * D_k once per
// bank.  4x the MFMA work of the aggregated form, but zero aggregation VALU and every weight fragment is loaded once per
// S samples (weights-stationary), PD k-steps ahead of its use.
// A wave owns MW M-tiles: every B fragment read from LDS feeds MW MFMAs (with one M-tile per wave the kernel issues one
// ds_read_b128 per MFMA, which saturates the LDS array at 4 SIMDs x 4 cycles per 16-cycle MFMA).  The per-bank scratch
// accumulators are folded into the output accumulators (out += alpha * D_k) at the end of each bank and reused.
// fp32 storage: the x tiles are PRE-SPLIT into hi / lo bf16 planes when they are staged, weight fragments (fp32 in the packed
// image) are split once per fragment in registers, every product is hi*hi + hi*lo + lo*hi (mfma.h).
template <typename T> struct KlW;     // one packed A fragment of this lane: raw load + conversion to the MFMA operand
template <> struct KlW<bf16> {
  using R = Mma<bf16>::V;
  static __device__ __forceinline__ R load(const char* p) { return Mma<bf16>::load_b(p); }
  static __device__ __forceinline__ Mma<bf16>::V op(const R& r) { return r; }
};
template <> struct KlW<f16> {
  using R = Mma<f16>::V;
  static __device__ __forceinline__ R load(const char* p) { return Mma<f16>::load_b(p); }
  static __device__ __forceinline__ Mma<f16>::V op(const R& r) { return r; }
};
template <> struct KlW<float> {
  struct R { f32x4 a, b; };
  static __device__ __forceinline__ R load(const char* p) {
    R r; r.a = reinterpret_cast<const f32x4*>(p)[0]; r.b = reinterpret_cast<const f32x4*>(p)[1]; return r;
  }
  static __device__ __forceinline__ Mma<float>::V op(const R& r) {
    const float f[8] = {r.a[0], r.a[1], r.a[2], r.a[3], r.b[0], r.b[1], r.b[2], r.b[3]};
    return Mma<float>::split(f);
  }
};

template <typename T, int S, int NB, int KB, int MW>
__global__ __launch_bounds__(256) void odconv_kloop_kernel(const T* __restrict__ x, const T* __restrict__ wp,
                                                           const T* __restrict__ bias, const float* __restrict__ alpha_in,
                                                           const float* __restrict__ pooled_in, const T* __restrict__ att_w,
                                                           const T* __restrict__ att_b, T* __restrict__ y,
                                                           float* __restrict__ pooled_out, OdP p) {
  using M = Mma<T>;
  using V = typename M::V;
  using KW = KlW<T>;
  constexpr int ES = M::ES;
  constexpr bool SPLIT = (ES == 4);
  constexpr int LES = 2;                           // operand element size in LDS (fp32 storage: hi / lo bf16 planes)
  extern __shared__ __align__(16) char lds[];
  float* alds = reinterpret_cast<float*>(lds);
  float* ascr = alds + S * OD_MAXK;                // scratch of the attention prologue: its own region, so that it runs under the staging
  char* xl = reinterpret_cast<char*>(ascr + S * (p.Cin > 256 ? p.Cin : 256));
  const int PLANE = p.Cin * LES;                   // byte offset of the lo plane inside a row (SPLIT)
  const int RS = lds_row_stride(SPLIT ? 2 * PLANE : p.Cin * ES, LES);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int mt0 = (blockIdx.x * 4 + wid) * MW;     // first M-tile of this wave
  const int b0 = blockIdx.y * S;
  const int n_mt = p.M / 16;
  const int ZR = p.nrows - 1;                      // index of the all-zero row
#ifdef MV_OD_TIMING
  long long tmk[8]; int ntm = 0;
#endif
  OD_TM();

  // ---- 16-bit storage: the whole (short) input of the S samples goes to LDS by LDS-DMA, one 1 KB piece of a row per instruction, all
  // of them issued before anything else (no register round trip, no wait between batches); rows outside the sample and the zero row
  // fail the buffer range check and arrive as zeros.  In-kernel marks of the register-staged form at the first upsampler: attention
  // prologue 25.5 k ticks, THEN staging 14.1 k (three dependent batches), main loop 32.6 k, epilogue 8.4 k.
  constexpr bool XDMA = !SPLIT;
#ifdef MV_OD_TIMING
  const int kdbg = p.dbg;
#else
  constexpr int kdbg = 0;
#endif
  // Attention on one wave per sample (the last S of the four waves; od_alpha_wave): no barrier, one round trip, requested ahead of the
  // staging traffic that the other waves issue.
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const bool wavealpha = XDMA && !alpha_in && !kdbg && S <= 2 && p.K <= 4 && p.Cin <= 64 * OD_LU;
  const int nstw = wavealpha ? 4 - S : 4;          // staging waves
  if (wavealpha && wu >= nstw) {
    const int s = wu - nstw;
    if (b0 + s < p.B) od_alpha_wave<T>(alds + s * OD_MAXK, b0 + s, p, pooled_in, att_w, att_b);
    else if (lane < OD_MAXK) alds[s * OD_MAXK + lane] = 0.f;
  } else if (XDMA && !(kdbg & 1)) {
    typedef __attribute__((address_space(3))) void lds_void;
    const int RB = p.Cin * ES;                     // bytes of one row
    const long xbytes = (long)p.B * p.Tin * RB;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(x)), 0,
                                                                         (int)(xbytes < 0x7fffffffL ? xbytes : 0x7fffffffL), 0x00020000);
    const int npc = (RB + 1023) >> 10;
    for (int i = wu; i < S * p.nrows; i += nstw) {
      const int s = i / p.nrows, r = i - s * p.nrows;
      const int tin = p.shift_lo + r;
      const bool ok = r < ZR && b0 + s < p.B && tin >= 0 && tin < p.Tin;
      const unsigned gbase = ok ? (unsigned)(((long)(b0 + s) * p.Tin + tin) * RB) : 0x7ffffff0u;      // (host: B * Tin * RB < 2^31 - 16 K)
      for (int pc = 0; pc < npc; ++pc)
        if (pc * 1024 + lane * 16 < RB)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(xl + i * RS + pc * 1024), 16, gbase + pc * 1024 + lane * 16, 0, 0, 0);
    }
  }
  // ---- alpha (odconv.py:36-40)
  if (wavealpha) {
    OD_TM(); OD_TM(); OD_TM(); OD_TM();
  } else if (alpha_in) {
    if (tid < S * p.K) {
      const int s = tid / p.K, kb = tid % p.K;
      alds[s * OD_MAXK + kb] = (b0 + s < p.B) ? alpha_in[(long)(b0 + s) * p.K + kb] : 0.f;
    }
  } else if (kdbg & 2) {
    if (tid < S * p.K) alds[(tid / p.K) * OD_MAXK + tid % p.K] = 0.25f;
    OD_TM(); OD_TM(); OD_TM(); OD_TM();
  } else {
#ifdef MV_OD_TIMING
    OD_TM();
    od_alpha_from_partials<T>(alds, ascr, S, b0, p, pooled_in, att_w, att_b, tmk + ntm);
    ntm += 3;
#else
    od_alpha_from_partials<T>(alds, ascr, S, b0, p, pooled_in, att_w, att_b);
#endif
  }
  OD_TM();
  // ---- stage the whole (short) input of every sample: row r <-> input step r + shift_lo, last row = zeros
  {
    const int cpr = p.Cin * ES / 16;               // 16-byte global pieces per row
    const int per = p.nrows * cpr;
    if constexpr (SPLIT) {
      constexpr int UB = 8;
      for (int i0 = tid; i0 < S * per; i0 += 256 * UB) {
        f32x4 v[UB];
        int d[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int i = i0 + u * 256;
          v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          d[u] = -1;
          if (i < S * per) {
            const int s = i / per, rem = i - s * per;
            const int r = rem / cpr, ch = rem - r * cpr;
            const int tin = p.shift_lo + r;
            d[u] = (s * p.nrows + r) * RS + ch * 8;
            if (r < ZR && b0 + s < p.B && tin >= 0 && tin < p.Tin)
              v[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(x + ((long)(b0 + s) * p.Tin + tin) * p.Cin) + ch * 16);
          }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
          if (d[u] >= 0) {
            u32x2 hi, lo;
            Mma<float>::split4(v[u], hi, lo);
            *reinterpret_cast<u32x2*>(xl + d[u]) = hi;
            *reinterpret_cast<u32x2*>(xl + d[u] + PLANE) = lo;
          }
      }
    }
  }
  if constexpr (XDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  OD_TM();

  // out = sum_kb alpha[b,kb] * D_kb; D_kb lives in `acc` while bank kb streams and is folded into `out` at the end of the bank
  f32x4 acc[MW][S][NB], out[MW][S][NB];
#pragma unroll
  for (int mw = 0; mw < MW; ++mw)
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int n = 0; n < NB; ++n) { acc[mw][s][n] = f32x4{0.f, 0.f, 0.f, 0.f}; out[mw][s][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const long bank_stride = (long)n_mt * p.ksteps * 512 * ES;
  const char* wlane[MW];
#pragma unroll
  for (int mw = 0; mw < MW; ++mw) {
    const int mtc = (mt0 + mw) < n_mt ? (mt0 + mw) : n_mt - 1;
    wlane[mw] = reinterpret_cast<const char*>(wp) + ((long)mtc * p.ksteps * 512 + lane * 8) * ES;
  }
  // PD-deep register ring with STATIC slots (ksteps % PD == 0): the wave covers the L2 latency of its weight stream itself.
  // Everything that steers the stream is decided once per CHUNK of PD k-steps and is wave-uniform (scalar): the chunk's
  // fragments are PD consecutive 512-element blocks of one bank, the prefetch target PD k-steps ahead is either the next chunk
  // of this bank or the first chunk of the next bank (past the end: this chunk again, unused), and - the host checks
  // (Cin/32) % PD == 0 - a chunk never straddles two taps, so the B-operand rows are fixed per chunk too.  Inside a chunk the
  // k-steps differ by compile-time offsets only: no branch, no integer arithmetic per k-step (the earlier per-k-step cursor
  // compiled to ~45 scalar / address instructions and a branch around 6 MFMAs: issue-bound at 4x the matrix-pipe time).
  constexpr int PD = SPLIT ? 4 : 8;
  constexpr long FRAG = 512 * ES;                   // bytes between consecutive k-step fragments of one M-tile
  typename KW::R ring[PD][MW];
#pragma unroll
  for (int j = 0; j < PD; ++j)
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) ring[j][mw] = KW::load(wlane[mw] + j * FRAG);
  // per-lane LDS byte offsets of the B operand for each of the (at most two) taps and column tiles, clamped to the zero row
  const int spt = p.Cin / 32;                        // k-steps per tap (multiple of PD)
  const int sample_stride = p.nrows * RS;
  unsigned lbase[2][NB];
#pragma unroll
  for (int tp = 0; tp < 2; ++tp)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int shift = p.transposed ? -tp : (tp * p.dil - p.pad);
      int r = col - p.shift_lo + n * 16 + shift;
      r = r < ZR ? r : ZR;
      lbase[tp][n] = (unsigned)(r * RS + g * 8 * LES);
    }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb < p.K) {
      for (int ks0 = 0; ks0 < p.ksteps; ks0 += PD) {
        // prefetch target of this chunk (scalar): next chunk of this bank / first chunk of the next bank / this chunk again
        const bool last_chunk = ks0 + PD >= p.ksteps;
        const bool last_bank = kb + 1 >= p.K;
        const long pf_off = last_chunk ? (last_bank ? (long)kb * bank_stride + (long)ks0 * FRAG : (long)(kb + 1) * bank_stride)
                                       : (long)kb * bank_stride + (long)(ks0 + PD) * FRAG;
        const int tap = ks0 >= spt ? 1 : 0;
        const unsigned koff0 = (unsigned)((ks0 - tap * spt) * 32 * LES);
        unsigned boff[S][NB];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int n = 0; n < NB; ++n) boff[s][n] = (tap ? lbase[1][n] : lbase[0][n]) + (unsigned)(s * sample_stride) + koff0;
        // B fragments of k-step j+1 are read while the MFMAs of k-step j run (two register sets; the scheduling barrier at the end
        // of a k-step would otherwise hold every k-step's reads behind the previous k-step's MFMAs: LDS round trip + MFMAs in series)
        constexpr bool BDB = (ES == 2);                // (fp32 operands are register pairs: a second set costs the second wave)
        V bfr[BDB ? 2 : 1][S][NB];
        auto ldb = [&](int j, int set) {
#pragma unroll
          for (int s = 0; s < S; ++s)
#pragma unroll
            for (int n = 0; n < NB; ++n) bfr[set][s][n] = M::load_bp(xl + boff[s][n] + j * 32 * LES, PLANE);
        };
        if constexpr (BDB) ldb(0, 0);
#pragma unroll
        for (int j = 0; j < PD; ++j) {
          V a0[MW];
#pragma unroll
          for (int mw = 0; mw < MW; ++mw) a0[mw] = KW::op(ring[j][mw]);
          if constexpr (BDB) { if (j + 1 < PD) ldb(j + 1, (j + 1) & 1); }
          else ldb(j, 0);
          if constexpr (BDB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < S; ++s)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
              for (int mw = 0; mw < MW; ++mw) acc[mw][s][n] = M::mma(a0[mw], bfr[BDB ? (j & 1) : 0][s][n], acc[mw][s][n]);
          // refill this ring slot for the next chunk NOW (it is consumed PD k-steps from here).  Left to itself the scheduler
          // sinks all PD loads to the end of the chunk - right in front of their first use - and every chunk then starts by
          // waiting out a full L2 round trip; the scheduling barrier pins the load behind this k-step's MFMAs.
#pragma unroll
          for (int mw = 0; mw < MW; ++mw) ring[j][mw] = KW::load(wlane[mw] + pf_off + j * FRAG);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // fold this bank into the output accumulators and clear the scratch set
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float al = alds[s * OD_MAXK + kb];
#pragma unroll
        for (int mw = 0; mw < MW; ++mw)
#pragma unroll
          for (int n = 0; n < NB; ++n) {
#pragma unroll
            for (int i = 0; i < 4; ++i) out[mw][s][n][i] += al * acc[mw][s][n][i];
            acc[mw][s][n] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
    }
  }

  OD_TM();
  // ---- epilogue: bias, activation -> LDS tile [s][q][rows of this workgroup] -> whole-row stores
  __syncthreads();
  constexpr int RW = 64 * MW;
  constexpr int ORS = RW * ES + 16;
  char* ol = xl;
  const int R0 = blockIdx.x * RW;
  auto epilogue = [&](auto actf, auto pool_c) {      // one body per (activation, pooling): no per-element branch trees
    constexpr bool HAS_POOL = decltype(pool_c)::value;
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) {
      const int mt = mt0 + mw;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int b = b0 + s;
        float rowsum[4] = {0.f, 0.f, 0.f, 0.f};
        if (mt < n_mt && b < p.B) {
          const int row = 16 * mt + 4 * g;
          const int r = p.transposed ? row / p.Cout : 0;
          const int o = p.transposed ? row % p.Cout : row;
          float bv[4] = {0.f, 0.f, 0.f, 0.f};
          if (bias) {                                   // bank biases of these 4 channels: independent loads, then the alpha mix
            float braw[KB][4];
#pragma unroll
            for (int k2 = 0; k2 < KB; ++k2) M::load4(bias + (long)(k2 < p.K ? k2 : 0) * p.Cout + o, braw[k2]);
#pragma unroll
            for (int k2 = 0; k2 < KB; ++k2)
#pragma unroll
              for (int i = 0; i < 4; ++i) bv[i] += (k2 < p.K ? alds[s * OD_MAXK + k2] : 0.f) * braw[k2][i];
          }
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const int q = n * 16 + col;
            const int u = p.transposed ? q * p.stride + r - p.pad : q;
            const bool ok = q < p.nq && u >= 0 && u < p.Tout;
            float ov[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              ov[i] = actf(out[mw][s][n][i] + bv[i]);
              if (HAS_POOL && ok) rowsum[i] += M::round_store(ov[i]);
            }
            M::store4(ol + ((long)(s * NB * 16 + n * 16 + col)) * ORS + (row - R0) * ES, ov);
          }
        }
        if (HAS_POOL) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = rowsum[i];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
            rowsum[i] = v;
          }
          if (col == 0 && mt < n_mt && b < p.B) {   // the K-loop kernel covers a sample's columns in one workgroup: one slot
            const int row = 16 * mt + 4 * g;
            *reinterpret_cast<f32x4*>(pooled_out + (long)b * p.M + row) = f32x4{rowsum[0], rowsum[1], rowsum[2], rowsum[3]};
          }
        }
      }
    }
  };
  {
    const ActLrelu al_{p.act == ACT_NONE ? 1.f : p.slope};
    const ActAny aa_{p.act, p.slope};
    if (pooled_out) { if (p.act <= ACT_LRELU) epilogue(al_, std::true_type{}); else epilogue(aa_, std::true_type{}); }
    else { if (p.act <= ACT_LRELU) epilogue(al_, std::false_type{}); else epilogue(aa_, std::false_type{}); }
  }
  __syncthreads();
  {
    constexpr int EPC = 16 / ES;
    constexpr int CPR = RW / EPC;
    static_assert(256 % CPR == 0, "store loop: fixed column per thread");
    const int ch = tid % CPR, row = R0 + ch * EPC;
    const int r = p.transposed ? row / p.Cout : 0;
    const int o = p.transposed ? row % p.Cout : row;
    if (row < p.M)
      for (int e = tid / CPR; e < S * NB * 16; e += 256 / CPR) {
        const int s = e / (NB * 16), qi = e - s * (NB * 16);
        const int b = b0 + s;
        const int u = p.transposed ? qi * p.stride + r - p.pad : qi;
        if (b >= p.B || qi >= p.nq || u < 0 || u >= p.Tout) continue;
        const u32x4 val = *reinterpret_cast<const u32x4*>(ol + ((long)e) * ORS + ch * 16);
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(y + ((long)b * p.Tout + u) * p.Cout + o)) = val;
      }
  }
#ifdef MV_OD_TIMING
  OD_TM();
  if (tid == 0 && od_dbg) {
    const long wgid = (long)blockIdx.y * gridDim.x + blockIdx.x;
    if (wgid < 8192) for (int i = 0; i < 8; ++i) od_dbg[wgid * 8 + i] = i < ntm ? tmk[i] - tmk[0] : -1;
  }
#endif
}

template <typename T, int S, int NB, int MW>
static int od_kloop_launch(const void* x, const void* wp, const void* bias, const float* alpha, const float* pooled_in,
                           const void* att_w, const void* att_b, void* y, float* pooled_out, OdP p, hipStream_t stream,
                           int* slots_out) {
  using M = Mma<T>;
  constexpr int PD = M::ES == 4 ? 4 : 8;
  if (p.nq > NB * 16 || p.K > 4 || p.Cin % 32 || (p.Cin / 32) % PD || p.ksteps % PD || p.nchunks != 4 * p.ksteps || p.ntaps > 2)
    return MV_ERR_UNSUPPORTED;
  // rows: every input step that any column can touch (shift_lo .. Tin-1 shifted) + one zero row
  p.nrows = p.Tin - p.shift_lo + 1;
  const size_t xbytes = (size_t)S * p.nrows * lds_row_stride(M::ES == 4 ? 4 * p.Cin : p.Cin * M::ES, 2);
  const size_t obytes = (size_t)S * NB * 16 * (64 * MW * M::ES + 16);
  const size_t lds = sizeof(float) * (S * OD_MAXK + S * (p.Cin > 256 ? p.Cin : 256)) + (xbytes > obytes ? xbytes : obytes);
  if (lds > 160 * 1024 || (long)p.B * p.Tin * p.Cin * M::ES > 0x7fff0000L) return MV_ERR_UNSUPPORTED;
  auto kern = odconv_kloop_kernel<T, S, NB, 4, MW>;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_set = lds;
  }
  dim3 grid(cdiv(p.M / 16, 4 * MW), cdiv(p.B, S));
  if (grid.y > 65535) return MV_ERR_UNSUPPORTED;
  if (slots_out) { *slots_out = 1; return MV_OK; }
#ifdef MV_OD_TIMING
  static long long* dbg = nullptr;
  static int calls = 0;
  if (!dbg) { hipMalloc(&dbg, 8192 * 8 * 8); hipMemcpyToSymbol(HIP_SYMBOL(od_dbg), &dbg, sizeof(dbg)); }
  hipMemsetAsync(dbg, 0xff, 8192 * 8 * 8, stream);
  { const char* e = getenv("MV_KL_DBG"); p.dbg = e ? atoi(e) : 0; }
#endif
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const T*)x, (const T*)wp, (const T*)bias, alpha, pooled_in,
                     (const T*)att_w, (const T*)att_b, (T*)y, pooled_out, p);
#ifdef MV_OD_TIMING
  if (++calls == 20) {
    hipStreamSynchronize(stream);
    static long long hbuf[8192 * 8];
    hipMemcpy(hbuf, dbg, sizeof(hbuf), hipMemcpyDeviceToHost);
    const long nwg = (long)grid.x * grid.y < 8192 ? (long)grid.x * grid.y : 8192;
    double avg[8] = {0}; int cnt[8] = {0};
    for (long w = 0; w < nwg; ++w) for (int i = 0; i < 8; ++i) { long long v = hbuf[w * 8 + i]; if (v >= 0) { avg[i] += (double)v; cnt[i]++; } }
    fprintf(stderr, "[kloop timing] S %d MW %d grid %u x %u marks (start, dma issued, sums, barrier, logits, alpha, staged, loop | end):", S, MW, grid.x, grid.y);
    for (int i = 0; i < 8; ++i) if (cnt[i]) fprintf(stderr, " %.0f", avg[i] / cnt[i]);
    fprintf(stderr, "\n");
  }
#endif
  return MV_OK;
}

template <typename T, int S, int MW, int NB, bool PFW, int KB>
static int od_launch(const void* x, const void* wp, const void* bias, const float* alpha, const float* pooled_in,
                     const void* att_w, const void* att_b, const void* film, void* y, float* pooled_out, OdP p,
                     hipStream_t stream, int* slots_out) {
  using M = Mma<T>;
  p.nrows = NB * 16 + (p.ntaps - 1) * (p.transposed ? 1 : p.dil);
  const size_t xbytes = (size_t)S * p.nrows * lds_row_stride(M::ES == 4 ? 4 * p.Cin : p.Cin * M::ES, 2);
  const size_t obytes = (size_t)S * NB * 16 * (4 * MW * 16 * M::ES + 16);    // staged output tile reuses the x region
  const size_t lds = sizeof(float) * (S * OD_MAXK + S * 4 * MW * 16) + (xbytes > obytes ? xbytes : obytes);
  if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
  auto kern = odconv_cl_kernel<T, S, MW, NB, (PFW && (M::ES == 2 || MW * NB <= 3)), KB>;   // fp32: prefetch only in the small (input_proj) instantiation
  static size_t lds_set = 0;
  if (lds > lds_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_set = lds;
  }
  dim3 grid(cdiv(p.nq, NB * 16), cdiv(p.M / 16, 4 * MW), cdiv(p.B, S));
  if (grid.y > 65535 || grid.z > 65535) return MV_ERR_UNSUPPORTED;
  if (slots_out) { *slots_out = (int)grid.x; return MV_OK; }
#ifdef MV_OD_TIMING
  static long long* dbg = nullptr;
  static int calls = 0;
  if (!dbg) { hipMalloc(&dbg, 8192 * 8 * 8); hipMemcpyToSymbol(HIP_SYMBOL(od_dbg), &dbg, sizeof(dbg)); }
  hipMemsetAsync(dbg, 0xff, 8192 * 8 * 8, stream);
#endif
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const T*)x, (const T*)wp, (const T*)bias, alpha, pooled_in,
                     (const T*)att_w, (const T*)att_b, (const T*)film, (T*)y, pooled_out, p);
#ifdef MV_OD_TIMING
  if (++calls == 20) {
    hipStreamSynchronize(stream);
    static long long hbuf[8192 * 8];
    hipMemcpy(hbuf, dbg, sizeof(hbuf), hipMemcpyDeviceToHost);
    const long nwg = (long)grid.x * grid.y * grid.z < 8192 ? (long)grid.x * grid.y * grid.z : 8192;
    double avg[8] = {0}; int cnt[8] = {0};
    for (long w = 0; w < nwg; ++w) for (int i = 0; i < 8; ++i) { long long v = hbuf[w * 8 + i]; if (v >= 0) { avg[i] += (double)v; cnt[i]++; } }
    fprintf(stderr, "[od timing] Cin %d M %d ksteps %d grid %u x %u x %u marks:", p.Cin, p.M, p.ksteps, grid.x, grid.y, grid.z);
    for (int i = 0; i < 8; ++i) if (cnt[i]) fprintf(stderr, " %.0f", avg[i] / cnt[i]);
    fprintf(stderr, "\n");
  }
#endif
  return MV_OK;
}

static bool od_make(OdP* p, int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad, int dil,
                    int transposed, int K, int act, float slope, int film_F) {
  if (B <= 0 || Cin <= 0 || Cin % 8 || Cout <= 0 || Cout % 8 || Tin <= 0 || ks <= 0 || K < 1 || K > OD_MAXK) return false;
  p->B = B; p->Cin = Cin; p->Tin = Tin; p->Cout = Cout; p->Tout = Tout; p->ks = ks; p->stride = stride; p->pad = pad;
  p->dil = dil; p->transposed = transposed; p->K = K; p->act = act; p->slope = slope; p->film_F = film_F;
  if (transposed) {
    if (dil != 1 || stride < 1 || ks % stride || pad < 0) return false;
    p->ntaps = ks / stride;
    p->M = stride * Cout;
    const int full = (Tin - 1) * stride - 2 * pad + ks;
    if (Tout < full || Tout >= full + stride) return false;
    p->nq = Tin + p->ntaps - 1 + (Tout > full ? 1 : 0);   // output_padding rows come from one more (zero-input) column
    p->shift_lo = -(p->ntaps - 1);
  } else {
    if (stride != 1 || dil < 1 || pad < 0) return false;
    if (Tout != Tin + 2 * pad - dil * (ks - 1)) return false;
    p->ntaps = ks;
    p->M = Cout;
    p->nq = Tout;
    p->shift_lo = -pad;
  }
  if (p->M % 16) return false;
  p->nchunks = p->ntaps * (Cin / 8);
  p->ksteps = cdiv(p->nchunks, 4);
  p->nrows = 0;
  p->pool_n = Cin;
  p->in_f16 = 0;
  p->out_pair = 0;
  return true;
}

}  // namespace mv

using namespace mv;

extern "C" size_t mv_odconv_cl_packed_bytes(int Cin, int Cout, int ks, int stride, int transposed, int K, int dtype) {
  OdP p;
  const int Tin = 4;
  const int Tout = transposed ? (Tin - 1) * stride + ks : Tin;   // any consistent length: only M/ksteps matter here
  if (!od_make(&p, 1, Cin, Tin, Cout, Tout, ks, stride, transposed ? 0 : 0, 1, transposed, K, 0, 0.f, 0)) {
    // regular conv with pad 0 needs Tout = Tin - (ks-1): retry with a consistent pair
    if (transposed || !od_make(&p, 1, Cin, ks + 3, Cout, 4, ks, 1, 0, 1, 0, K, 0, 0.f, 0)) return 0;
  }
  const size_t es = dtype == MV_F32 ? 4 : 2;
  return (size_t)K * (p.M / 16) * p.ksteps * 512 * es;
}

extern "C" int mv_odconv_cl_pack(const void* kernels, int param_dtype, void* packed, int Cin, int Cout, int ks,
                                 int stride, int transposed, int K, int dtype, void* stream) {
  MV_CHECK_ARG(kernels && packed);
  OdP p;
  const int Tin = ks + 3;
  const int Tout = transposed ? (Tin - 1) * stride + ks : Tin - (ks - 1);
  if (!od_make(&p, 1, Cin, Tin, Cout, Tout, ks, stride, 0, 1, transposed, K, 0, 0.f, 0)) return MV_ERR_UNSUPPORTED;
  const long total = (long)K * (p.M / 16) * p.ksteps * 512;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  MV_DISPATCH(dtype, {
    switch (param_dtype) {
      case MV_F32: hipLaunchKernelGGL((odconv_pack_kernel<T, float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)kernels, (T*)packed, p); break;
      case MV_BF16: hipLaunchKernelGGL((odconv_pack_kernel<T, bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)kernels, (T*)packed, p); break;
      case MV_F16: hipLaunchKernelGGL((odconv_pack_kernel<T, f16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16*)kernels, (T*)packed, p); break;
      default: return MV_ERR_DTYPE;
    }
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}

// ------------------------------------------------------------------------------------------------ streaming variant
// The last two upsamplers in fp32 storage (ODConvTranspose1d, ks = 2 * stride, 256-byte input rows: 64 fp32 channels or 128 fp16
// channels, small banks, long sequences).  The multi-tile kernel above moves 1.5-2.9 TB/s here: four waves share an x tile between
// two barriers per 64 columns, each wave redoes commit -> barrier -> MFMAs -> stores in step with the others.  This form follows the
// streaming MRF passes (mrf_stream.hip):
//   * one workgroup of 8 waves per CU; the sample's aggregated kernel sum_k alpha_k W_k is formed ONCE per workgroup (each wave mixes
//     its share of the bank fragments) and kept in LDS as f16 hi + lo fragment images, scaled by 2^6 so the lo parts stay clear of
//     the f16 subnormal floor (the matrix pipe keeps f16 subnormals - measured - the scale is a margin, and it is exact);
//   * every wave owns a span of L input columns and walks it in 16-column tiles from a PRIVATE two-slot ring (+ a one-row carry for
//     the second tap): no workgroup barrier after the prologue; the rows of tiles j + 1 and j + 2 are in flight in registers;
//   * fp16 rows are exact f16 operands (a_hi x + a_lo x: two products); fp32 rows are split into f16 hi + lo as they are committed
//     (three products); out-of-range rows arrive as zeros and out-of-range output rows are dropped by the buffer range check.
// A wave writes all M rows of its columns (M / Cout whole output rows per input column).
constexpr int OS_NW = 8, OS_RB = 256, OS_RS = OS_RB + 16, OS_SLOT = 16 * OS_RS, OS_RING = 2 * OS_SLOT + OS_RS;
constexpr float OS_WSCALE = 64.f, OS_WUNSCALE = 1.f / 64.f;
#ifdef MV_OS_TIMING
__device__ long long* os_dbg = nullptr;   // [wave][8]: alpha, bias + aggregation, first fill, tiles (compute), tiles (carry + commit), pooled, total
#define OS_TM(slot) do { const long long t_ = clock64(); tacc[slot] += t_ - tlast; tlast = t_; } while (0)
#else
#define OS_TM(slot) do {} while (0)
#endif

template <int CIN, int MT, bool XF16>
__global__ __launch_bounds__(OS_NW * 64) void odconv_stream_kernel(const void* __restrict__ x, const float* __restrict__ wp,
                                                                  const float* __restrict__ bias, const float* __restrict__ alpha_in,
                                                                  const float* __restrict__ pooled_in, const float* __restrict__ att_w,
                                                                  const float* __restrict__ att_b, float* __restrict__ y,
                                                                  float* __restrict__ pooled_out, OdP p, int L) {
  static_assert((XF16 ? CIN * 2 : CIN * 4) == OS_RB, "256-byte input rows");
  constexpr int CPC = CIN / 8, KST = 2 * CIN / 32, NF = MT * KST, KB = 4;
  constexpr int GB = XF16 ? 16 : 32;                     // bytes of one 8-channel group in an LDS row (f16 | f16 hi + lo)
  using MH = Mma<f16>;
  extern __shared__ __align__(16) char lds[];
  char* aimg = lds;                                      // [NF][hi 1 KB | lo 1 KB]
  float* alds = reinterpret_cast<float*>(lds + NF * 2048);
  float* bias_l = alds + OD_MAXK;                        // [MT * 16]
  float* red = bias_l + MT * 16;                         // [OS_NW][MT * 16]
  char* rings = reinterpret_cast<char*>(red + OS_NW * MT * 16);
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, g = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, R0 = blockIdx.y * MT * 16, n_mt = p.M / 16;
  const int ring = (int)(rings - lds) + wid * OS_RING, carry = ring + 2 * OS_SLOT;
#ifdef MV_OS_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
  const long long tfirst = tlast, t_abs0 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- this wave's span of input columns (the sample's last wave also takes the ragged end: nq = Tin + 1 leaves one column over)
  const int wave_id = blockIdx.x * OS_NW + wid;
  const int s0 = wave_id * L;
  const int span = (wave_id == gridDim.x * OS_NW - 1 ? p.nq : min(p.nq, s0 + L)) - s0;
  const int nt = span > 0 ? (span + 15) >> 4 : 0;
  const long xsample = (long)b * p.Tin * OS_RB;
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(x)) + xsample, 0, p.Tin * OS_RB, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry =
      __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(y) + (long)b * p.Tout * p.Cout * 4, 0, p.Tout * p.Cout * 4, 0x00020000);
  // staging: a batch = 16 rows x 256 B = 4 x 16 B per lane.  fp16 rows: lane -> 16-byte chunk (lane & 15) of rows (lane >> 4) + 4 k;
  // fp32 rows: lane -> 8-channel group (lane & 7) of rows (lane >> 3) + 8 k, two 16-byte pieces each
  auto row_of = [&](int k) { return XF16 ? (lane >> 4) + 4 * k : (lane >> 3) + 8 * (k >> 1); };
  auto off_of = [&](int k) { return XF16 ? (lane & 15) * 16 : (lane & 7) * 32 + (k & 1) * 16; };
  u32x4 ra[4], rb[4];
  auto issue = [&](u32x4 (&r)[4], int jb) {              // batch jb = input rows s0 + 16 jb ..
#pragma unroll
    for (int k = 0; k < 4; ++k)
      r[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, (s0 + 16 * jb + row_of(k)) * OS_RB + off_of(k), 0, 0);
  };
  auto put = [&](int dst_row0, int r, int k, const u32x4& v0, const u32x4& v1) {
    // XF16: one 16-byte chunk as it is; fp32: an 8-channel group (v0, v1) -> f16 hi | lo
    if constexpr (XF16) {
      *reinterpret_cast<u32x4*>(lds + dst_row0 + r * OS_RS + off_of(k)) = v0;
    } else {
      const f32x4 a = __builtin_bit_cast(f32x4, v0), c = __builtin_bit_cast(f32x4, v1);
      uint32_t h[4], l[4];
      Mma<f32w16>::split2(a[0], a[1], h[0], l[0]); Mma<f32w16>::split2(a[2], a[3], h[1], l[1]);
      Mma<f32w16>::split2(c[0], c[1], h[2], l[2]); Mma<f32w16>::split2(c[2], c[3], h[3], l[3]);
      char* d = lds + dst_row0 + r * OS_RS + (lane & 7) * 32;
      *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
      *reinterpret_cast<u32x4*>(d + 16) = u32x4{l[0], l[1], l[2], l[3]};
    }
  };
  auto commit = [&](const u32x4 (&r)[4], int slot) {
    if constexpr (XF16) {
#pragma unroll
      for (int k = 0; k < 4; ++k) put(ring + slot * OS_SLOT, row_of(k), k, r[k], r[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k += 2) put(ring + slot * OS_SLOT, row_of(k), k, r[k], r[k + 1]);
    }
  };
  // ---- prologue: attention chain (three dependent round trips on a cold chip: ~11.6 k cycles), then this wave's share of the bank
  //      fragments (256 KB per workgroup through the CU's memory path: ~10 k cycles) with the first rows of the span right behind them
  u32x4 c0 = {0u, 0u, 0u, 0u}, c1 = {0u, 0u, 0u, 0u};
  const bool carry_mine = XF16 ? lane < 16 : lane < 8;
  auto first_rows = [&]() {
    if (nt > 0) {
      if (carry_mine) {
        c0 = __builtin_amdgcn_raw_buffer_load_b128(rx, (s0 - 1) * OS_RB + (XF16 ? lane * 16 : lane * 32), 0, 0);
        if constexpr (!XF16) c1 = __builtin_amdgcn_raw_buffer_load_b128(rx, (s0 - 1) * OS_RB + lane * 32 + 16, 0, 0);
      }
      issue(ra, 0);
      if (nt > 1) issue(rb, 1);
    }
  };
  constexpr int FPW = NF / OS_NW;                        // fragments per wave
  static_assert(NF % OS_NW == 0, "fragments split evenly over the waves");
  WLoad<float>::R wr[FPW][KB];
  auto frag_loads = [&]() {
    const long bank_stride = (long)n_mt * KST * 512;      // floats
    const float* wlane = wp + (long)lane * 8;
#pragma unroll
    for (int u = 0; u < FPW; ++u) {
      const int f = wid * FPW + u;
      const int mtl = f / KST, ks = f - mtl * KST;
      const int mt = (blockIdx.y * MT + mtl) < n_mt ? (blockIdx.y * MT + mtl) : (n_mt - 1);
      const float* wbase = wlane + ((long)mt * KST + ks) * 512;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) wr[u][kb] = WLoad<float>::load(reinterpret_cast<const char*>(wbase + (kb < p.K ? kb : 0) * bank_stride));
    }
  };
  // vector-memory results return in issue order per wave: the waves that load the attention chain's partial sums and weights (0-3)
  // must not queue 32 fragment loads in front of them; the other half of the workgroup has nothing else to wait for
  // (measured: requesting any of this under the attention chain only queues the chain's own small loads behind it in the CU's memory
  //  path - 172 KB requested by the idle half of the workgroup took the chain from 11.6 k to 33 k cycles; the order that works is chain
  //  first, then the fragments, with the rows of the first tiles right behind them)
  constexpr bool early = false;

  // (The one-wave chain of od_alpha_wave is a single round trip requested by wave 0 before anything else it asks for: the other seven
  //  waves' fragment and row requests go out at once without getting in front of it.)
  const bool wavealpha = !alpha_in && p.K <= 4 && CIN <= 64 * OD_LU;
  if (wavealpha) {
    if (__builtin_amdgcn_readfirstlane(wid) == 0) od_alpha_wave<float>(alds, b, p, pooled_in, att_w, att_b);
    frag_loads(); first_rows();
    lds_barrier();
  } else {
    if (alpha_in) {
      if (tid < p.K) alds[tid] = alpha_in[(long)b * p.K + tid];
    } else {
      od_alpha_from_partials<float>(alds, reinterpret_cast<float*>(rings), 1, b, p, pooled_in, att_w, att_b);   // rings: nothing staged yet
    }
    __syncthreads();
  }
  OS_TM(0);
  if (!early && !wavealpha) { frag_loads(); first_rows(); }
  float al[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) al[kb] = kb < p.K ? alds[kb] : 0.f;
  if (tid < MT * 16) {
    const int row = R0 + tid;
    float a = 0.f;
    if (bias && row < p.M)
      for (int kb = 0; kb < p.K; ++kb) a += al[kb] * bias[(long)kb * p.Cout + row % p.Cout];
    bias_l[tid] = a;
  }
  // this sample's kernel, once per workgroup: fragment f = (local M-tile, k-step) -> hi / lo f16 images
#pragma unroll
  for (int u = 0; u < FPW; ++u) {
    float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) WLoad<float>::fma8(wr[u][kb], al[kb] * OS_WSCALE, f);
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) Mma<f32w16>::split2(f[2 * e], f[2 * e + 1], h[e], l[e]);
    char* dst = aimg + (size_t)(wid * FPW + u) * 2048 + lane * 16;
    *reinterpret_cast<u32x4*>(dst) = u32x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(dst + 1024) = u32x4{l[0], l[1], l[2], l[3]};
  }
  if (nt > 0) {                                          // (the alpha scratch in `rings` is dead: od_alpha_from_partials ends behind a barrier)
    if (carry_mine) put(carry, 0, 0, c0, c1);
    commit(ra, 0);
  }
  __syncthreads();
  OS_TM(1);

  // B-operand bases of this lane: k-step ks -> chunk 4 ks + g -> (tap, 8-channel group); tap 0 reads row col of the tile's slot, tap 1
  // row col - 1 (column 0: the carry row = the last row of the previous batch)
  int baddr[KST][2];
#pragma unroll
  for (int ks = 0; ks < KST; ++ks) {
    const int chunk = 4 * ks + g, tap = chunk / CPC, c8 = chunk % CPC;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      baddr[ks][u] = (tap == 1 && col == 0) ? carry + c8 * GB : ring + u * OS_SLOT + (col - tap) * OS_RS + c8 * GB;
  }
  // output rows of this lane: GEMM row R0 + 16 m + 4 g (+ i) = (phase, channel); u = q * stride + phase - pad
  int vo[MT], uofs[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row = R0 + 16 * m + 4 * g, rr = row / p.Cout, o = row - rr * p.Cout;
    uofs[m] = rr - p.pad;
    vo[m] = p.out_pair ? uofs[m] * p.Cout * 4 + 32 * (o >> 3) + 8 * ((o >> 2) & 1) : (uofs[m] * p.Cout + o) * 4;
  }
  const float slope = p.act == ACT_NONE ? 1.f : p.slope;
  float psum[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) psum[m][i] = 0.f;

  auto tile = [&](auto uc, int j) {
    constexpr int U = decltype(uc)::value;
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // flat (k-step, M-tile) sequence, operands one step ahead of their MFMAs
    MH::V bh[2], bl[2], ah[2], alo[2];
    auto ldb = [&](int ks, int set) {
      bh[set] = MH::load_b(lds + baddr[ks][U]);
      if constexpr (!XF16) bl[set] = MH::load_b(lds + baddr[ks][U] + 16);
    };
    auto lda = [&](int sq, int set) {
      const int ks = sq / MT, m = sq % MT;
      const char* ap = aimg + (size_t)(m * KST + ks) * 2048 + lane * 16;
      ah[set] = MH::load_b(ap);
      alo[set] = MH::load_b(ap + 1024);
    };
    ldb(0, 0);
    lda(0, 0);
#pragma unroll
    for (int sq = 0; sq < NF; ++sq) {
      const int ks = sq / MT, m = sq % MT;
      if (sq + 1 < NF) {
        lda(sq + 1, (sq + 1) & 1);
        if ((sq + 1) % MT == 0) ldb(ks + 1, (ks + 1) & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      acc[m] = MH::mma(alo[sq & 1], bh[ks & 1], acc[m]);
      if constexpr (!XF16) acc[m] = MH::mma(ah[sq & 1], bl[ks & 1], acc[m]);
      acc[m] = MH::mma(ah[sq & 1], bh[ks & 1], acc[m]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: un-scale, mixed bias, activation, pooling sums for the next layer's attention, 16-byte stores (rows outside the sample
    // are dropped by the range check)
    const int q = s0 + 16 * j + col;
    const int qo = q * p.stride * p.Cout * 4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + 16 * m + 4 * g);
      const int uu = q * p.stride + uofs[m];
      const bool ok = q < p.nq && uu >= 0 && uu < p.Tout && (R0 + 16 * m) < p.M;
      f32x4 ov;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = acc[m][i] * OS_WUNSCALE + bv[i];
        ov[i] = v >= 0.f ? v : v * slope;
        psum[m][i] += ok ? ov[i] : 0.f;
      }
      if ((R0 + 16 * m) < p.M) {
        if (p.out_pair) {                                     // [8 x f16 hi | 8 x f16 lo] per 8-channel group: this lane's 4 channels = 8 + 8 bytes
          uint32_t h0, l0, h1, l1;
          Mma<f32w16>::split2(ov[0], ov[1], h0, l0);
          Mma<f32w16>::split2(ov[2], ov[3], h1, l1);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{h0, h1}, ry, qo + vo[m], 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{l0, l1}, ry, qo + vo[m] + 16, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), ry, qo + vo[m], 0, 0);
        }
      }
    }
  };

  if (nt > 0) {
    auto step = [&](auto uc, int j, u32x4 (&rnext)[4], u32x4 (&rfree)[4]) {
      // on entry: batch j is in slot U, batch j + 1 travels in rnext; rfree takes batch j + 2
      constexpr int U = decltype(uc)::value;
      if (j + 2 < nt) issue(rfree, j + 2);
      tile(uc, j);
      OS_TM(3);
      if (j + 1 < nt) {
        // the next tile's carry = the last row of this batch; then batch j + 1 lands in the other slot
        if (lane < 16) {
          const u32x4 t_ = *reinterpret_cast<const u32x4*>(lds + ring + U * OS_SLOT + 15 * OS_RS + lane * 16);
          *reinterpret_cast<u32x4*>(lds + carry + lane * 16) = t_;
        }
        commit(rnext, U ^ 1);
      }
      OS_TM(4);
    };
    for (int j = 0; j < nt; j += 2) {
      step(std::integral_constant<int, 0>{}, j, rb, ra);
      if (j + 1 < nt) step(std::integral_constant<int, 1>{}, j + 1, ra, rb);
    }
  }
#ifdef MV_OS_TIMING
  if (lane == 0 && os_dbg) {
    const long w = (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * OS_NW + wid;
    if (w < 65536) {
      long long* d = os_dbg + w * 8;
      for (int i = 0; i < 5; ++i) d[i] = tacc[i];
      d[5] = clock64() - tfirst; d[6] = t_abs0; d[7] = __builtin_amdgcn_s_memrealtime();
    }
  }
#endif

  if (pooled_out) {                                      // one partial per (row, workgroup): slot blockIdx.x of this sample
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = psum[m][i];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
        if (col == 0) red[wid * MT * 16 + 16 * m + 4 * g + i] = v;
      }
    __syncthreads();
    if (tid < MT * 16 && R0 + tid < p.M) {
      float a = 0.f;
      for (int w = 0; w < OS_NW; ++w) a += red[w * MT * 16 + tid];
      pooled_out[((long)b * gridDim.x + blockIdx.x) * p.M + R0 + tid] = a;
    }
  }
}

template <int CIN, int MT, bool XF16>
static int od_stream_launch(const void* x, const void* wp, const void* bias, const float* alpha, const float* pooled_in, const void* att_w,
                            const void* att_b, void* y, float* pooled_out, OdP p, hipStream_t stream, int* slots_out) {
  constexpr int KST = 2 * CIN / 32;
  if ((long)p.Tin * OS_RB >= (1l << 31) || (long)p.Tout * p.Cout * 4 >= (1l << 31) || p.Cout % 16 || p.K > 4) return MV_ERR_UNSUPPORTED;
  const size_t lds = (size_t)MT * KST * 2048 + sizeof(float) * (OD_MAXK + MT * 16 + OS_NW * MT * 16) + (size_t)OS_NW * OS_RING;
  if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
  const int gy = cdiv(p.M / 16, MT);
  // nq = Tin + 1 columns: the odd one is taken by the sample's last wave (a 257th tile would cost every wave a fifth tile at C2)
  const int tiles = p.nq >= 32 ? (p.nq - 1) / 16 : cdiv(p.nq, 16);
  const int wgs = 256 / (p.B * gy) > 1 ? 256 / (p.B * gy) : 1;     // workgroups per (sample, row block) that fill the chip once
  int tpw = cdiv(tiles, wgs * OS_NW);
  if (tpw < 1) tpw = 1;
  const int gx = cdiv(tiles, tpw * OS_NW);
  if (slots_out) { *slots_out = gx; return MV_OK; }
  if (gy > 65535 || p.B > 65535) return MV_ERR_UNSUPPORTED;
  auto kern = odconv_stream_kernel<CIN, MT, XF16>;
  static bool set = false;
  if (!set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
#ifdef MV_OS_TIMING
  static long long* dbg = nullptr;
  static int calls = 0;
  if (!dbg) { (void)hipMalloc(&dbg, 65536 * 8 * 8); (void)hipMemcpyToSymbol(HIP_SYMBOL(os_dbg), &dbg, sizeof(dbg)); }
  (void)hipMemsetAsync(dbg, 0xff, 65536 * 8 * 8, stream);
#endif
  hipLaunchKernelGGL(kern, dim3(gx, gy, p.B), dim3(OS_NW * 64), lds, stream, x, (const float*)wp, (const float*)bias, alpha, pooled_in,
                     (const float*)att_w, (const float*)att_b, (float*)y, pooled_out, p, tpw * 16);
#ifdef MV_OS_TIMING
  {
    const char* e = getenv("MV_MRF_TIMING_CALL");
    if (++calls == (e ? atoi(e) : 30)) {
      (void)hipStreamSynchronize(stream);
      static long long hbuf[65536 * 8];
      (void)hipMemcpy(hbuf, dbg, sizeof(hbuf), hipMemcpyDeviceToHost);
      double av[6] = {0}; int cnt = 0; long long tmin = -1, tmax = -1;
      for (int w = 0; w < 65536; ++w) {
        const long long* d = hbuf + (size_t)w * 8;
        if (d[5] < 0) continue;
        for (int k = 0; k < 6; ++k) av[k] += (double)d[k];
        ++cnt;
        if (tmin < 0 || d[6] < tmin) tmin = d[6];
        if (d[7] > tmax) tmax = d[7];
      }
      if (cnt) fprintf(stderr, "[od stream timing] Cin %d MT %d grid %d x %d x %d tpw %d waves %d: span %.2f us; per wave cycles: alpha %.0f bias+mix %.0f first fill %.0f tiles %.0f carry+commit %.0f total %.0f\n",
                       CIN, MT, gx, gy, p.B, tpw, cnt, (tmax - tmin) / 100.0, av[0] / cnt, av[1] / cnt, av[2] / cnt, av[3] / cnt, av[4] / cnt, av[5] / cnt);
    }
  }
#endif
  return MV_OK;
}

// Picks the kernel variant for a layer and launches it - or, with slots_out, only reports how many partial-sum slots per sample
// that launch writes to pooled_out (the variant sets the grid, so the caller sizes pooled_out from the same decision).
static int od_dispatch(const void* x, const void* packed, const void* bias, const float* alpha, const float* pooled_in,
                       const void* att_w, const void* att_b, const void* film_proj, void* y, float* pooled_out, OdP p,
                       int dtype, hipStream_t st_, int* slots_out) {
  const int K = p.K, B = p.B, Cin = p.Cin, transposed = p.transposed, act = p.act;
  const int ntiles = cdiv(p.nq, 16);
  const long wbytes = (long)K * p.M * p.ksteps * 32 * (dtype == MV_F32 ? 4 : 2);
  int rc = MV_ERR_DTYPE;
  // with the unconditional (clamped) prefetch the prefetching instantiation wins for the short-K upsamplers as well
  // (ups3 48 -> 44 us); MV_OD_PF=0 selects the non-prefetching one for comparison
  static int force_pf = -1;
  if (force_pf < 0) { const char* e = getenv("MV_OD_PF"); force_pf = e ? atoi(e) : 1; }
#define OD_GO(S_, MW_, NB_) do { \
    if (K <= 4 && p.ksteps <= 8 && !force_pf) rc = od_launch<T, S_, MW_, NB_, false, 4>(x, packed, bias, alpha, pooled_in, att_w, att_b, film_proj, y, pooled_out, p, st_, slots_out); \
    else if (K <= 4) rc = od_launch<T, S_, MW_, NB_, true, 4>(x, packed, bias, alpha, pooled_in, att_w, att_b, film_proj, y, pooled_out, p, st_, slots_out); \
    else rc = od_launch<T, S_, MW_, NB_, true, 8>(x, packed, bias, alpha, pooled_in, att_w, att_b, film_proj, y, pooled_out, p, st_, slots_out); } while (0)
  if (p.in_f16 && (dtype != MV_F32 || ntiles <= 3 || wbytes > (1 << 20))) return MV_ERR_UNSUPPORTED;   // (only the multi-tile kernel below)
  if (p.out_pair && (dtype != MV_F32 || ntiles <= 3 || wbytes > (1 << 20) || p.Cout != 64 || p.in_f16)) return MV_ERR_UNSUPPORTED;   // (only the 64-channel streaming kernel)
  MV_DISPATCH(dtype, {
    if (ntiles <= 3) {                       // short sequences, big kernels (input_proj, first upsampler)
      rc = MV_ERR_UNSUPPORTED;
      if (!film_proj && wbytes > (4 << 20) && B >= 2)   // K-loop, weights-stationary over S samples
      {   // S = 2 measured best; S = 1 if the tiles do not fit.  MV_KLOOP_MW selects the M-tiles per wave (A/B measurements)
        // measured at ups0 (C2): 16-bit 41.5 us with one M-tile per wave (two workgroups per CU) vs 43.8 with two; fp32 (split
        // operands, 3 MFMAs per product) 97 us with two M-tiles per wave vs 133 with one.  The launch is bound by the weight
        // stream out of L2 (16 sample groups x 16.8 MB = 268 MB), not by the matrix pipe (12 us) - see DESIGN.md
        static int kmw = -1;
        if (kmw < 0) { const char* e = getenv("MV_KLOOP_MW"); kmw = e ? atoi(e) : 0; }
        const int mw_sel = kmw ? kmw : (dtype == MV_F32 ? 2 : 1);
        if (mw_sel == 2) {   // (four samples per workgroup halve the L2 stream but leave one wave per SIMD: 58.6 us)
          rc = od_kloop_launch<T, 2, 3, 2>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
          if (rc == MV_ERR_UNSUPPORTED) rc = od_kloop_launch<T, 1, 3, 2>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
        } else {
          rc = od_kloop_launch<T, 2, 3, 1>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
          if (rc == MV_ERR_UNSUPPORTED) rc = od_kloop_launch<T, 1, 3, 1>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
        }
      }
      // two samples per workgroup share the staged weights, but a grid that leaves CUs idle loses more: input_proj at B = 32 is
      // 1 x 8 x 16 = 128 workgroups with S = 2 - one sample per workgroup fills the chip (17.7 -> ~14 us; MV_OD_S1=0: never)
      static int s1 = -1;
      if (s1 < 0) { const char* e = getenv("MV_OD_S1"); s1 = e ? atoi(e) : 1; }
      if (rc == MV_ERR_UNSUPPORTED && s1 && (long)cdiv(B, 2) * cdiv(p.M, 64) * cdiv(ntiles, 3) < 256) OD_GO(1, 1, 3);
      if (rc == MV_ERR_UNSUPPORTED) OD_GO(2, 1, 3);
      if (rc == MV_ERR_UNSUPPORTED) OD_GO(1, 1, 3);
    } else if (wbytes > (1 << 20)) {         // big kernels, medium sequences: 144-column blocks amortise the aggregation
      rc = MV_ERR_UNSUPPORTED;
      {
        static int mt1 = -1;
        if (mt1 < 0) { const char* e = getenv("MV_OD_MT1"); mt1 = e ? atoi(e) : 1; }
        // 256 input channels, ks = 2*stride (ups1): 64-row workgroups keep their 16 mixed fragments per wave resident and walk
        // the sample's 96-column tiles - the bank fragments are fetched from L2 and mixed once per (sample, row block) instead
        // of once per tile (37 -> 28.5 us; 144-column tiles without the x look-ahead measured 34 us)
        static int smp = -1;
        if (smp < 0) { const char* e = getenv("MV_OD_SAMPLE"); smp = e ? atoi(e) : 1; }
        // the whole sample resident in LDS, the mixed fragments streamed (ups1 at B = 32: see odconv_sample_kernel)
        if (smp && transposed && p.ntaps == 2 && K <= 4 && !film_proj && act <= ACT_LRELU && Cin == 256 && B * (p.M / 128) >= 128) {
          if constexpr (sizeof(T) == 2) rc = od_sample_launch<T, 256, 17>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
        }
        if (rc == MV_ERR_UNSUPPORTED && mt1 && transposed && p.ntaps == 2 && K <= 4 && !film_proj && act <= ACT_LRELU && Cin == 256) {
          // (fp32: 16 resident fragment PAIRS per wave need 256 VGPRs = one wave per SIMD: 98 us vs 77 on the one-tile kernel - not used)
          if constexpr (sizeof(T) == 2) rc = od_mt_launch<T, 1, 6, 256>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
        }
      }
      if (rc == MV_ERR_UNSUPPORTED) OD_GO(1, 2, 9);      // (fp32: 64- / 96-column tiles measured 96 / 94 us vs 76)
      if (rc == MV_ERR_UNSUPPORTED) OD_GO(1, 1, 3);
    } else {                                 // small kernels, long sequences: HBM-streaming regime
      rc = MV_ERR_UNSUPPORTED;
      {
        static int mt_on = -1;
        if (mt_on < 0) { const char* e = getenv("MV_OD_MT"); mt_on = e ? atoi(e) : 1; }
        if (mt_on && transposed && p.ntaps == 2 && K <= 4 && !film_proj && act <= ACT_LRELU && (Cin == 64 || Cin == 128)) {
          // 128-column tiles for 64 input channels; 64-column tiles for 128 (8 resident A fragments per M-tile: the wider tile
          // would need > 256 VGPRs, i.e. one wave per SIMD - measured 42 vs 24.5 us)
          if constexpr (sizeof(T) == 2) {
            if (Cin == 64) rc = od_mt_launch<T, 2, 8, 64>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
            else rc = od_mt_launch<T, 2, 4, 128>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
          } else {   // fp32 (operands are hi/lo register pairs): 64-column tiles; at 128 channels ONE M-tile per wave (two cost a wave per SIMD: 60 us)
            static int os_on = -1;
            if (os_on < 0) { const char* e = getenv("MV_OD_STREAM"); os_on = e ? atoi(e) : 1; }
            if (os_on && act <= ACT_LRELU) {    // streaming form: 256-byte input rows (64 fp32 / 128 fp16 channels)
              if (p.in_f16 && Cin == 128) rc = od_stream_launch<128, 4, true>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
              else if (!p.in_f16 && Cin == 64) rc = od_stream_launch<64, 8, false>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
            }
            if (rc != MV_ERR_UNSUPPORTED) { /* launched (or sized) */ }
            else if (p.out_pair) return rc;
            else if (p.in_f16) {
              if (Cin == 64) rc = od_mt_launch<T, 2, 4, 64, f16>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
              else rc = od_mt_launch<T, 1, 4, 128, f16>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
            } else if (Cin == 64) rc = od_mt_launch<T, 2, 4, 64>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);
            else rc = od_mt_launch<T, 1, 4, 128>(x, packed, bias, alpha, pooled_in, att_w, att_b, y, pooled_out, p, st_, slots_out);   // ups2 49 -> 38 us
          }
        }
      }
      if (p.in_f16 || p.out_pair) return rc;                 // only the multi-tile / streaming kernels widen their input / write pair rows
      if (rc == MV_ERR_UNSUPPORTED) OD_GO(1, 2, 8);
      if (rc == MV_ERR_UNSUPPORTED) OD_GO(1, 1, 3);
    }
  });
#undef OD_GO
  return rc;
}

static int odconv_cl_fwd_impl(const void* x, const void* packed, const void* bias, const float* alpha,
                              const float* pooled_in, int pooled_in_count, const void* att_w, const void* att_b,
                              const void* film_proj, int film_F, void* y, float* pooled_out, int B, int Cin, int Tin,
                              int Cout, int Tout, int ks, int stride, int pad, int dil, int transposed, int K, int act,
                              float slope, int dtype, void* stream, int in_f16, int out_pair = 0) {
  MV_CHECK_ARG(x && packed && y && (alpha || (pooled_in && att_w && pooled_in_count > 0 && pooled_in_count % Cin == 0)));
  MV_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)packed & 15) == 0);
  MV_CHECK_ARG(((uintptr_t)pooled_out & 15) == 0);
  OdP p;
  if (!od_make(&p, B, Cin, Tin, Cout, Tout, ks, stride, pad, dil, transposed, K, act, slope, film_proj ? film_F : 0))
    return MV_ERR_UNSUPPORTED;
  if (pooled_in) p.pool_n = pooled_in_count;
  p.in_f16 = in_f16;
  p.out_pair = out_pair;
  const int rc = od_dispatch(x, packed, bias, alpha, pooled_in, att_w, att_b, film_proj, y, pooled_out, p, dtype,
                             (hipStream_t)stream, nullptr);
  if (rc != MV_OK) return rc;
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_odconv_cl_fwd(const void* x, const void* packed, const void* bias, const float* alpha,
                                const float* pooled_in, int pooled_in_count, const void* att_w, const void* att_b,
                                const void* film_proj, int film_F, void* y, float* pooled_out, int B, int Cin, int Tin,
                                int Cout, int Tout, int ks, int stride, int pad, int dil, int transposed, int K, int act,
                                float slope, int dtype, void* stream) {
  return odconv_cl_fwd_impl(x, packed, bias, alpha, pooled_in, pooled_in_count, att_w, att_b, film_proj, film_F, y, pooled_out, B, Cin,
                            Tin, Cout, Tout, ks, stride, pad, dil, transposed, K, act, slope, dtype, stream, 0);
}

extern "C" int mv_odconv_cl_fwd_in16(const void* x_f16, const void* packed, const void* bias, const float* alpha,
                                     const float* pooled_in, int pooled_in_count, const void* att_w, const void* att_b,
                                     void* y, float* pooled_out, int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride,
                                     int pad, int dil, int transposed, int K, int act, float slope, void* stream) {
  return odconv_cl_fwd_impl(x_f16, packed, bias, alpha, pooled_in, pooled_in_count, att_w, att_b, nullptr, 0, y, pooled_out, B, Cin,
                            Tin, Cout, Tout, ks, stride, pad, dil, transposed, K, act, slope, MV_F32, stream, 1);
}

extern "C" int mv_odconv_cl_fwd_pair(const void* x, const void* packed, const void* bias, const float* alpha,
                                     const float* pooled_in, int pooled_in_count, const void* att_w, const void* att_b,
                                     void* y_pair, float* pooled_out, int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride,
                                     int pad, int dil, int transposed, int K, int act, float slope, void* stream) {
  return odconv_cl_fwd_impl(x, packed, bias, alpha, pooled_in, pooled_in_count, att_w, att_b, nullptr, 0, y_pair, pooled_out, B, Cin,
                            Tin, Cout, Tout, ks, stride, pad, dil, transposed, K, act, slope, MV_F32, stream, 0, 1);
}

extern "C" size_t mv_odconv_cl_pool_floats(int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad, int dil,
                                         int transposed, int K, int act, int has_film, int dtype) {
  return mv_odconv_cl_pool_floats_in(B, Cin, Tin, Cout, Tout, ks, stride, pad, dil, transposed, K, act, has_film, dtype, 0);
}

extern "C" size_t mv_odconv_cl_pool_floats_in(int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad, int dil,
                                            int transposed, int K, int act, int has_film, int dtype, int in_f16) {
  OdP p;
  if (!od_make(&p, B, Cin, Tin, Cout, Tout, ks, stride, pad, dil, transposed, K, act, 0.1f, has_film ? 8 : 0)) return 0;
  p.in_f16 = (in_f16 && dtype == MV_F32) ? 1 : 0;
  int slots = 0;
  static const int dummy = 0;   // non-null stand-ins: the dry run launches nothing and touches no memory
  const void* d = &dummy;
  const int rc = od_dispatch(d, d, nullptr, nullptr, nullptr, nullptr, nullptr, has_film ? d : nullptr, nullptr, nullptr, p,
                             dtype, nullptr, &slots);
  if (rc != MV_OK || slots <= 0) return 0;
  return (size_t)slots * p.M;
}
