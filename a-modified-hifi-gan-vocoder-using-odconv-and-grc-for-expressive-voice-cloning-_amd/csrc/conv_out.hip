// Output projection of the generator, channels-last input: Conv1d(C -> 1, ks, pad ks/2) + tanh
//   reference arithmetic: SURVEY.md Appendix A item 4 / forward (nn.Conv1d(64,1,11,padding=5) then torch.tanh).
// x [B][T][C] (NTC)  ->  y [B][1][T] (which is also NCT).  HBM-bound: reads the stream once, writes 1/C of it.
// One thread per output step; the workgroup's (256 + ks - 1) input rows are staged in LDS with a padded row
// stride; weights are read through wave-uniform (scalar) loads.
#include "mfma.h"

namespace mv {

template <typename T, int C>
__global__ __launch_bounds__(256) void conv_out_tanh_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                            float bias, T* __restrict__ y, int Tn, int ks, int pad,
                                                            int act) {
  using M = Mma<T>;
  constexpr int ES = M::ES;
  constexpr int RS = C * ES + 16;
  extern __shared__ __align__(16) char lds[];
  const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
  const int rows = 256 + ks - 1;
  constexpr int CPR = C * ES / 16;
  const T* xb = x + (size_t)b * Tn * C;
  stage_batched<9, 256>(tid, rows * CPR, lds, [&](int i, const void*& src, int& dst) {
    const int r = i / CPR, ch = i % CPR;
    const int t = t0 - pad + r;
    if (t >= 0 && t < Tn) src = reinterpret_cast<const char*>(xb + (size_t)t * C) + ch * 16;
    dst = r * RS + ch * 16;
  });
  __syncthreads();
  float acc0 = bias, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  for (int j = 0; j < ks; ++j) {
    const char* row = lds + (size_t)(tid + j) * RS;
    const float* wj = w + j * C;
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      float xv[4];
      M::load4(row + c * ES, xv);
      acc0 += wj[c] * xv[0];
      acc1 += wj[c + 1] * xv[1];
      acc2 += wj[c + 2] * xv[2];
      acc3 += wj[c + 3] * xv[3];
    }
  }
  const int t = t0 + tid;
  if (t < Tn) st<T>(y + (size_t)b * Tn + t, apply_act((acc0 + acc1) + (acc2 + acc3), act, 0.f));
}

// The same output convolution fed by the LAST block of an MRF chain (mrf_fused.hip): the input row is formed on the way into LDS as
// a[b][c] * f + b[b][c] + x  (that block's GroupNorm(8,64) + residual add, deferred), so the block's output never makes a round
// trip through HBM.  ab = fp32 [B][2][C].
template <typename T, int C>
__global__ __launch_bounds__(256) void conv_out_affine_kernel(const T* __restrict__ f, const T* __restrict__ x,
                                                              const float* __restrict__ ab, const float* __restrict__ w, float bias,
                                                              T* __restrict__ y, int Tn, int ks, int pad, int act) {
  using M = Mma<T>;
  constexpr int ES = M::ES;
  constexpr int RS = C * ES + 16;
  constexpr int CPR = C * ES / 16, EPC = 16 / ES;
  extern __shared__ __align__(16) char lds[];
  const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
  const int rows = 256 + ks - 1;
  const T* xb = x + (size_t)b * Tn * C;
  const T* fb = f + (size_t)b * Tn * C;
  // a thread keeps its 16-byte column (256 % CPR == 0): its EPC channels' affine stays in registers
  static_assert(256 % CPR == 0, "fixed column per thread");
  const int ch = tid % CPR;
  float ra[EPC], rb[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) { ra[j] = ab[(size_t)b * 2 * C + ch * EPC + j]; rb[j] = ab[(size_t)b * 2 * C + C + ch * EPC + j]; }
  constexpr int UB = 9;      // (256 + ks - 1) rows / 16 row groups = 17 rows per thread: two batches of loads in flight
  for (int r0 = tid / CPR; r0 < rows; r0 += (256 / CPR) * UB) {
    alignas(16) T xv[UB][EPC];
    alignas(16) T fv[UB][EPC];
    bool ok[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int r = r0 + u * (256 / CPR), t = t0 - pad + r;
      ok[u] = r < rows && t >= 0 && t < Tn;
      const size_t off = (size_t)(ok[u] ? t : 0) * C + ch * EPC;
      *reinterpret_cast<uint4*>(xv[u]) = *reinterpret_cast<const uint4*>(xb + off);
      *reinterpret_cast<uint4*>(fv[u]) = *reinterpret_cast<const uint4*>(fb + off);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int r = r0 + u * (256 / CPR);
      if (r < rows) {
        alignas(16) T o[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) st<T>(o + j, ok[u] ? ra[j] * ld<T>(fv[u] + j) + rb[j] + ld<T>(xv[u] + j) : 0.f);
        *reinterpret_cast<uint4*>(lds + (size_t)r * RS + ch * 16) = *reinterpret_cast<const uint4*>(o);
      }
    }
  }
  __syncthreads();
  float acc0 = bias, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  for (int j = 0; j < ks; ++j) {
    const char* row = lds + (size_t)(tid + j) * RS;
    const float* wj = w + j * C;
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      float xq[4];
      M::load4(row + c * ES, xq);
      acc0 += wj[c] * xq[0];
      acc1 += wj[c + 1] * xq[1];
      acc2 += wj[c + 2] * xq[2];
      acc3 += wj[c + 3] * xq[3];
    }
  }
  const int t = t0 + tid;
  if (t < Tn) st<T>(y + (size_t)b * Tn + t, apply_act((acc0 + acc1) + (acc2 + acc3), act, 0.f));
}

// w_t[j][c] = w[0][c][j] as fp32 (tiny)
template <typename P>
__global__ void conv_out_pack_kernel(const P* __restrict__ w, float* __restrict__ wt, int C, int ks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C * ks) { const int j = i / C, c = i % C; wt[i] = ld<P>(w + c * ks + j); }
}


// 16-bit storage: the same sliding dot product on v_dot2c_f32_{bf16,f16} (two MACs per lane and instruction, fp32 accumulate, no
// unpacking).  The VALU form above spends 4 instructions per 2 MACs (2 converts + 2 FMAs) and is VALU-bound at 16.6 us for the
// 33.5 MB it reads; here a lane reads 16 bytes of its row (4 channel pairs) per LDS access and the weight pairs arrive as
// wave-uniform scalar loads from a pre-packed image in the activation type.
template <typename T> struct Dot2;
template <> struct Dot2<bf16> {
  static __device__ __forceinline__ float f(uint32_t a, uint32_t b, float c) {
    typedef __attribute__((ext_vector_type(2))) __bf16 v2;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
  }
};
template <> struct Dot2<f16> {
  static __device__ __forceinline__ float f(uint32_t a, uint32_t b, float c) {
    typedef __attribute__((ext_vector_type(2))) _Float16 v2;
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
  }
};

template <typename T, int C>
__global__ __launch_bounds__(256) void conv_out_dot2_kernel(const T* __restrict__ x, const uint32_t* __restrict__ wpairs, float bias,
                                                            T* __restrict__ y, int Tn, int ks, int pad, int act) {
  constexpr int ES = 2, RS = C * ES + 16, CPR = C * ES / 16;
  extern __shared__ __align__(16) char lds[];
  const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
  const int rows = 256 + ks - 1;
  const T* xb = x + (size_t)b * Tn * C;
  stage_batched<9, 256>(tid, rows * CPR, lds, [&](int i, const void*& src, int& dst) {
    const int r = i / CPR, ch = i % CPR;
    const int t = t0 - pad + r;
    if (t >= 0 && t < Tn) src = reinterpret_cast<const char*>(xb + (size_t)t * C) + ch * 16;
    dst = r * RS + ch * 16;
  });
  __syncthreads();
  float acc0 = bias, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  for (int j = 0; j < ks; ++j) {
    const char* row = lds + (size_t)(tid + j) * RS;
    const uint32_t* wj = wpairs + j * (C / 2);                // wave-uniform
#pragma unroll
    for (int c8 = 0; c8 < CPR; ++c8) {
      const u32x4 xv = *reinterpret_cast<const u32x4*>(row + c8 * 16);
      acc0 = Dot2<T>::f(xv[0], wj[c8 * 4 + 0], acc0);
      acc1 = Dot2<T>::f(xv[1], wj[c8 * 4 + 1], acc1);
      acc2 = Dot2<T>::f(xv[2], wj[c8 * 4 + 2], acc2);
      acc3 = Dot2<T>::f(xv[3], wj[c8 * 4 + 3], acc3);
    }
  }
  const int t = t0 + tid;
  if (t < Tn) st<T>(y + (size_t)b * Tn + t, apply_act((acc0 + acc1) + (acc2 + acc3), act, 0.f));
}

// 16-bit storage with the taps on the MFMA rows (as conv_out_affine_mfma_kernel below does for fp32): z[tap][pos] = sum_c w[tap][c] x[pos][c]
// (A = the [ks][C] weight image of this type, 16-byte fragments straight from it; B = the staged rows as they are; 2 k-steps per 16
// positions, every row read once), then y[t] = bias + sum_tap z[tap][t + tap] from a z tile that takes the rows' place in LDS.  The dot2
// form reads every row ks times and fetches its weight pairs through scalar loads inside the tap loop.
template <typename T, int C>
__global__ __launch_bounds__(256) void conv_out_mfma16_kernel(const T* __restrict__ x, const T* __restrict__ wimg, float bias,
                                                              T* __restrict__ y, int Tn, int ks, int pad, int act) {
  static_assert(C == 64 && sizeof(T) == 2, "two 32-channel k-steps of a 16-bit type");
  using M = Mma<T>;
  using V = typename M::V;
  constexpr int TS = 256, ES = 2, RS = C * ES + 16, CPR = C * ES / 16;
  extern __shared__ __align__(16) char lds[];
  const int b = blockIdx.y, t0 = blockIdx.x * TS, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int rows = TS + ks - 1, nblk = (rows + 15) / 16;
  const T* xb = x + (size_t)b * Tn * C;
  V a[2];
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2) {
    alignas(16) T wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) st<T>(wv + e, 0.f);
    if (col < ks) *reinterpret_cast<u32x4*>(wv) = *reinterpret_cast<const u32x4*>(wimg + col * C + k2 * 32 + 8 * g);
    a[k2] = M::load_b(wv);
  }
  stage_batched<9, 256>(tid, nblk * 16 * CPR, lds, [&](int i, const void*& src, int& dst) {   // rows past `rows`: zeros (src stays null)
    const int r = i / CPR, ch = i % CPR;
    const int t = t0 - pad + r;
    if (r < rows && t >= 0 && t < Tn) src = reinterpret_cast<const char*>(xb + (size_t)t * C) + ch * 16;
    dst = r * RS + ch * 16;
  });
  __syncthreads();
  constexpr int MAXB = 5;                   // position blocks per wave: 17 blocks over 4 waves
  f32x4 z[MAXB];
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pb = wid + 4 * i;
    if (pb < nblk) {
      const char* brow = lds + (size_t)(pb * 16 + col) * RS + g * 16;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) z[i] = M::mma(a[k2], M::load_b(brow + k2 * 64), z[i]);
    }
  }
  __syncthreads();                          // every wave is done with the rows: the z tile takes their place
  constexpr int ZS = TS + 32;
  float* zt = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int pb = wid + 4 * i;
    if (pb < nblk) {
#pragma unroll
      for (int r = 0; r < 4; ++r) zt[(4 * g + r) * ZS + pb * 16 + col] = z[i][r];
    }
  }
  __syncthreads();
  float acc = bias;
  for (int j = 0; j < ks; ++j) acc += zt[j * ZS + tid + j];
  const int t = t0 + tid;
  if (t < Tn) st<T>(y + (size_t)b * Tn + t, apply_act(acc, act, 0.f));
}

// all three images of w [1][C][ks]: fp32 [ks][C] | bf16 [ks][C] | f16 [ks][C]
template <typename P>
__global__ void conv_out_pack_all_kernel(const P* __restrict__ w, char* __restrict__ out, int C, int ks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C * ks) {
    const int j = i / C, c = i % C;
    const float v = ld<P>(w + c * ks + j);
    reinterpret_cast<float*>(out)[i] = v;
    st<bf16>(reinterpret_cast<bf16*>(out + (size_t)C * ks * 4) + i, v);
    st<f16>(reinterpret_cast<f16*>(out + (size_t)C * ks * 6) + i, v);
  }
}

// fp32 storage: the same fused ending with the channel contraction on the matrix pipe.  The VALU form above spends 23 k of its 44 k ticks
// per workgroup in the dot products: a thread needs all ks x 64 weights, they arrive through ~14 serialised scalar loads per tap, and the
// rows are read 11 times (176 ds_read_b128 per thread).  Here the taps are the MFMA rows, as in the discriminator head:
//   z[tap][pos] = sum_c w[tap][c] * row[pos][c]   (A = the weights as a 16 x 64 operand, zero beyond ks; B = the staged rows, PRE-SPLIT
//   into hi + lo bf16 planes when they are committed; 2 k-steps x 3 products per 16 positions, every row read ONCE),
//   y[t] = bias + sum_tap z[tap][t + tap]          (the z tile replaces the rows in LDS after a barrier; 11 adds per output).
// XFMT: 0 = f and x fp32 rows; 1 = x as pair rows (8 groups of [8 x f16 hi | 8 x f16 lo], mrf_stream.hip), f fp32; 2 = f and x as hl8
// rows (mfma.h: 192 bytes, f16 hi plane + e4m3 lo plane); 3 = f hl8, x fp32 (a one-block chain); 4 = x hl8, f fp32
template <int C, int XFMT>
__global__ __launch_bounds__(256) void conv_out_affine_mfma_kernel(const float* __restrict__ f, const float* __restrict__ x,
                                                                   const float* __restrict__ ab, const float* __restrict__ w, float bias,
                                                                   float* __restrict__ y, int Tn, int ks, int pad, int act,
                                                                   const float* __restrict__ part8, const float* __restrict__ tab8, int nwg,
                                                                   float eps) {
  static_assert(C == 64, "two 32-channel k-steps");
  using M = Mma<float>;
  using V = M::V;
  constexpr int TS = 256;
  constexpr int PL = C * 2;                 // bytes of one bf16 plane of a row
  constexpr int RS = 2 * PL + 16;           // hi plane, lo plane, pad
  constexpr int CPR = C / 4;                // 4-channel columns per row (16 bytes of fp32 in, 8 + 8 bytes of bf16 out)
  constexpr int UB = 9;
  extern __shared__ __align__(16) char lds[];
  const int b = blockIdx.y, t0 = blockIdx.x * TS, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int rows = TS + ks - 1, nblk = (rows + 15) / 16;
  const float* xb = x + (size_t)b * Tn * C;
  const float* fb = f + (size_t)b * Tn * C;
  const int ch = tid % CPR;
  float ra[4], rb[4];
  if (part8) {   // the last MRF block's GroupNorm(8,64) affine straight from its per-workgroup partial sums (the sums of mrf_affine_kernel,
                 // in its order): a = rstd * gamma, b = beta - mean * a; this thread's 4 channels share one group
    const int q = (ch * 4) >> 3;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < nwg; ++i) {
      s1 += part8[((size_t)(b * nwg + i) * 8 + q) * 2];
      s2 += part8[((size_t)(b * nwg + i) * 8 + q) * 2 + 1];
    }
    const float n = 8.f * (float)Tn, mu = s1 / n;
    const float rs = rsqrtf(fmaxf(s2 / n - mu * mu, 0.f) + eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) { ra[j] = rs * tab8[320 + ch * 4 + j]; rb[j] = tab8[384 + ch * 4 + j] - mu * ra[j]; }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) { ra[j] = ab[(size_t)b * 2 * C + ch * 4 + j]; rb[j] = ab[(size_t)b * 2 * C + C + ch * 4 + j]; }
  }
  // A operand: row = tap (lane & 15), k = 32 * kstep + 8 * (lane >> 4) + e  ->  w[tap][k]  (w is [ks][C] fp32)
  V a[2];
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2) {
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) wv[e] = col < ks ? w[col * C + k2 * 32 + 8 * g + e] : 0.f;
    a[k2] = M::split(wv);
  }
  for (int r0 = tid / CPR; r0 < nblk * 16; r0 += (256 / CPR) * UB) {
    f32x4 xv[UB], fv[UB];
    bool ok[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int r = r0 + u * (256 / CPR), t = t0 - pad + r;
      ok[u] = r < rows && t >= 0 && t < Tn;
      const size_t off = (size_t)(ok[u] ? t : 0) * C + ch * 4;
      typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
      // this thread's 4 channels of an hl8 row: 8 bytes of the hi plane, 4 of the lo plane (the row index is per SAMPLE-relative t)
      auto hl4 = [&](const float* base) {
        const char* p = reinterpret_cast<const char*>(base) + ((size_t)b * Tn + (ok[u] ? t : 0)) * SR_HLB;
        const f16x4_t h = *reinterpret_cast<const f16x4_t*>(p + ch * 8);
        const int l = *reinterpret_cast<const int*>(p + 128 + ch * 4);
        const f32x2_t l0 = __builtin_amdgcn_cvt_pk_f32_fp8(l, false), l1 = __builtin_amdgcn_cvt_pk_f32_fp8(l, true);
        return f32x4{(float)h[0] + l0[0] * SR_LOI, (float)h[1] + l0[1] * SR_LOI, (float)h[2] + l1[0] * SR_LOI, (float)h[3] + l1[1] * SR_LOI};
      };
      if constexpr (XFMT == 2 || XFMT == 4) {
        xv[u] = hl4(x);
      } else if constexpr (XFMT == 1) {
        const char* px = reinterpret_cast<const char*>(xb) + (size_t)(ok[u] ? t : 0) * (C * 4) + (ch >> 1) * 32 + (ch & 1) * 8;
        const f16x4_t h = *reinterpret_cast<const f16x4_t*>(px), l = *reinterpret_cast<const f16x4_t*>(px + 16);
        xv[u] = f32x4{(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
      } else {
        xv[u] = *reinterpret_cast<const f32x4*>(xb + off);
      }
      if constexpr (XFMT == 2 || XFMT == 3) fv[u] = hl4(f);
      else fv[u] = *reinterpret_cast<const f32x4*>(fb + off);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int r = r0 + u * (256 / CPR);
      if (r < nblk * 16) {                  // rows beyond `rows` (the last block's tail) are zero operands
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = ok[u] ? ra[j] * fv[u][j] + rb[j] + xv[u][j] : 0.f;
        u32x2 hi, lo;
        M::split4(o, hi, lo);
        *reinterpret_cast<u32x2*>(lds + (size_t)r * RS + ch * 8) = hi;
        *reinterpret_cast<u32x2*>(lds + (size_t)r * RS + PL + ch * 8) = lo;
      }
    }
  }
  __syncthreads();
  constexpr int MAXB = 5;                   // position blocks per wave: (256 + 15 + 15) / 16 = 17 blocks over 4 waves
  f32x4 z[MAXB];
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pb = wid + 4 * i;
    if (pb < nblk) {
      const char* brow = lds + (size_t)(pb * 16 + col) * RS + g * 16;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) z[i] = M::mma(a[k2], M::load_bp(brow + k2 * 64, PL), z[i]);
    }
  }
  __syncthreads();                          // every wave is done with the rows: the z tile takes their place
  constexpr int ZS = TS + 32;               // floats per tap row of the z tile (>= 16 * nblk)
  float* zt = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int i = 0; i < MAXB; ++i) {
    const int pb = wid + 4 * i;
    if (pb < nblk) {
#pragma unroll
      for (int r = 0; r < 4; ++r) zt[(4 * g + r) * ZS + pb * 16 + col] = z[i][r];
    }
  }
  __syncthreads();
  float acc = bias;
  for (int j = 0; j < ks; ++j) acc += zt[j * ZS + tid + j];
  const int t = t0 + tid;
  if (t < Tn) y[(size_t)b * Tn + t] = apply_act(acc, act, 0.f);
}

}  // namespace mv

using namespace mv;

// internal (not part of the C ABI): used by the MRF chain's fused ending, mrf_fused.hip
bool mvi_conv_out_affine_takes_partials(int dtype, int ks) {
  static int use_mfma = -1;
  if (use_mfma < 0) { const char* e = getenv("MV_CONV_OUT_MFMA"); use_mfma = e ? atoi(e) : 1; }
  return dtype == MV_F32 && use_mfma && ks <= 16;
}

int mvi_conv_out_affine(const void* f, const void* x, const float* ab, const float* wt, float bias, void* y, int B, int T_, int C, int ks,
                        int pad, int act, int dtype, hipStream_t stream, const float* part8, const float* tab8, int nwg, float eps,
                        int x_pair) {
  if (C != 64 || 2 * pad != ks - 1) return MV_ERR_UNSUPPORTED;
  if (x_pair && !(dtype == MV_F32 && mvi_conv_out_affine_takes_partials(dtype, ks))) return MV_ERR_UNSUPPORTED;
  if (part8 && !mvi_conv_out_affine_takes_partials(dtype, ks)) return MV_ERR_UNSUPPORTED;
  dim3 grid(cdiv(T_, 256), B);
  static int use_mfma = -1;
  if (use_mfma < 0) { const char* e = getenv("MV_CONV_OUT_MFMA"); use_mfma = e ? atoi(e) : 1; }
  if (dtype == MV_F32 && use_mfma && ks <= 16) {
    const size_t rows16 = (size_t)((256 + ks - 1 + 15) / 16) * 16;
    const size_t lds = rows16 * (2 * 64 * 2 + 16);       // >= the z tile (16 x 288 floats)
    if (x_pair < 0 || x_pair > 4) return MV_ERR_ARG;
    auto kern = x_pair == 4 ? conv_out_affine_mfma_kernel<64, 4> : x_pair == 3 ? conv_out_affine_mfma_kernel<64, 3>
              : x_pair == 2 ? conv_out_affine_mfma_kernel<64, 2> : x_pair == 1 ? conv_out_affine_mfma_kernel<64, 1>
              : conv_out_affine_mfma_kernel<64, 0>;
    static size_t lds_set_m[5] = {0, 0, 0, 0, 0};
    if (lds > lds_set_m[x_pair]) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      lds_set_m[x_pair] = lds;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const float*)f, (const float*)x, ab, wt, bias, (float*)y, T_, ks, pad, act,
                       part8, tab8, nwg, eps);
    return MV_OK;
  }
  MV_DISPATCH(dtype, {
    const size_t lds = (size_t)(256 + ks - 1) * (64 * Mma<T>::ES + 16);
    if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
    auto kern = conv_out_affine_kernel<T, 64>;
    static size_t lds_set = 0;
    if (lds > lds_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); lds_set = lds; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const T*)f, (const T*)x, ab, wt, bias, (T*)y, T_, ks, pad, act);
  });
  return MV_OK;
}

extern "C" int mv_conv_out_pack(const void* w, int param_dtype, float* wt, int C, int ks, void* stream) {
  MV_CHECK_ARG(w && wt && C > 0 && ks > 0);
  const dim3 g(cdiv(C * ks, 256)), b(256);
  switch (param_dtype) {
    case MV_F32: hipLaunchKernelGGL(conv_out_pack_kernel<float>, g, b, 0, (hipStream_t)stream, (const float*)w, wt, C, ks); break;
    case MV_BF16: hipLaunchKernelGGL(conv_out_pack_kernel<bf16>, g, b, 0, (hipStream_t)stream, (const bf16*)w, wt, C, ks); break;
    case MV_F16: hipLaunchKernelGGL(conv_out_pack_kernel<f16>, g, b, 0, (hipStream_t)stream, (const f16*)w, wt, C, ks); break;
    default: return MV_ERR_DTYPE;
  }
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_conv_out_act_cl(const void* x, const float* wt, float bias, void* y, int B, int T_, int C, int ks,
                                  int pad, int act, int dtype, void* stream) {
  MV_CHECK_ARG(x && wt && y && B > 0 && B <= 65535 && T_ > 0 && ks > 0 && pad >= 0 && ((uintptr_t)x & 15) == 0);
  if (C != 64 || 2 * pad != ks - 1) return MV_ERR_UNSUPPORTED;
  dim3 grid(cdiv(T_, 256), B);
  MV_DISPATCH(dtype, {
    const size_t lds = (size_t)(256 + ks - 1) * (64 * Mma<T>::ES + 16);
    if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
    auto kern = conv_out_tanh_kernel<T, 64>;
    static size_t lds_set = 0;
    if (lds > lds_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); lds_set = lds; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, (const T*)x, wt, bias, (T*)y, T_, ks, pad, act);
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" size_t mv_conv_out_packed_bytes(int C, int ks) { return (size_t)C * ks * 8; }

extern "C" int mv_conv_out_pack_all(const void* w, int param_dtype, void* packed, int C, int ks, void* stream) {
  MV_CHECK_ARG(w && packed && C > 0 && ks > 0 && ((uintptr_t)packed & 15) == 0 && (C * ks) % 8 == 0);
  const dim3 g(cdiv(C * ks, 256)), b(256);
  switch (param_dtype) {
    case MV_F32: hipLaunchKernelGGL(conv_out_pack_all_kernel<float>, g, b, 0, (hipStream_t)stream, (const float*)w, (char*)packed, C, ks); break;
    case MV_BF16: hipLaunchKernelGGL(conv_out_pack_all_kernel<bf16>, g, b, 0, (hipStream_t)stream, (const bf16*)w, (char*)packed, C, ks); break;
    case MV_F16: hipLaunchKernelGGL(conv_out_pack_all_kernel<f16>, g, b, 0, (hipStream_t)stream, (const f16*)w, (char*)packed, C, ks); break;
    default: return MV_ERR_DTYPE;
  }
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_conv_out_act_packed_cl(const void* x, const void* packed, float bias, void* y, int B, int T_, int C, int ks, int pad,
                                         int act, int dtype, void* stream) {
  MV_CHECK_ARG(x && packed && y && B > 0 && B <= 65535 && T_ > 0 && ks > 0 && pad >= 0 && ((uintptr_t)x & 15) == 0);
  if (C != 64 || 2 * pad != ks - 1) return MV_ERR_UNSUPPORTED;
  const char* pk = (const char*)packed;
  if (dtype == MV_F32) return mv_conv_out_act_cl(x, (const float*)pk, bias, y, B, T_, C, ks, pad, act, dtype, stream);
  dim3 grid(cdiv(T_, 256), B);
  static int use_mfma = -1;
  if (use_mfma < 0) { const char* e = getenv("MV_CONV_OUT_MFMA"); use_mfma = e ? atoi(e) : 1; }
  if (use_mfma && ks <= 16 && (dtype == MV_BF16 || dtype == MV_F16)) {
    const size_t ldsm = (size_t)((256 + ks - 1 + 15) / 16) * 16 * (64 * 2 + 16);      // >= the z tile (16 x 288 floats = 18 KB)
    if (dtype == MV_BF16)
      hipLaunchKernelGGL((conv_out_mfma16_kernel<bf16, 64>), grid, dim3(256), ldsm, (hipStream_t)stream, (const bf16*)x,
                         (const bf16*)(pk + (size_t)C * ks * 4), bias, (bf16*)y, T_, ks, pad, act);
    else
      hipLaunchKernelGGL((conv_out_mfma16_kernel<f16, 64>), grid, dim3(256), ldsm, (hipStream_t)stream, (const f16*)x,
                         (const f16*)(pk + (size_t)C * ks * 6), bias, (f16*)y, T_, ks, pad, act);
    MV_LAUNCH_CHECK();
    return MV_OK;
  }
  const size_t lds = (size_t)(256 + ks - 1) * (64 * 2 + 16);
  if (lds > 64 * 1024) return MV_ERR_UNSUPPORTED;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((conv_out_dot2_kernel<bf16, 64>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16*)x,
                       (const uint32_t*)(pk + (size_t)C * ks * 4), bias, (bf16*)y, T_, ks, pad, act);
  else if (dtype == MV_F16)
    hipLaunchKernelGGL((conv_out_dot2_kernel<f16, 64>), grid, dim3(256), lds, (hipStream_t)stream, (const f16*)x,
                       (const uint32_t*)(pk + (size_t)C * ks * 6), bias, (f16*)y, T_, ks, pad, act);
  else return MV_ERR_DTYPE;
  MV_LAUNCH_CHECK();
  return MV_OK;
}
