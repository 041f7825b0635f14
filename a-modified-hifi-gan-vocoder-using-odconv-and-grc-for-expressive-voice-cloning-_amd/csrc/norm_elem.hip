// ODConv attention, GroupNorm (two-phase), FiLM, activations, pooling, MPD fold, layout transposes,
// casts, small dense layers and the GRC+LoRA weight fold.  Generic shapes, NCT layout.
#include "common.h"

namespace mv {

// ---------------------------------------------------------------- ODConv attention (odconv.py:36-40)
// one workgroup per sample: wave w reduces channels w, w+nw, ...; then K dot products + softmax.
template <typename T>
__global__ __launch_bounds__(256) void odconv_attn_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                          const T* __restrict__ bias, float* __restrict__ alpha,
                                                          float* __restrict__ pooled, int C, int Tn, int K) {
  extern __shared__ float sm[];  // [C] means, then [K] logits
  float* mean = sm;
  float* logit = sm + C;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const T* xb = x + (long)b * C * Tn;
  const float inv = 1.f / (float)Tn;
  for (int c = wid; c < C; c += nw) {
    float s = 0.f;
    for (int t = lane; t < Tn; t += 64) s += ld<T>(xb + (long)c * Tn + t);
    s = wave_sum(s);
    if (lane == 0) {
      mean[c] = s * inv;
      if (pooled) pooled[(long)b * C + c] = s * inv;
    }
  }
  __syncthreads();
  for (int k = wid; k < K; k += nw) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ld<T>(w + (long)k * C + c) * mean[c];
    s = wave_sum(s);
    if (lane == 0) logit[k] = s + (bias ? ld<T>(bias + k) : 0.f);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, logit[k]);
    float den = 0.f;
    for (int k = 0; k < K; ++k) den += expf(logit[k] - m);
    for (int k = 0; k < K; ++k) alpha[(long)b * K + k] = expf(logit[k] - m) / den;
  }
}

// long inputs: the per-channel means come from a chip-wide launch (one wave per (sample, channel) row, 16-byte loads),
// then one small workgroup per sample forms the logits and the softmax from them
template <typename T>
__global__ __launch_bounds__(256) void rowmean_kernel(const T* __restrict__ x, float* __restrict__ pooled, long nrows, int Tn) {
  constexpr int EPV = 16 / sizeof(T);
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const T* p = x + row * Tn;
  float s = 0.f;
  if (Tn % EPV == 0 && ((uintptr_t)x & 15) == 0) {
    for (int t = lane * EPV; t < Tn; t += 64 * EPV) {
      alignas(16) T tmp[EPV];
      *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(p + t);
#pragma unroll
      for (int e = 0; e < EPV; ++e) s += ld<T>(tmp + e);
    }
  } else {
    for (int t = lane; t < Tn; t += 64) s += ld<T>(p + t);
  }
  s = wave_sum(s);
  if (lane == 0) pooled[row] = s / (float)Tn;
}
template <typename T>
__global__ __launch_bounds__(64) void odconv_attn_pooled_kernel(const float* __restrict__ pooled, const T* __restrict__ w,
                                                                const T* __restrict__ bias, float* __restrict__ alpha, int C, int K) {
  __shared__ float logit[64];
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ld<T>(w + (long)k * C + c) * pooled[(long)b * C + c];
    s = wave_sum(s);
    if (lane == 0) logit[k] = s + (bias ? ld<T>(bias + k) : 0.f);
  }
  __syncthreads();
  if (lane == 0) {
    float m = -INFINITY, den = 0.f;
    for (int k = 0; k < K; ++k) m = fmaxf(m, logit[k]);
    for (int k = 0; k < K; ++k) den += expf(logit[k] - m);
    for (int k = 0; k < K; ++k) alpha[(long)b * K + k] = expf(logit[k] - m) / den;
  }
}

// ---------------------------------------------------------------- generator prologue (one launch, one workgroup per sample)
// Everything the channels-last generator needs before its first conv, fused: (1) input_proj's ODConv attention
// alpha = softmax(Wa . mean_t mel + ba) (odconv.py:36-40), (2) mel [C][T] -> channels-last [T][C], (3) the FiLM projection
// proj = W . cond + b with cond = cat(spk, emo) truncated / zero-padded to the projection's input width (grc_lora.py:82-105),
// (4) zeroing of this sample's share of the pooled-sum buffers the upsamplers accumulate into.
template <typename T>
__global__ __launch_bounds__(1024) void gen_prologue_kernel(const T* __restrict__ mel, const T* __restrict__ att_w, const T* __restrict__ att_b,
                                                           const T* __restrict__ spk, const T* __restrict__ emo, const T* __restrict__ film_w,
                                                           const T* __restrict__ film_b, float* __restrict__ alpha, T* __restrict__ x_cl,
                                                           T* __restrict__ film_proj, float* __restrict__ zero_buf, long zero_n, int C,
                                                           int Tn, int K, int ds, int de, int cond_dim, int F2) {
  extern __shared__ float sm[];          // [C*Tn] the sample, [C] means, [K] logits, [cond_dim] condition
  float* xs = sm;
  float* mean = xs + C * Tn;
  float* logit = mean + C;
  float* cond = logit + K;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nt = blockDim.x, nw = nt >> 6;
  const T* xb = mel + (long)b * C * Tn;
  for (int i = tid; i < C * Tn; i += nt) xs[i] = ld<T>(xb + i);
  for (int i = tid; i < cond_dim; i += nt) {
    float v = 0.f;
    if (i < ds) v = ld<T>(spk + (long)b * ds + i);
    else if (i < ds + de) v = ld<T>(emo + (long)b * de + (i - ds));
    cond[i] = v;
  }
  // this sample's slice of the buffers that must be zero before the upsamplers run
  {
    const long per = (zero_n + gridDim.x - 1) / gridDim.x, z0 = (long)b * per;
    const long z1 = z0 + per < zero_n ? z0 + per : zero_n;
    for (long i = z0 + tid; i < z1; i += nt) zero_buf[i] = 0.f;
  }
  __syncthreads();
  // channels-last copy
  T* yb = x_cl + (long)b * Tn * C;
  for (int i = tid; i < C * Tn; i += nt) {
    const int t = i / C, c = i - t * C;
    st<T>(yb + i, xs[c * Tn + t]);
  }
  // attention
  const float inv = 1.f / (float)Tn;
  for (int c = wid; c < C; c += nw) {
    float a = 0.f;
    for (int t = lane; t < Tn; t += 64) a += xs[c * Tn + t];
    a = wave_sum(a);
    if (lane == 0) mean[c] = a * inv;
  }
  // FiLM projection: a wave owns 4 output rows at a time (their loads are independent and in flight together); 16-bit
  // weights are read 16 bytes per lane
  if (film_proj) {
    constexpr int EPV = 16 / sizeof(T);
    const bool vec = sizeof(T) == 2 && cond_dim % EPV == 0 && ((uintptr_t)film_w & 15) == 0;
    for (int j0 = wid * 4; j0 < F2; j0 += 4 * nw) {
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        for (int pc = lane; pc < cond_dim / EPV; pc += 64) {
          uint4 wv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            wv[r] = *reinterpret_cast<const uint4*>(film_w + (long)(j0 + r < F2 ? j0 + r : F2 - 1) * cond_dim + pc * EPV);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            alignas(16) T tmp[EPV];
            *reinterpret_cast<uint4*>(tmp) = wv[r];
#pragma unroll
            for (int e = 0; e < EPV; ++e) a[r] += ld<T>(tmp + e) * cond[pc * EPV + e];
          }
        }
      } else {
        for (int i = lane; i < cond_dim; i += 64)
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r] += ld<T>(film_w + (long)(j0 + r < F2 ? j0 + r : F2 - 1) * cond_dim + i) * cond[i];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = wave_sum(a[r]);
        if (lane == 0 && j0 + r < F2) st<T>(film_proj + (long)b * F2 + j0 + r, v + (film_b ? ld<T>(film_b + j0 + r) : 0.f));
      }
    }
  }
  __syncthreads();
  for (int k = wid; k < K; k += nw) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += ld<T>(att_w + (long)k * C + c) * mean[c];
    a = wave_sum(a);
    if (lane == 0) logit[k] = a + (att_b ? ld<T>(att_b + k) : 0.f);
  }
  __syncthreads();
  if (tid == 0) {
    float m = -INFINITY, den = 0.f;
    for (int k = 0; k < K; ++k) m = fmaxf(m, logit[k]);
    for (int k = 0; k < K; ++k) den += expf(logit[k] - m);
    for (int k = 0; k < K; ++k) alpha[(long)b * K + k] = expf(logit[k] - m) / den;
  }
}

// ---------------------------------------------------------------- GroupNorm
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4v;
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ mean,
                                                       float* __restrict__ rstd, int C, int Tn, int G, float eps,
                                                       long x_bs, long x_cs) {
  __shared__ float red[32];
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int cg = C / G;
  const T* xb = x + (long)b * x_bs + (long)(g * cg) * x_cs;
  const long n = (long)cg * Tn;
  // two-pass (mean, then centred second moment) for accuracy: the data is L2-resident on the second pass.
  // Rows are walked channel by channel (no per-element division); 16-bit rows that allow it are read 16 bytes at a time.
  constexpr int EPV = 16 / sizeof(T);
  const bool vec = sizeof(T) == 2 && Tn % EPV == 0 && x_cs % EPV == 0 && x_bs % EPV == 0 && ((uintptr_t)x & 15) == 0;
  float s = 0.f;
  for (int c = 0; c < cg; ++c) {
    const T* row = xb + (long)c * x_cs;
    if (vec) {
      for (int t = threadIdx.x * EPV; t < Tn; t += blockDim.x * EPV) {
        alignas(16) T tmp[EPV];
        *reinterpret_cast<u32x4v*>(tmp) = *reinterpret_cast<const u32x4v*>(row + t);
#pragma unroll
        for (int e = 0; e < EPV; ++e) s += ld<T>(tmp + e);
      }
    } else {
      for (int t = threadIdx.x; t < Tn; t += blockDim.x) s += ld<T>(row + t);
    }
  }
  const float mu = block_sum(s, red) / (float)n;
  float q = 0.f;
  for (int c = 0; c < cg; ++c) {
    const T* row = xb + (long)c * x_cs;
    if (vec) {
      for (int t = threadIdx.x * EPV; t < Tn; t += blockDim.x * EPV) {
        alignas(16) T tmp[EPV];
        *reinterpret_cast<u32x4v*>(tmp) = *reinterpret_cast<const u32x4v*>(row + t);
#pragma unroll
        for (int e = 0; e < EPV; ++e) { const float d = ld<T>(tmp + e) - mu; q += d * d; }
      }
    } else {
      for (int t = threadIdx.x; t < Tn; t += blockDim.x) { const float d = ld<T>(row + t) - mu; q += d * d; }
    }
  }
  const float var = block_sum(q, red) / (float)n;
  if (threadIdx.x == 0) {
    mean[blockIdx.x] = mu;
    rstd[blockIdx.x] = rsqrtf(var + eps);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const T* __restrict__ gw,
                                                       const T* __restrict__ gb, const T* __restrict__ res,
                                                       const uint8_t* __restrict__ mask, float mask_scale,
                                                       T* __restrict__ y, int C, int Tn, int G, int act, float slope,
                                                       long x_bs, long x_cs, long r_bs, long r_cs, long y_bs,
                                                       long y_cs) {
  const int bc = blockIdx.y;  // b*C + c
  const int b = bc / C, c = bc % C;
  const int g = c / (C / G);
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  const float a = rs * (gw ? ld<T>(gw + c) : 1.f);
  const float sh = (gb ? ld<T>(gb + c) : 0.f) - mu * a;
  const T* xr = x + (long)b * x_bs + (long)c * x_cs;
  const T* rr = res ? res + (long)b * r_bs + (long)c * r_cs : nullptr;
  const uint8_t* mr = mask ? mask + (long)bc * Tn : nullptr;
  T* yr = y + (long)b * y_bs + (long)c * y_cs;
  // 16-bit rows that allow it move 16 bytes per thread and access; the activation is chosen once (functor), not per element
  const bool vec = sizeof(T) == 2 && all_mult8(Tn, x_bs, x_cs, r_bs, r_cs, y_bs, y_cs) &&
                   (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0;
  auto body = [&](auto actf) {
    if (vec) {
      for (int t = (blockIdx.x * blockDim.x + threadIdx.x) * 8; t < Tn; t += gridDim.x * blockDim.x * 8) {
        float v[8], r8[8];
        load8f<T>(xr + t, v);
        if (rr) load8f<T>(rr + t, r8);
        uint2 m8 = {0u, 0u};
        if (mr) m8 = *reinterpret_cast<const uint2*>(mr + t);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float w = actf(v[e] * a + sh);
          if (mr) w = (((e < 4 ? m8.x : m8.y) >> (8 * (e & 3))) & 0xffu) ? w * mask_scale : 0.f;
          if (rr) w += r8[e];
          v[e] = w;
        }
        store8f<T>(yr + t, v);
      }
    } else {
      for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Tn; t += gridDim.x * blockDim.x) {
        float v = actf(ld<T>(xr + t) * a + sh);
        if (mr) v = mr[t] ? v * mask_scale : 0.f;
        if (rr) v += ld<T>(rr + t);
        st<T>(yr + t, v);
      }
    }
  };
  if (act <= ACT_LRELU) body(ActLrelu{act == ACT_NONE ? 1.f : slope}); else body(ActAny{act, slope});
}

// ---------------------------------------------------------------- FiLM / scale-shift / activations
template <typename T>
__global__ __launch_bounds__(256) void film_kernel(const T* __restrict__ x, const T* __restrict__ proj,
                                                   T* __restrict__ y, int C, int Tn, int F) {
  const int bc = blockIdx.y, b = bc / C, c = bc % C;
  float gam = 1.f, bet = 0.f;
  if (c < F) {
    gam = ld<T>(proj + (long)b * 2 * F + c);
    bet = ld<T>(proj + (long)b * 2 * F + F + c);
  }
  const T* xr = x + (long)bc * Tn;
  T* yr = y + (long)bc * Tn;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Tn; t += gridDim.x * blockDim.x)
    st<T>(yr + t, ld<T>(xr + t) * gam + bet);
}

template <typename T>
__global__ __launch_bounds__(256) void scale_shift_kernel(const T* __restrict__ x, const T* __restrict__ scale,
                                                          const T* __restrict__ shift, T* __restrict__ y, int Tn) {
  const int bc = blockIdx.y;
  const float s = ld<T>(scale + bc), h = ld<T>(shift + bc);
  const T* xr = x + (long)bc * Tn;
  T* yr = y + (long)bc * Tn;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Tn; t += gridDim.x * blockDim.x)
    st<T>(yr + t, s * ld<T>(xr + t) + h);
}

template <typename T>
__global__ __launch_bounds__(256) void act_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                  T* __restrict__ y, long n, int act, float slope) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = apply_act(ld<T>(x + i), act, slope);
    if (res) v += ld<T>(res + i);
    st<T>(y + i, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int Tn, int To,
                                                      int s) {
  const long row = blockIdx.y;
  const T* xr = x + row * Tn;
  T* yr = y + row * To;
  const float inv = 1.f / (float)s;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < To; t += gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int i = 0; i < s; ++i) a += ld<T>(xr + (long)t * s + i);
    st<T>(yr + t, a * inv);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mpd_fold_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                       int64_t* __restrict__ index, int Tn, int Tp) {
  const long row = blockIdx.y;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Tp; t += gridDim.x * blockDim.x) {
    st<T>(y + row * Tp + t, t < Tn ? ld<T>(x + row * Tn + t) : 0.f);
    if (index && row == 0) index[t] = t < Tn ? (int64_t)t : (int64_t)-1;
  }
}

// ---------------------------------------------------------------- layout transposes (32x32 LDS tiles)
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ x, T* __restrict__ y, int R, int Cn, int ldx, int ldy,
                                                        long x_bs, long y_bs) {
  // x [batch][R][ldx] (columns < Cn used) -> y [batch][Cn][ldy]; y columns R..ldy-1 are zero-filled (channel padding);
  // x_bs / y_bs: elements between batches (y may be a window of a larger, pre-zeroed buffer)
  __shared__ float tile[32][33];
  const long bx = (long)blockIdx.z * x_bs, by = (long)blockIdx.z * y_bs;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    tile[i][tx] = (r0 + i < R && c0 + tx < Cn) ? ld<T>(x + bx + (long)(r0 + i) * ldx + c0 + tx) : 0.f;
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < Cn && r0 + tx < ldy) st<T>(y + by + (long)(c0 + i) * ldy + r0 + tx, tile[tx][i]);
}

// 16-bit variant: 64 x 64 tiles, every global access is 16 bytes (8 elements) along the contiguous axis of its tensor;
// the tile is transposed through LDS with 2-byte writes.  Requires Cn, ldx, ldy multiples of 8 and 16-byte aligned bases.
template <typename T>
__global__ __launch_bounds__(256) void transpose16_kernel(const T* __restrict__ x, T* __restrict__ y, int R, int Cn, int ldx, int ldy,
                                                          long x_bs, long y_bs) {
  __shared__ unsigned short tile[64][64 + 8];          // [col of x][row of x], padded rows
  const long bx = (long)blockIdx.z * x_bs, by = (long)blockIdx.z * y_bs;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * 256;                      // 512 chunks: row = idx / 8, chunk = idx % 8
    const int rr = idx >> 3, ch = idx & 7;
    uint4 v = {0u, 0u, 0u, 0u};
    if (r0 + rr < R && c0 + ch * 8 < Cn) v = *reinterpret_cast<const uint4*>(x + bx + (long)(r0 + rr) * ldx + c0 + ch * 8);
    const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[ch * 8 + j][rr] = e[j];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * 256;                      // output row = column c of x, 8 chunks of 8 x-rows each
    const int cc = idx >> 3, ch = idx & 7;
    if (c0 + cc < Cn && r0 + ch * 8 < ldy)
      *reinterpret_cast<uint4*>(y + by + (long)(c0 + cc) * ldy + r0 + ch * 8) = *reinterpret_cast<const uint4*>(&tile[cc][ch * 8]);
  }
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ x, D* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    st<D>(y + i, ld<S>(x + i));
}

// ---------------------------------------------------------------- small dense layer: one wave per output element
template <typename T>
__global__ __launch_bounds__(256) void linear_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                     const T* __restrict__ b, T* __restrict__ y, int M, int N,
                                                     int Kd) {
  const int lane = threadIdx.x & 63;
  const long o = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (o >= (long)M * N) return;
  const int m = (int)(o / N), n = (int)(o % N);
  float s = 0.f;
  for (int k = lane; k < Kd; k += 64) s += ld<T>(x + (long)m * Kd + k) * ld<T>(w + (long)n * Kd + k);
  s = wave_sum(s);
  if (lane == 0) st<T>(y + o, s + (b ? ld<T>(b + n) : 0.f));
}

// ---------------------------------------------------------------- GRC + LoRA weight fold (grc_lora.py:33-57)
template <typename T, bool LDS>
__global__ __launch_bounds__(256) void grc_fold_kernel(const T* __restrict__ conv_w, const T* __restrict__ conv_b,
                                                       const T* __restrict__ A, const T* __restrict__ Bm,
                                                       const T* __restrict__ scal, const T* __restrict__ proj_w,
                                                       const T* __restrict__ proj_b, T* __restrict__ w_eff,
                                                       T* __restrict__ b_eff, int Cin, int Cout, int ks, int groups,
                                                       int rank) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = Cout * Cin * ks;
  const int cin_g = Cin / groups, cout_g = Cout / groups;
  const float s = ld<T>(scal);
  // the four small operands go to LDS first (all their loads in flight together): read element by element from global inside the
  // Cout x rank loop nest below they were a chain of dependent L2 round trips (34 us for 3840 outputs)
  extern __shared__ float gsm[];
  float* Al = gsm;                                   // [Cin][rank]
  float* Bl = Al + Cin * rank;                       // [rank][Cout]
  float* Pl = Bl + rank * Cout;                      // [Cout][Cout]
  float* Wl = Pl + Cout * Cout;                      // [Cout][cin_g][ks]
  if (LDS) {
    for (int i = threadIdx.x; i < Cin * rank; i += blockDim.x) Al[i] = ld<T>(A + i);
    for (int i = threadIdx.x; i < rank * Cout; i += blockDim.x) Bl[i] = ld<T>(Bm + i);
    for (int i = threadIdx.x; i < Cout * Cout; i += blockDim.x) Pl[i] = ld<T>(proj_w + i);
    for (int i = threadIdx.x; i < Cout * cin_g * ks; i += blockDim.x) Wl[i] = ld<T>(conv_w + i);
    __syncthreads();
  }
  if (!LDS && idx < total) {                         // operands too large for LDS: straight from global
    const int j = idx % ks, c = (idx / ks) % Cin, o = idx / (ks * Cin);
    float acc = 0.f;
    for (int op = 0; op < Cout; ++op) {
      float comb = 0.f;
      const int g = op / cout_g;
      if (c / cin_g == g) comb = ld<T>(conv_w + ((long)op * cin_g + (c - g * cin_g)) * ks + j);
      if (j == ks / 2) {
        float l = 0.f;
        for (int r = 0; r < rank; ++r) l += ld<T>(A + (long)c * rank + r) * ld<T>(Bm + (long)r * Cout + op);
        comb += s * l;
      }
      acc += ld<T>(proj_w + (long)o * Cout + op) * comb;
    }
    st<T>(w_eff + idx, acc);
  }
  if (LDS && idx < total) {
    const int j = idx % ks, c = (idx / ks) % Cin, o = idx / (ks * Cin);
    float acc = 0.f;
    for (int op = 0; op < Cout; ++op) {
      float comb = 0.f;
      const int g = op / cout_g;
      if (c / cin_g == g) comb = Wl[(op * cin_g + (c - g * cin_g)) * ks + j];
      if (j == ks / 2) {
        float l = 0.f;
        for (int r = 0; r < rank; ++r) l += Al[c * rank + r] * Bl[r * Cout + op];
        comb += s * l;
      }
      acc += Pl[o * Cout + op] * comb;
    }
    st<T>(w_eff + idx, acc);
  }
  if (idx < Cout) {
    float acc = ld<T>(proj_b + idx);
    for (int op = 0; op < Cout; ++op) acc += ld<T>(proj_w + (long)idx * Cout + op) * ld<T>(conv_b + op);
    st<T>(b_eff + idx, acc);
  }
}

}  // namespace mv

using namespace mv;

static inline int grid_for(long n, int block = 256, int cap = 2048) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

extern "C" int mv_abi_version(void) { return 1; }
extern "C" const char* mv_build_target(void) { return "gfx950"; }

extern "C" int mv_odconv_attn_fwd(const void* x, const void* w, const void* bias, float* alpha, float* pooled, int B,
                                  int C, int T_, int K, int dtype, void* stream) {
  MV_CHECK_ARG(x && w && alpha && B > 0 && C > 0 && T_ > 0 && K > 0 && K <= 64);
  const size_t lds = sizeof(float) * (C + K);
  MV_CHECK_ARG(lds <= 64 * 1024);
  if (pooled && (long)C * T_ >= 16384) {      // long rows: spread the means over the chip (B workgroups alone leave it idle)
    const long nrows = (long)B * C;
    MV_DISPATCH(dtype, {
      hipLaunchKernelGGL(rowmean_kernel<T>, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)x, pooled, nrows, T_);
      hipLaunchKernelGGL(odconv_attn_pooled_kernel<T>, dim3(B), dim3(64), 0, (hipStream_t)stream, pooled, (const T*)w, (const T*)bias, alpha, C, K);
    });
    MV_LAUNCH_CHECK();
    return MV_OK;
  }
  MV_DISPATCH(dtype, hipLaunchKernelGGL(odconv_attn_kernel<T>, dim3(B), dim3(256), lds, (hipStream_t)stream,
                                        (const T*)x, (const T*)w, (const T*)bias, alpha, pooled, C, T_, K));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_gen_prologue(const void* mel, const void* att_w, const void* att_b, const void* spk, const void* emo,
                               const void* film_w, const void* film_b, float* alpha, void* x_cl, void* film_proj, float* zero_buf,
                               long zero_n, int B, int C, int T_, int K, int ds, int de, int cond_dim, int F2, int dtype, void* stream) {
  MV_CHECK_ARG(mel && att_w && alpha && x_cl && B > 0 && C > 0 && T_ > 0 && K > 0 && K <= 64 && ds >= 0 && de >= 0 && zero_n >= 0);
  MV_CHECK_ARG((ds == 0 || spk) && (de == 0 || emo) && (!film_proj || (film_w && cond_dim > 0 && F2 > 0)) && (zero_n == 0 || zero_buf));
  const size_t lds = sizeof(float) * ((size_t)C * T_ + C + K + (film_proj ? cond_dim : 0));
  if (lds > 64 * 1024) return MV_ERR_UNSUPPORTED;   // long inputs: the caller issues the separate launches
  MV_DISPATCH(dtype, hipLaunchKernelGGL(gen_prologue_kernel<T>, dim3(B), dim3(1024), lds, (hipStream_t)stream, (const T*)mel,
                                        (const T*)att_w, (const T*)att_b, (const T*)spk, (const T*)emo, (const T*)film_w,
                                        (const T*)film_b, alpha, (T*)x_cl, (T*)film_proj, zero_buf, zero_n, C, T_, K, ds, de,
                                        film_proj ? cond_dim : 0, F2));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_groupnorm_stats(const void* x, float* mean, float* rstd, int B, int C, int T_, int G, float eps,
                                  long x_bs, long x_cs, int dtype, void* stream) {
  MV_CHECK_ARG(x && mean && rstd && B > 0 && C > 0 && T_ > 0 && G > 0 && C % G == 0);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(B * G), dim3(256), 0, (hipStream_t)stream,
                                        (const T*)x, mean, rstd, C, T_, G, eps, x_bs, x_cs));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_groupnorm_apply(const void* x, const float* mean, const float* rstd, const void* gw, const void* gb,
                                  const void* res, const uint8_t* mask, float mask_scale, void* y, int B, int C,
                                  int T_, int G, int act, float slope, long x_bs, long x_cs, long r_bs, long r_cs,
                                  long y_bs, long y_cs, int dtype, void* stream) {
  MV_CHECK_ARG(x && mean && rstd && y && B > 0 && C > 0 && T_ > 0 && G > 0 && C % G == 0 && (long)B * C <= 65535);
  // 16-bit rows that qualify for the 16-byte path (same test as in the kernel) need 8x fewer threads along T
  const bool vec = dtype != MV_F32 && all_mult8(T_, x_bs, x_cs, r_bs, r_cs, y_bs, y_cs) &&
                   (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0;
  dim3 grid(grid_for(vec ? cdiv(T_, 8) : T_, 256, 64), B * C);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(gn_apply_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, mean,
                                        rstd, (const T*)gw, (const T*)gb, (const T*)res, mask, mask_scale, (T*)y, C,
                                        T_, G, act, slope, x_bs, x_cs, r_bs, r_cs, y_bs, y_cs));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_film_fwd(const void* x, const void* proj, void* y, int B, int C, int T_, int F, int dtype,
                           void* stream) {
  MV_CHECK_ARG(x && proj && y && B > 0 && C > 0 && T_ > 0 && F > 0 && (long)B * C <= 65535);
  dim3 grid(grid_for(T_, 256, 64), B * C);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(film_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                        (const T*)proj, (T*)y, C, T_, F));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_scale_shift_fwd(const void* x, const void* scale, const void* shift, void* y, int B, int C, int T_,
                                  int dtype, void* stream) {
  MV_CHECK_ARG(x && scale && shift && y && B > 0 && C > 0 && T_ > 0 && (long)B * C <= 65535);
  dim3 grid(grid_for(T_, 256, 64), B * C);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(scale_shift_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                        (const T*)scale, (const T*)shift, (T*)y, T_));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_act_fwd(const void* x, const void* res, void* y, long n, int act, float slope, int dtype,
                          void* stream) {
  MV_CHECK_ARG(x && y && n > 0);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(act_kernel<T>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                                        (const T*)x, (const T*)res, (T*)y, n, act, slope));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_avgpool1d_fwd(const void* x, void* y, long rows, int T_, int s, int dtype, void* stream) {
  MV_CHECK_ARG(x && y && rows > 0 && rows <= 65535 && s > 0 && T_ >= s);
  const int To = T_ / s;
  dim3 grid(grid_for(To, 256, 64), (int)rows);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(avgpool_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y,
                                        T_, To, s));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_mpd_fold(const void* x, void* y, int64_t* index, long rows, int T_, int P, int dtype, void* stream) {
  MV_CHECK_ARG(x && y && rows > 0 && rows <= 65535 && T_ > 0 && P > 0);
  const int Tp = (T_ % P == 0) ? T_ : T_ + (P - T_ % P);
  dim3 grid(grid_for(Tp, 256, 64), (int)rows);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(mpd_fold_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                        (T*)y, index, T_, Tp));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

static int transpose_launch(const void* x, void* y, int batch, int R, int Cn, int ldx, int ldy, int dtype, void* stream,
                            long x_bs = -1, long y_bs = -1) {
  MV_CHECK_ARG(x && y && batch > 0 && batch <= 65535 && R > 0 && Cn > 0 && ldx >= Cn && ldy >= R && cdiv(ldy, 32) <= 65535);
  if (x_bs < 0) x_bs = (long)R * ldx;
  if (y_bs < 0) y_bs = (long)Cn * ldy;
  if (dtype != MV_F32 && all_mult8(Cn, ldx, ldy, x_bs, y_bs) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    dim3 grid(cdiv(Cn, 64), cdiv(ldy, 64), batch);
    if (dtype == MV_BF16) hipLaunchKernelGGL(transpose16_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, R, Cn, ldx, ldy, x_bs, y_bs);
    else hipLaunchKernelGGL(transpose16_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, R, Cn, ldx, ldy, x_bs, y_bs);
    MV_LAUNCH_CHECK();
    return MV_OK;
  }
  dim3 grid(cdiv(Cn, 32), cdiv(ldy, 32), batch);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                        (T*)y, R, Cn, ldx, ldy, x_bs, y_bs));
  MV_LAUNCH_CHECK();
  return MV_OK;
}
extern "C" int mv_nct_to_ntc(const void* x, void* y, int B, int C, int T_, int dtype, void* stream) {
  return transpose_launch(x, y, B, C, T_, T_, C, dtype, stream);
}
extern "C" int mv_ntc_to_nct(const void* x, void* y, int B, int C, int T_, int dtype, void* stream) {
  return transpose_launch(x, y, B, T_, C, C, T_, dtype, stream);
}
extern "C" int mv_nct_to_ntc_pad(const void* x, void* y, int B, int C, int T_, int Cpad, int dtype, void* stream) {
  MV_CHECK_ARG(Cpad >= C);
  return transpose_launch(x, y, B, C, T_, T_, Cpad, dtype, stream);
}
// adjoint operator of ODConvTranspose1d with ks = 2*stride as a 2-tap stride-1 conv over rows of `stride` output steps:
// out[k][c][r*Cout + o][qt] = w[k][c][o][qt*stride + r]   (w = kernels [K][Cin][Cout][ks])
template <typename P>
__global__ __launch_bounds__(256) void odconvT_adjoint_kernel(const P* __restrict__ w, P* __restrict__ out, long n, int Cout, int ks, int stride) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int qt = (int)(i % 2);
    const long t1 = i / 2;
    const int ro = (int)(t1 % ((long)stride * Cout));
    const long kc = t1 / ((long)stride * Cout);
    const int r = ro / Cout, o = ro - r * Cout;
    out[i] = w[(kc * Cout + o) * ks + qt * stride + r];
  }
}
extern "C" int mv_odconvT_adjoint_weights(const void* kernels, void* out, int K, int Cin, int Cout, int ks, int stride, int dtype,
                                          void* stream) {
  MV_CHECK_ARG(kernels && out && K > 0 && Cin > 0 && Cout > 0 && stride > 0 && ks == 2 * stride);
  const long n = (long)K * Cin * Cout * ks;
  MV_DISPATCH(dtype, hipLaunchKernelGGL(odconvT_adjoint_kernel<T>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const T*)kernels,
                                        (T*)out, n, Cout, ks, stride));
  MV_LAUNCH_CHECK();
  return MV_OK;
}
extern "C" int mv_nct_to_ntc_window(const void* x, void* y, int B, int C, int T_, long y_batch_stride, int dtype, void* stream) {
  MV_CHECK_ARG(y_batch_stride >= (long)T_ * C);
  return transpose_launch(x, y, B, C, T_, T_, C, dtype, stream, -1, y_batch_stride);
}
extern "C" int mv_ntc_to_nct_crop(const void* x, void* y, int B, int C, int T_, int Cpad, int dtype, void* stream) {
  MV_CHECK_ARG(Cpad >= C);
  return transpose_launch(x, y, B, T_, C, Cpad, T_, dtype, stream);
}

template <typename S>
static int cast_from(const void* x, void* y, int dst, long n, void* stream) {
  const dim3 g(grid_for(n)), b(256);
  switch (dst) {
    case MV_F32: hipLaunchKernelGGL((cast_kernel<S, float>), g, b, 0, (hipStream_t)stream, (const S*)x, (float*)y, n); break;
    case MV_BF16: hipLaunchKernelGGL((cast_kernel<S, bf16>), g, b, 0, (hipStream_t)stream, (const S*)x, (bf16*)y, n); break;
    case MV_F16: hipLaunchKernelGGL((cast_kernel<S, f16>), g, b, 0, (hipStream_t)stream, (const S*)x, (f16*)y, n); break;
    default: return MV_ERR_DTYPE;
  }
  MV_LAUNCH_CHECK();
  return MV_OK;
}
extern "C" int mv_cast(const void* x, int src_dtype, void* y, int dst_dtype, long n, void* stream) {
  MV_CHECK_ARG(x && y && n > 0);
  switch (src_dtype) {
    case MV_F32: return cast_from<float>(x, y, dst_dtype, n, stream);
    case MV_BF16: return cast_from<bf16>(x, y, dst_dtype, n, stream);
    case MV_F16: return cast_from<f16>(x, y, dst_dtype, n, stream);
    default: return MV_ERR_DTYPE;
  }
}

extern "C" int mv_linear_fwd(const void* x, const void* w, const void* b, void* y, int M, int N, int Kd, int dtype,
                             void* stream) {
  MV_CHECK_ARG(x && w && y && M > 0 && N > 0 && Kd > 0);
  const long outs = (long)M * N;
  MV_DISPATCH(dtype, hipLaunchKernelGGL(linear_kernel<T>, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0,
                                        (hipStream_t)stream, (const T*)x, (const T*)w, (const T*)b, (T*)y, M, N, Kd));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_grc_fold_weights(const void* conv_w, const void* conv_b, const void* lora_A, const void* lora_B,
                                   const void* lora_scaling, const void* proj_w, const void* proj_b, void* w_eff,
                                   void* b_eff, int Cin, int Cout, int ks, int groups, int rank, int dtype,
                                   void* stream) {
  MV_CHECK_ARG(conv_w && conv_b && lora_A && lora_B && lora_scaling && proj_w && proj_b && w_eff && b_eff);
  MV_CHECK_ARG(Cin > 0 && Cout > 0 && ks > 0 && (ks & 1) && groups > 0 && Cin % groups == 0 && Cout % groups == 0 && rank > 0);
  const int total = Cout * Cin * ks;
  const size_t flds = sizeof(float) * ((size_t)Cin * rank + (size_t)rank * Cout + (size_t)Cout * Cout + (size_t)Cout * (Cin / groups) * ks);
  if (flds <= 48 * 1024) {
    MV_DISPATCH(dtype, hipLaunchKernelGGL((grc_fold_kernel<T, true>), dim3(cdiv(total > Cout ? total : Cout, 256)), dim3(256), flds,
                                          (hipStream_t)stream, (const T*)conv_w, (const T*)conv_b, (const T*)lora_A,
                                          (const T*)lora_B, (const T*)lora_scaling, (const T*)proj_w, (const T*)proj_b,
                                          (T*)w_eff, (T*)b_eff, Cin, Cout, ks, groups, rank));
  } else {
    MV_DISPATCH(dtype, hipLaunchKernelGGL((grc_fold_kernel<T, false>), dim3(cdiv(total > Cout ? total : Cout, 256)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)conv_w, (const T*)conv_b, (const T*)lora_A,
                                          (const T*)lora_B, (const T*)lora_scaling, (const T*)proj_w, (const T*)proj_b,
                                          (T*)w_eff, (T*)b_eff, Cin, Cout, ks, groups, rank));
  }
  MV_LAUNCH_CHECK();
  return MV_OK;
}
