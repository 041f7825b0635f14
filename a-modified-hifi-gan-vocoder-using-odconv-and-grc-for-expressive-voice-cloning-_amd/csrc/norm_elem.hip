// ODConv attention, GroupNorm (two-phase), FiLM, activations, pooling, MPD fold, layout transposes,
// casts, small dense layers and the GRC+LoRA weight fold.  Generic shapes, NCT layout.
#include "common.h"
#include "mfma.h"

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

// ---------------------------------------------------------------- generator prologue (one launch)
// Everything the channels-last generator needs before its first conv, fused: (1) input_proj's ODConv attention
// alpha = softmax(Wa . mean_t mel + ba) (odconv.py:36-40), (2) mel [C][T] -> channels-last [T][C], (3) the FiLM projection
// proj = W . cond + b with cond = cat(spk, emo) truncated / zero-padded to the projection's input width (grc_lora.py:82-105),
// (4) zeroing of the buffers the caller names (none since the pooling sums became per-workgroup partials).
// Two workgroup roles in one grid.  Workgroups [0, B): one sample each - (1), (2), (4).  Workgroups [B, B + n_film): one
// (row group, sample group) of the FiLM projection each - PRO_SG samples x one row per wave, so the projection weights cross
// L2 -> CU once per sample GROUP.  (With the projection inside the per-sample workgroups every one of them pulled the whole
// weight matrix - 229 KB in fp32 - through its CU's 64 B/clk vector-memory path: 5 of the kernel's 13 us; the attention part was
// another 7 us of wave-tree reductions, 64-way LDS bank conflicts of the [C][T] tile read column-wise, and barriers.)
constexpr int PRO_SG = 8;      // samples per FiLM workgroup
constexpr int PRO_FI = 8;      // projection inputs per lane held in registers (cond_dim <= 64 * PRO_FI on the fast path)
#ifdef MV_PRO_TIMING
__device__ long long* pro_dbg = nullptr;
#define PRO_TM() do { if (ntm < 8) tmk[ntm++] = clock64(); } while (0)
#else
#define PRO_TM() do {} while (0)
#endif
template <typename T, typename TI>
__global__ __launch_bounds__(1024) void gen_prologue_kernel(const TI* __restrict__ mel, const T* __restrict__ att_w, const T* __restrict__ att_b,
                                                           const TI* __restrict__ spk, const TI* __restrict__ emo, const T* __restrict__ film_w,
                                                           const T* __restrict__ film_b, float* __restrict__ alpha, T* __restrict__ x_cl,
                                                           T* __restrict__ film_proj, float* __restrict__ zero_buf, long zero_n, int B,
                                                           int C, int Tn, int K, int ds, int de, int cond_dim, int F2) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nt = blockDim.x, nw = nt >> 6;
#ifdef MV_PRO_TIMING
  long long tmk[8]; int ntm = 0;
#endif
  PRO_TM();

  if ((int)blockIdx.x >= B) {
    // ---- FiLM role: rows j = rg * nw + wid, samples s0 .. s0 + PRO_SG
    const int fw = blockIdx.x - B, n_rg = (F2 + nw - 1) / nw;
    const int rg = fw % n_rg, s0 = (fw / n_rg) * PRO_SG;
    const int j = rg * nw + wid;
    float* cl = sm;                                              // [PRO_SG][cond_dim] the condition vectors
    const bool fast = cond_dim <= 64 * PRO_FI;
    float wv[PRO_FI];
    if (fast) {
#pragma unroll
      for (int i = 0; i < PRO_FI; ++i)
        wv[i] = (j < F2 && lane + 64 * i < cond_dim) ? ld<T>(film_w + (long)j * cond_dim + lane + 64 * i) : 0.f;
    }
    const float bj = (film_b && j < F2) ? ld<T>(film_b + j) : 0.f;
    for (int e = tid; e < PRO_SG * cond_dim; e += nt) {          // cat(spk, emo), truncated / zero-padded (grc_lora.py:82-105)
      const int s = e / cond_dim, i = e - s * cond_dim, b = s0 + s;
      float v = 0.f;
      if (b < B) {
        if (i < ds) v = ld<TI>(spk + (long)b * ds + i);
        else if (i < ds + de) v = ld<TI>(emo + (long)b * de + (i - ds));
      }
      cl[e] = v;
    }
    __syncthreads();
    float a[PRO_SG];
    if (fast) {
#pragma unroll
      for (int s = 0; s < PRO_SG; ++s) {
        a[s] = 0.f;
#pragma unroll
        for (int i = 0; i < PRO_FI; ++i) if (lane + 64 * i < cond_dim) a[s] += wv[i] * cl[s * cond_dim + lane + 64 * i];
      }
    } else {
#pragma unroll
      for (int s = 0; s < PRO_SG; ++s) a[s] = 0.f;
      if (j < F2)
        for (int i = lane; i < cond_dim; i += 64) {
          const float w = ld<T>(film_w + (long)j * cond_dim + i);
#pragma unroll
          for (int s = 0; s < PRO_SG; ++s) a[s] += w * cl[s * cond_dim + i];
        }
    }
#pragma unroll
    for (int s = 0; s < PRO_SG; ++s) a[s] = wave_sum(a[s]);      // PRO_SG independent trees: their shuffles overlap
    if (j < F2 && lane < PRO_SG && s0 + lane < B) {
      float v = a[0];
#pragma unroll
      for (int s = 1; s < PRO_SG; ++s) v = lane == s ? a[s] : v;
      st<T>(film_proj + (long)(s0 + lane) * F2 + j, v + bj);
    }
    return;
  }

  // ---- sample role
  const int b = blockIdx.x;
  const int TP = Tn | 1;                 // odd row stride: the [C][T] tile is read row-wise AND column-wise without bank conflicts
  const int P = nt / C;                  // time slices of the mean (>= 1: the host checks C <= blockDim)
  float* xs = sm;                        // [C][TP]
  float* psum = xs + C * TP;             // [P][C] slice sums
  float* mean = psum + P * C;            // [C]
  const TI* xb = mel + (long)b * C * Tn;
  // every global load is issued before the first barrier: the mel and (wave 0) the attention weights, into registers
  if (C * Tn <= 4 * nt) {
    float mv4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) mv4[u] = tid + u * nt < C * Tn ? ld<TI>(xb + tid + u * nt) : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * nt, c = i / Tn;
      if (i < C * Tn) xs[c * TP + (i - c * Tn)] = mv4[u];
    }
  } else {
    for (int i = tid; i < C * Tn; i += nt) { const int c = i / Tn; xs[c * TP + (i - c * Tn)] = ld<TI>(xb + i); }
  }
  constexpr int AK = 4, AWN = 4;         // attention weights wave 0 keeps: banks < AK, channels lane + 64 i, i < AWN
  const bool areg = K <= AK && C <= 64 * AWN;
  float aw[AK][AWN], ab[AK];
  if (wid == 0 && areg) {
#pragma unroll
    for (int k = 0; k < AK; ++k) {
#pragma unroll
      for (int i = 0; i < AWN; ++i) aw[k][i] = (k < K && lane + 64 * i < C) ? ld<T>(att_w + (long)k * C + lane + 64 * i) : 0.f;
      ab[k] = (k < K && att_b) ? ld<T>(att_b + k) : 0.f;
    }
  }
  {
    const long per = (zero_n + B - 1) / B, z0 = (long)b * per;
    const long z1 = z0 + per < zero_n ? z0 + per : zero_n;
    for (long i = z0 + tid; i < z1; i += nt) zero_buf[i] = 0.f;
  }
  PRO_TM();
  __syncthreads();
  PRO_TM();
  // channels-last copy
  T* yb = x_cl + (long)b * Tn * C;
  for (int i = tid; i < C * Tn; i += nt) {
    const int t = i / C, c = i - t * C;
    st<T>(yb + i, xs[c * TP + t]);
  }
  // mean over T (odconv.py:36): P time slices per channel, then the P slice sums in a fixed order
  if (tid < P * C) {
    const int part = tid / C, c = tid - part * C;
    float a = 0.f;
    for (int t = part; t < Tn; t += P) a += xs[c * TP + t];
    psum[tid] = a;
  }
  PRO_TM();
  __syncthreads();
  if (tid < C) {
    float a = 0.f;
    for (int q = 0; q < P; ++q) a += psum[q * C + tid];
    mean[tid] = a / (float)Tn;
  }
  __syncthreads();
  PRO_TM();
  // logits + softmax by wave 0 (odconv.py:37-40)
  if (wid == 0) {
    if (areg) {
      float lg[AK];
#pragma unroll
      for (int k = 0; k < AK; ++k) {
        lg[k] = 0.f;
#pragma unroll
        for (int i = 0; i < AWN; ++i) if (lane + 64 * i < C) lg[k] += aw[k][i] * mean[lane + 64 * i];
      }
#pragma unroll
      for (int k = 0; k < AK; ++k) lg[k] = wave_sum(lg[k]) + ab[k];
      float m = -INFINITY, den = 0.f;
#pragma unroll
      for (int k = 0; k < AK; ++k) if (k < K) m = fmaxf(m, lg[k]);
#pragma unroll
      for (int k = 0; k < AK; ++k) { lg[k] = k < K ? expf(lg[k] - m) : 0.f; den += lg[k]; }
      if (lane < K) {
        float v = lg[0];
#pragma unroll
        for (int k = 1; k < AK; ++k) v = lane == k ? lg[k] : v;
        alpha[(long)b * K + lane] = v / den;
      }
    } else {                              // any K <= 64: lane k ends up with logit k
      float mine = -INFINITY;
      for (int k = 0; k < K; ++k) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += ld<T>(att_w + (long)k * C + c) * mean[c];
        a = wave_sum(a) + (att_b ? ld<T>(att_b + k) : 0.f);
        if (lane == k) mine = a;
      }
      const float m = wave_max(mine);
      const float e = lane < K ? expf(mine - m) : 0.f;
      const float den = wave_sum(e);
      if (lane < K) alpha[(long)b * K + lane] = e / den;
    }
  }
#ifdef MV_PRO_TIMING
  PRO_TM();
  if (tid == 0 && pro_dbg) for (int i = 0; i < 8; ++i) pro_dbg[b * 8 + i] = i < ntm ? tmk[i] - tmk[0] : -1;
#endif
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

extern "C" int mv_gen_prologue_in(const void* mel, const void* att_w, const void* att_b, const void* spk, const void* emo,
                                  const void* film_w, const void* film_b, float* alpha, void* x_cl, void* film_proj, float* zero_buf,
                                  long zero_n, int B, int C, int T_, int K, int ds, int de, int cond_dim, int F2, int in_dtype, int dtype,
                                  void* stream);
extern "C" int mv_gen_prologue(const void* mel, const void* att_w, const void* att_b, const void* spk, const void* emo,
                               const void* film_w, const void* film_b, float* alpha, void* x_cl, void* film_proj, float* zero_buf,
                               long zero_n, int B, int C, int T_, int K, int ds, int de, int cond_dim, int F2, int dtype, void* stream) {
  return mv_gen_prologue_in(mel, att_w, att_b, spk, emo, film_w, film_b, alpha, x_cl, film_proj, zero_buf, zero_n, B, C, T_, K, ds, de,
                            cond_dim, F2, dtype, dtype, stream);
}

extern "C" int mv_gen_prologue_in(const void* mel, const void* att_w, const void* att_b, const void* spk, const void* emo,
                                  const void* film_w, const void* film_b, float* alpha, void* x_cl, void* film_proj, float* zero_buf,
                                  long zero_n, int B, int C, int T_, int K, int ds, int de, int cond_dim, int F2, int in_dtype, int dtype,
                                  void* stream) {
  if (in_dtype != dtype && in_dtype != MV_F32) return MV_ERR_DTYPE;      // inputs (mel, spk, emo): the storage type, or fp32
  MV_CHECK_ARG(mel && att_w && alpha && x_cl && B > 0 && C > 0 && T_ > 0 && K > 0 && K <= 64 && ds >= 0 && de >= 0 && zero_n >= 0);
  MV_CHECK_ARG((ds == 0 || spk) && (de == 0 || emo) && (!film_proj || (film_w && cond_dim > 0 && F2 > 0)) && (zero_n == 0 || zero_buf));
  constexpr int NT = 1024, NW = NT / 64;
  if (C > NT) return MV_ERR_UNSUPPORTED;
  const size_t lds_sample = sizeof(float) * ((size_t)C * (T_ | 1) + (size_t)(NT / C) * C + C);
  const size_t lds_film = film_proj ? sizeof(float) * (size_t)PRO_SG * cond_dim : 0;
  const size_t lds = lds_sample > lds_film ? lds_sample : lds_film;
  if (lds > 64 * 1024) return MV_ERR_UNSUPPORTED;   // long inputs: the caller issues the separate launches
  const int n_film = film_proj ? ((F2 + NW - 1) / NW) * ((B + PRO_SG - 1) / PRO_SG) : 0;
#ifdef MV_PRO_TIMING
  static long long* dbg = nullptr;
  static int calls = 0;
  if (!dbg) { hipMalloc(&dbg, 1024 * 8 * 8); hipMemcpyToSymbol(HIP_SYMBOL(pro_dbg), &dbg, sizeof(dbg)); }
#endif
  MV_DISPATCH(dtype, {
    if (in_dtype == dtype)
      hipLaunchKernelGGL((gen_prologue_kernel<T, T>), dim3(B + n_film), dim3(NT), lds, (hipStream_t)stream, (const T*)mel,
                         (const T*)att_w, (const T*)att_b, (const T*)spk, (const T*)emo, (const T*)film_w, (const T*)film_b, alpha,
                         (T*)x_cl, (T*)film_proj, zero_buf, zero_n, B, C, T_, K, ds, de, film_proj ? cond_dim : 0, F2);
    else
      hipLaunchKernelGGL((gen_prologue_kernel<T, float>), dim3(B + n_film), dim3(NT), lds, (hipStream_t)stream, (const float*)mel,
                         (const T*)att_w, (const T*)att_b, (const float*)spk, (const float*)emo, (const T*)film_w, (const T*)film_b,
                         alpha, (T*)x_cl, (T*)film_proj, zero_buf, zero_n, B, C, T_, K, ds, de, film_proj ? cond_dim : 0, F2);
  });
#ifdef MV_PRO_TIMING
  if (++calls == 2 && B <= 1024) {
    hipStreamSynchronize((hipStream_t)stream);
    static long long hbuf[1024 * 8];
    hipMemcpy(hbuf, dbg, sizeof(long long) * 8 * B, hipMemcpyDeviceToHost);
    double avg[8] = {0};
    for (int w = 0; w < B; ++w) for (int i = 0; i < 8; ++i) avg[i] += (double)hbuf[w * 8 + i] / B;
    fprintf(stderr, "[prologue timing] dtype %d marks (start, mel issued, film, sync1, copy, means+sync2, logits+softmax):", dtype);
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %.0f", avg[i]);
    fprintf(stderr, "\n");
  }
#endif
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

// ---- zero fill as a kernel (common.h: mvi_zero_async)
namespace mv {
__global__ __launch_bounds__(256) void zero_fill_kernel(uint32_t* __restrict__ p, size_t n32, size_t tail_bytes) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n128 = n32 / 4;
  u32x4* p4 = reinterpret_cast<u32x4*>(p);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n128; i += stride) p4[i] = u32x4{0u, 0u, 0u, 0u};
  if (blockIdx.x == 0 && threadIdx.x < 4) {
    const size_t i = n128 * 4 + threadIdx.x;
    if (i < n32) p[i] = 0u;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned char* b = reinterpret_cast<unsigned char*>(p) + n32 * 4;
    for (size_t k = 0; k < tail_bytes; ++k) b[k] = 0;
  }
}
}  // namespace mv

hipError_t mvi_zero_async(void* p, size_t bytes, hipStream_t stream) {
  if (!bytes) return hipSuccess;
  if (((uintptr_t)p & 15) != 0) return hipMemsetAsync(p, 0, bytes, stream);      // (every caller passes 16-byte aligned buffers)
  const size_t n32 = bytes / 4, tail = bytes % 4;
  size_t blocks = (n32 / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mv::zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<uint32_t*>(p), n32, tail);
  return hipGetLastError();
}
