// Backward kernels of the generic (any-shape, NCT) primitives.  Parameter gradients are always produced in fp32.
// Math: SURVEY.md Appendix B.1-B.5 (ODConv: gW[k] = sum_b alpha[b,k] gW~_b, galpha[b,k] = <gW~_b, W[k]> + <gb~_b, bias[k]>,
// softmax/pooling chain; GroupNorm; FiLM).  Data gradients of convolutions re-use the forward kernels
// (dgrad of conv1d = transposed conv with the same weights and vice versa), see hifigan_modified/functional.py.
#include "common.h"

namespace mv {

// ------------------------------------------------------------------------------------------- activation backward
// gx = gy * act'(.) expressed through the OUTPUT y of the activation (lrelu: sign(y) = sign(pre-activation); tanh: 1 - y^2)
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ y,
                                                      T* __restrict__ gx, long n, int act, float slope) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = ld<T>(gy + i), v = ld<T>(y + i);
    float d = 1.f;
    if (act == ACT_LRELU) d = v >= 0.f ? 1.f : slope;
    else if (act == ACT_TANH) d = 1.f - v * v;
    st<T>(gx + i, g * d);
  }
}

// ------------------------------------------------------------------------------------------- conv weight gradient
// conv1d:  y[b,o,t] = sum_{c,j} w[o,c,j] x[b,c,t*stride - pad + j*dil]   (groups = 1)
// gws[b?][o][c][j] = sum_t gy[b,o,t] x[b,c,t*stride-pad+j*dil]
// One workgroup = 16(o) x 16(c) weight tile, all taps (ks <= 16), looping over time; per-sample results are
//   nbanks == 1: gw [Cout][Cin][ks] += tile
//   nbanks  > 1: gw[k] += alpha[b,k] * tile ;  galpha[b,k] += <tile, w[k] tile>   (both linear in the tile, so partial
//                time chunks can be added independently; fp32 atomics)
constexpr int WG_TT = 64;
constexpr int WG_MAXKS = 16;

template <typename T>
__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                           const T* __restrict__ w, const float* __restrict__ alpha,
                                                           float* __restrict__ gw, float* __restrict__ galpha,
                                                           int B, int Cin, int Tin, int Cout, int Tout, int ks,
                                                           int stride, int pad, int dil, int nbanks,
                                                           long x_bs, long x_cs, long g_bs, long g_cs, int tsplit, int tchunk,
                                                           int per_sample) {
  // grid.z = B * tsplit: every workgroup reduces one (sample, time chunk) and adds its tile with fp32 atomics
  extern __shared__ __align__(16) float sm[];
  const int xw = (WG_TT - 1) * stride + (ks - 1) * dil + 1;
  float* xs = sm;                 // [16][xw]
  float* gs = sm + 16 * xw;       // [16][WG_TT]
  float* red = gs + 16 * WG_TT;   // [32]
  const int tid = threadIdx.x, ol = tid >> 4, cl = tid & 15;
  const int o = blockIdx.y * 16 + ol, c = blockIdx.x * 16 + cl;
  const long wbank = (long)Cout * Cin * ks;
  const int b = blockIdx.z / tsplit, tc = blockIdx.z % tsplit;
  const int tbeg = tc * tchunk, tend = (tbeg + tchunk < Tout) ? tbeg + tchunk : Tout;
  float acc[WG_MAXKS];
#pragma unroll
  for (int j = 0; j < WG_MAXKS; ++j) acc[j] = 0.f;
  for (int t0 = tbeg; t0 < tend; t0 += WG_TT) {
    const int tin0 = t0 * stride - pad;
    for (int i = tid; i < 16 * xw; i += 256) {
      const int cc = i / xw, xi = i % xw, ci = blockIdx.x * 16 + cc, tin = tin0 + xi;
      xs[i] = (ci < Cin && tin >= 0 && tin < Tin) ? ld<T>(x + (long)b * x_bs + (long)ci * x_cs + tin) : 0.f;
    }
    for (int i = tid; i < 16 * WG_TT; i += 256) {
      const int oo = i / WG_TT, ti = i % WG_TT, co = blockIdx.y * 16 + oo, t = t0 + ti;
      gs[i] = (co < Cout && t < tend) ? ld<T>(gy + (long)b * g_bs + (long)co * g_cs + t) : 0.f;
    }
    __syncthreads();
    for (int t = 0; t < WG_TT; ++t) {
      const float g = gs[ol * WG_TT + t];
      const float* xr = xs + cl * xw + t * stride;
#pragma unroll
      for (int j = 0; j < WG_MAXKS; ++j)
        if (j < ks) acc[j] += g * xr[j * dil];
    }
    __syncthreads();
  }
  const bool ok = o < Cout && c < Cin;
  if (per_sample) {           // gw = per-sample tiles [B][Cout][Cin][ks]; the alpha chain is applied by odconv_wgrad_reduce_kernel
    if (ok) {
      float* gwk = gw + (long)b * wbank + ((long)o * Cin + c) * ks;
#pragma unroll
      for (int j = 0; j < WG_MAXKS; ++j)
        if (j < ks) { if (tsplit == 1) gwk[j] = acc[j]; else atomicAdd(gwk + j, acc[j]); }
    }
  } else if (nbanks == 1) {
    if (ok) {
      float* gwk = gw + ((long)o * Cin + c) * ks;
#pragma unroll
      for (int j = 0; j < WG_MAXKS; ++j)
        if (j < ks) atomicAdd(gwk + j, acc[j]);
    }
  } else {
    for (int k = 0; k < nbanks; ++k) {
      const float a = alpha[b * nbanks + k];
      float dot = 0.f;
      if (ok) {
        float* gwk = gw + k * wbank + ((long)o * Cin + c) * ks;
        const T* wk = w + k * wbank + ((long)o * Cin + c) * ks;
#pragma unroll
        for (int j = 0; j < WG_MAXKS; ++j)
          if (j < ks) { atomicAdd(gwk + j, a * acc[j]); dot += acc[j] * ld<T>(wk + j); }
      }
      dot = block_sum(dot, red);
      if (tid == 0) atomicAdd(galpha + b * nbanks + k, dot);
    }
  }
}

// ODConv bank gradients from per-sample tiles: gw[k][e] = sum_b alpha[b,k] gws[b][e];
// galpha[b,k] += <gws[b], w[k]>.  One thread per weight element e, a block covers `epb` elements.
template <typename T>
__global__ __launch_bounds__(256) void odconv_wgrad_reduce_kernel(const float* __restrict__ gws, const T* __restrict__ w,
                                                                  const float* __restrict__ alpha, float* __restrict__ gw,
                                                                  float* __restrict__ galpha, int B, int K, long nelem, int epb) {
  extern __shared__ float dots[];              // [B][K] block-local <gws[b], w[k]>
  for (int i = threadIdx.x; i < B * K; i += blockDim.x) dots[i] = 0.f;
  __syncthreads();
  const long e0 = (long)blockIdx.x * epb;
  const int lane = threadIdx.x & 63;
  for (long e = e0 + threadIdx.x; e < e0 + epb; e += blockDim.x) {
    const bool ok = e < nelem;
    float wk[8], acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { wk[k] = (ok && k < K) ? ld<T>(w + (long)k * nelem + e) : 0.f; acc[k] = 0.f; }
    for (int b = 0; b < B; ++b) {
      const float g = ok ? gws[(long)b * nelem + e] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < K) {
          acc[k] += alpha[b * K + k] * g;
          const float d = wave_sum(g * wk[k]);
          if (lane == 0) atomicAdd(dots + b * K + k, d);
        }
    }
    if (ok)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < K) gw[(long)k * nelem + e] = acc[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < B * K; i += blockDim.x) atomicAdd(galpha + i, dots[i]);
}

// Same reduction, 4 consecutive elements per thread (K <= 4, nelem % 4 == 0): one 16-byte load of the sample's tile per step, and the
// B x K wave reductions of the alpha-gradient dots run once per 4 elements instead of once per element (they were the bulk of the
// instruction count: 6 shuffles per (sample, bank) against one load and K FMAs).
template <typename T>
__global__ __launch_bounds__(256) void odconv_wgrad_reduce4_kernel(const float* __restrict__ gws, const T* __restrict__ w,
                                                                   const float* __restrict__ alpha, float* __restrict__ gw,
                                                                   float* __restrict__ galpha, int B, int K, long nelem) {
  extern __shared__ float dots[];              // [B][K]
  for (int i = threadIdx.x; i < B * K; i += blockDim.x) dots[i] = 0.f;
  __syncthreads();
  const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const bool ok = e < nelem;
  const int lane = threadIdx.x & 63;
  float wk[4][4], acc[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) { wk[k][j] = (ok && k < K) ? ld<T>(w + (long)k * nelem + e + j) : 0.f; acc[k][j] = 0.f; }
  for (int b = 0; b < B; ++b) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) g = *reinterpret_cast<const float4*>(gws + (long)b * nelem + e);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < K) {
        const float a = alpha[b * K + k];
        acc[k][0] += a * g.x; acc[k][1] += a * g.y; acc[k][2] += a * g.z; acc[k][3] += a * g.w;
        const float d = wave_sum(g.x * wk[k][0] + g.y * wk[k][1] + g.z * wk[k][2] + g.w * wk[k][3]);
        if (lane == 0) atomicAdd(dots + b * K + k, d);
      }
  }
  if (ok)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < K) *reinterpret_cast<float4*>(gw + (long)k * nelem + e) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
  __syncthreads();
  for (int i = threadIdx.x; i < B * K; i += blockDim.x) atomicAdd(galpha + i, dots[i]);
}

// rowsum[b][o] = sum_t gy[b,o,t]
template <typename T>
__global__ __launch_bounds__(256) void rowsum_kernel(const T* __restrict__ gy, float* __restrict__ out, int C, int Tn,
                                                     long g_bs, long g_cs, long nrows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int b = (int)(row / C), c = (int)(row % C);
  float s = 0.f;
  const T* p = gy + (long)b * g_bs + (long)c * g_cs;
  for (int t = lane; t < Tn; t += 64) s += ld<T>(p + t);
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

// gbias[k][o] = sum_b alpha[b,k] rs[b][o] ; galpha[b][k] += sum_o rs[b][o] bias[k][o]
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ rs, const float* __restrict__ alpha,
                                                        const T* __restrict__ bias, float* __restrict__ gbias,
                                                        float* __restrict__ galpha, int B, int C, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K * C) {
    const int k = i / C, o = i % C;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += (alpha ? alpha[b * K + k] : 1.f) * rs[(long)b * C + o];
    gbias[i] = s;
  }
  if (galpha && i < B * K) {
    const int b = i / K, k = i % K;
    float s = 0.f;
    for (int o = 0; o < C; ++o) s += rs[(long)b * C + o] * ld<T>(bias + (long)k * C + o);
    galpha[i] += s;
  }
}

// ------------------------------------------------------------------------------------------- ODConv attention backward
// gz = alpha * (galpha - sum_k alpha galpha); gWa[k][c] = sum_b gz[b,k] m[b,c]; gba[k] = sum_b gz[b,k];
// gm[b][c] = (1/T) sum_k Wa[k][c] gz[b,k]   (added to every time step of gx)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ alpha, const float* __restrict__ galpha,
                                                       const float* __restrict__ pooled, const T* __restrict__ wa,
                                                       float* __restrict__ gwa, float* __restrict__ gba,
                                                       float* __restrict__ gm, int B, int C, int K, float inv_t) {
  extern __shared__ float gz[];  // [B][K]
  for (int i = threadIdx.x; i < B * K; i += blockDim.x) {
    const int b = i / K;
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += alpha[b * K + k] * galpha[b * K + k];
    gz[i] = alpha[i] * (galpha[i] - dot);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
    const int k = i / C, c = i % C;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += gz[b * K + k] * pooled[(long)b * C + c];
    gwa[i] = s;
  }
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += gz[b * K + k];
    gba[k] = s;
  }
  for (int i = threadIdx.x; i < B * C; i += blockDim.x) {
    const int b = i / C, c = i % C;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += ld<T>(wa + (long)k * C + c) * gz[b * K + k];
    gm[i] = s * inv_t;
  }
}

// x[b,c,:] += v[b][c]
template <typename T>
__global__ __launch_bounds__(256) void add_rowconst_kernel(T* __restrict__ x, const float* __restrict__ v, int Tn) {
  const long row = blockIdx.y;
  const float a = v[row];
  T* p = x + row * Tn;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Tn; t += gridDim.x * blockDim.x) st<T>(p + t, ld<T>(p + t) + a);
}

// ------------------------------------------------------------------------------------------- GroupNorm backward
// forward: xh = (x-mu)*rstd ; z = xh*gw+gb ; y = act(z) [* mask*scale] [+ res]
// input gz = dL/dz (activation / mask already applied by the caller through act_bwd / gn_pre_bwd)
// phase 1 (per (b,g) workgroup): s1 = sum gz*gw, s2 = sum gz*gw*xh ; per-(b,c) sums dgb[b][c] = sum_t gz, dgw[b][c] = sum_t gz*xh
// phase 2: gx = rstd * (gz*gw - s1/n - xh*s2/n)
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ gz,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const T* __restrict__ gw, float* __restrict__ s12,
                                                           float* __restrict__ dgw_bc, float* __restrict__ dgb_bc,
                                                           int C, int Tn, int G, long x_bs, long x_cs, long g_bs, long g_cs) {
  __shared__ float red[32];
  const int b = blockIdx.x / G, g = blockIdx.x % G, cg = C / G;
  const float mu = mean[blockIdx.x], rs = rstd[blockIdx.x];
  float s1 = 0.f, s2 = 0.f;
  for (int cc = 0; cc < cg; ++cc) {
    const int c = g * cg + cc;
    const float gam = gw ? ld<T>(gw + c) : 1.f;
    float a = 0.f, bsum = 0.f;
    const T* gr = gz + (long)b * g_bs + (long)c * g_cs;
    const T* xr = x + (long)b * x_bs + (long)c * x_cs;
    if (sizeof(T) == 2 && all_mult8(Tn, x_bs, x_cs, g_bs, g_cs) && (((uintptr_t)x | (uintptr_t)gz) & 15) == 0) {
      for (int t = threadIdx.x * 8; t < Tn; t += blockDim.x * 8) {
        float gv[8], xv[8];
        load8f<T>(gr + t, gv);
        load8f<T>(xr + t, xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { a += gv[e] * (xv[e] - mu) * rs; bsum += gv[e]; }
      }
    } else {
      for (int t = threadIdx.x; t < Tn; t += blockDim.x) {
        const float gv = ld<T>(gr + t);
        const float xh = (ld<T>(xr + t) - mu) * rs;
        a += gv * xh;
        bsum += gv;
      }
    }
    a = block_sum(a, red);
    bsum = block_sum(bsum, red);
    if (threadIdx.x == 0) { dgw_bc[(long)b * C + c] = a; dgb_bc[(long)b * C + c] = bsum; }
    s1 += gam * bsum;
    s2 += gam * a;
  }
  if (threadIdx.x == 0) { s12[blockIdx.x * 2] = s1; s12[blockIdx.x * 2 + 1] = s2; }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ gz,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const T* __restrict__ gw, const float* __restrict__ s12,
                                                           T* __restrict__ gx, int C, int Tn, int G, long x_bs, long x_cs,
                                                           long g_bs, long g_cs) {
  const int bc = blockIdx.y, b = bc / C, c = bc % C, g = c / (C / G);
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  const float inv_n = 1.f / ((float)(C / G) * (float)Tn);
  const float s1 = s12[(b * G + g) * 2] * inv_n, s2 = s12[(b * G + g) * 2 + 1] * inv_n;
  const float gam = gw ? ld<T>(gw + c) : 1.f;
  const T* gr = gz + (long)b * g_bs + (long)c * g_cs;
  const T* xr = x + (long)b * x_bs + (long)c * x_cs;
  T* or_ = gx + (long)bc * Tn;
  if (sizeof(T) == 2 && all_mult8(Tn, x_bs, x_cs, g_bs, g_cs) && (((uintptr_t)x | (uintptr_t)gz | (uintptr_t)gx) & 15) == 0) {
    for (int t = (blockIdx.x * blockDim.x + threadIdx.x) * 8; t < Tn; t += gridDim.x * blockDim.x * 8) {
      float gv[8], xv[8];
      load8f<T>(gr + t, gv);
      load8f<T>(xr + t, xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) gv[e] = rs * (gv[e] * gam - s1 - (xv[e] - mu) * rs * s2);
      store8f<T>(or_ + t, gv);
    }
    return;
  }
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Tn; t += gridDim.x * blockDim.x) {
    const float gv = ld<T>(gr + t);
    const float xh = (ld<T>(xr + t) - mu) * rs;
    st<T>(or_ + t, rs * (gv * gam - s1 - xh * s2));
  }
}

// gz = gy * d(act)/dz * mask*scale, where z = (x-mu)*rstd*gw+gb is recomputed from x (SiLU needs z itself)
template <typename T>
__global__ __launch_bounds__(256) void gn_pre_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const T* __restrict__ gw, const T* __restrict__ gb,
                                                         const uint8_t* __restrict__ mask, float mask_scale,
                                                         T* __restrict__ gz, int C, int Tn, int G, int act, float slope,
                                                         long x_bs, long x_cs, long g_bs, long g_cs) {
  const int bc = blockIdx.y, b = bc / C, c = bc % C, g = c / (C / G);
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  const float a = rs * (gw ? ld<T>(gw + c) : 1.f);
  const float sh = (gb ? ld<T>(gb + c) : 0.f) - mu * a;
  const T* gr = gy + (long)b * g_bs + (long)c * g_cs;
  const T* xr = x + (long)b * x_bs + (long)c * x_cs;
  const uint8_t* mr = mask ? mask + (long)bc * Tn : nullptr;
  T* or_ = gz + (long)bc * Tn;
  auto dact = [&](float z) {
    float d = 1.f;
    if (act == ACT_SILU) { const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-z)); d = sg * (1.f + z * (1.f - sg)); }
    else if (act == ACT_LRELU) d = z >= 0.f ? 1.f : slope;
    else if (act == ACT_TANH) { const float th = tanhf(z); d = 1.f - th * th; }
    return d;
  };
  if (sizeof(T) == 2 && all_mult8(Tn, x_bs, x_cs, g_bs, g_cs) && (((uintptr_t)x | (uintptr_t)gy | (uintptr_t)gz) & 15) == 0) {
    for (int t = (blockIdx.x * blockDim.x + threadIdx.x) * 8; t < Tn; t += gridDim.x * blockDim.x * 8) {
      float gv[8], xv[8];
      load8f<T>(gr + t, gv);
      load8f<T>(xr + t, xv);
      uint2 m8 = {0u, 0u};
      if (mr) m8 = *reinterpret_cast<const uint2*>(mr + t);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float g1 = gv[e];
        if (mr) g1 = (((e < 4 ? m8.x : m8.y) >> (8 * (e & 3))) & 0xffu) ? g1 * mask_scale : 0.f;
        gv[e] = g1 * dact(xv[e] * a + sh);
      }
      store8f<T>(or_ + t, gv);
    }
    return;
  }
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Tn; t += gridDim.x * blockDim.x) {
    float gv = ld<T>(gr + t);
    if (mr) gv = mr[t] ? gv * mask_scale : 0.f;
    st<T>(or_ + t, gv * dact(ld<T>(xr + t) * a + sh));
  }
}

// out[c] = sum_b in[b][c]
__global__ void colsum_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { float s = 0.f; for (int b = 0; b < B; ++b) s += in[(long)b * C + c]; out[c] = s; }
}

// ------------------------------------------------------------------------------------------- FiLM / linear backward
// y = x*gamma + beta (c < F), pass-through otherwise.  gx = gy*gamma ; gproj[b][c] = sum_t gy*x ; gproj[b][F+c] = sum_t gy
template <typename T>
__global__ __launch_bounds__(256) void film_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                       const T* __restrict__ proj, T* __restrict__ gx,
                                                       float* __restrict__ gproj, int C, int Tn, int F) {
  __shared__ float red[32];
  const int bc = blockIdx.x, b = bc / C, c = bc % C;
  const float gam = c < F ? ld<T>(proj + (long)b * 2 * F + c) : 1.f;
  float a = 0.f, s = 0.f;
  for (int t = threadIdx.x; t < Tn; t += blockDim.x) {
    const float g = ld<T>(gy + (long)bc * Tn + t);
    a += g * ld<T>(x + (long)bc * Tn + t);
    s += g;
    st<T>(gx + (long)bc * Tn + t, g * gam);
  }
  if (c < F) {
    a = block_sum(a, red);
    s = block_sum(s, red);
    if (threadIdx.x == 0) { gproj[(long)b * 2 * F + c] = a; gproj[(long)b * 2 * F + F + c] = s; }
  }
}

// y[m,n] = sum_k x[m,k] w[n,k] + b[n]:  gx[m,k] = sum_n gy[m,n] w[n,k]; gw[n,k] = sum_m gy[m,n] x[m,k]; gb[n] = sum_m gy[m,n]
template <typename T>
__global__ __launch_bounds__(256) void linear_bwd_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                         const float* __restrict__ gy, float* __restrict__ gx,
                                                         float* __restrict__ gw, float* __restrict__ gb, int M, int N, int Kd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gx && i < (long)M * Kd) {
    const int m = (int)(i / Kd), k = (int)(i % Kd);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += gy[(long)m * N + n] * ld<T>(w + (long)n * Kd + k);
    gx[i] = s;
  }
  if (i < (long)N * Kd) {
    const int n = (int)(i / Kd), k = (int)(i % Kd);
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += gy[(long)m * N + n] * ld<T>(x + (long)m * Kd + k);
    gw[i] = s;
  }
  if (gb && i < N) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += gy[(long)m * N + i];
    gb[i] = s;
  }
}

// ------------------------------------------------------------------------------------------- pooling / fold backward
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int Tn, int To, int s) {
  const long row = blockIdx.y;
  const float inv = 1.f / (float)s;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Tn; t += gridDim.x * blockDim.x) {
    const int q = t / s;
    st<T>(gx + row * Tn + t, q < To ? ld<T>(gy + row * To + q) * inv : 0.f);
  }
}

// strided 2-D copy: dst[r][0..n) = src[r][0..n)  (row strides in elements) - cat / slice / fold-backward
template <typename T>
__global__ __launch_bounds__(256) void copy2d_kernel(const T* __restrict__ src, T* __restrict__ dst, int n, long s_rs, long d_rs,
                                                     int rows_inner, long s_os, long d_os) {
  // row index r = (outer, inner): address = outer*?_os + inner*?_rs
  const int r = blockIdx.y, outer = r / rows_inner, inner = r % rows_inner;
  const T* s = src + outer * s_os + inner * s_rs;
  T* d = dst + outer * d_os + inner * d_rs;
  // whole 16-byte pieces when the row geometry allows it (bytes: the element type does not matter for a copy)
  constexpr int EPV = 16 / sizeof(T);
  if (n % EPV == 0 && s_rs % EPV == 0 && d_rs % EPV == 0 && s_os % EPV == 0 && d_os % EPV == 0 &&
      (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const uint4* s4 = reinterpret_cast<const uint4*>(s);
    uint4* d4 = reinterpret_cast<uint4*>(d);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n / EPV; i += gridDim.x * blockDim.x) d4[i] = s4[i];
    return;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = s[i];
}

// conv2d weight transform for dgrad: wt[c][o][kh-1-i][kw-1-j] = w[o][c][i][j]
template <typename T>
__global__ void conv2d_flip_kernel(const T* __restrict__ w, T* __restrict__ wt, int Cout, int Cin, int kh, int kw) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int kk = kh * kw;
  if (idx < Cout * Cin * kk) {
    const int j = idx % kw, i = (idx / kw) % kh, c = (idx / kk) % Cin, o = idx / (kk * Cin);
    wt[((long)c * Cout + o) * kk + (kh - 1 - i) * kw + (kw - 1 - j)] = w[idx];
  }
}

// conv2d wgrad (stride 1): gw[o][c][i][j] = sum_{b,h,w} gy[b,o,h,w] x[b,c,h+i-ph,w+j-pw]; one workgroup per (o, c-tile of 4)
template <typename T>
__global__ __launch_bounds__(256) void conv2d_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                           float* __restrict__ gw, int B, int Cin, int H, int W, int Cout,
                                                           int Ho, int Wo, int kh, int kw, int ph, int pw) {
  __shared__ float red[32];
  const int o = blockIdx.x, c = blockIdx.y;
  const int kk = kh * kw;
  for (int tap = 0; tap < kk; ++tap) {
    const int i = tap / kw, j = tap % kw;
    float s = 0.f;
    const long total = (long)B * Ho * Wo;
    for (long idx = threadIdx.x; idx < total; idx += blockDim.x) {
      const int wo = (int)(idx % Wo), ho = (int)((idx / Wo) % Ho), b = (int)(idx / ((long)Wo * Ho));
      const int hi = ho + i - ph, wi = wo + j - pw;
      if (hi >= 0 && hi < H && wi >= 0 && wi < W)
        s += ld<T>(gy + (((long)b * Cout + o) * Ho + ho) * Wo + wo) * ld<T>(x + (((long)b * Cin + c) * H + hi) * W + wi);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) gw[((long)o * Cin + c) * kk + tap] = s;
  }
}

}  // namespace mv

using namespace mv;

static inline int grid_for(long n, int block = 256, int cap = 2048) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

extern "C" int mv_act_bwd(const void* gy, const void* y, void* gx, long n, int act, float slope, int dtype, void* stream) {
  MV_CHECK_ARG(gy && y && gx && n > 0);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                                        (const T*)gy, (const T*)y, (T*)gx, n, act, slope));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" size_t mv_conv1d_wgrad_workspace_bytes(int B, int Cin, int Cout, int ks, int nbanks) {
  return nbanks > 1 ? sizeof(float) * (size_t)B * Cout * Cin * ks : 0;
}

extern "C" int mv_conv1d_wgrad(const void* x, const void* gy, const void* w, const float* alpha, float* gw, float* galpha,
                               float* workspace, int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad,
                               int dil, int nbanks, long x_bs, long x_cs, long g_bs, long g_cs, int dtype, void* stream) {
  MV_CHECK_ARG(x && gy && gw && B > 0 && Cin > 0 && Cout > 0 && Tin > 0 && Tout > 0 && ks > 0 && ks <= WG_MAXKS);
  MV_CHECK_ARG(nbanks >= 1 && nbanks <= 8 && (nbanks == 1 || (alpha && galpha && w && workspace)));
  const int xw = (WG_TT - 1) * stride + (ks - 1) * dil + 1;
  const size_t lds = sizeof(float) * (16 * xw + 16 * WG_TT + 32);
  if (lds > 64 * 1024) return MV_ERR_UNSUPPORTED;
  // split time so that ~2048 workgroups exist; partial tiles are combined with fp32 atomics
  const int tiles = cdiv(Cin, 16) * cdiv(Cout, 16);
  int tsplit = cdiv(2048, tiles * B);
  const int max_split = cdiv(Tout, WG_TT);
  if (tsplit > max_split) tsplit = max_split;
  if (tsplit < 1) tsplit = 1;
  const int tchunk = cdiv(cdiv(Tout, tsplit), WG_TT) * WG_TT;
  tsplit = cdiv(Tout, tchunk);
  if ((long)B * tsplit > 65535) return MV_ERR_UNSUPPORTED;
  dim3 grid(cdiv(Cin, 16), cdiv(Cout, 16), B * tsplit);
  const long nelem = (long)Cout * Cin * ks;
  const int per_sample = nbanks > 1;
  float* dst = per_sample ? workspace : gw;
  // ODConv (nbanks > 1): per-sample tiles first (no cross-sample contention), then one reduction applies the alpha chain
  if (!per_sample || tsplit > 1)
    MV_HIP(mvi_zero_async(dst, sizeof(float) * (size_t)(per_sample ? B : 1) * nelem, (hipStream_t)stream));
  MV_DISPATCH(dtype, {
    hipLaunchKernelGGL(conv1d_wgrad_kernel<T>, grid, dim3(256), lds, (hipStream_t)stream, (const T*)x, (const T*)gy,
                       (const T*)w, alpha, dst, galpha, B, Cin, Tin, Cout, Tout, ks, stride, pad, dil, nbanks, x_bs, x_cs,
                       g_bs, g_cs, tsplit, tchunk, per_sample);
    if (per_sample) {
      if (nbanks <= 4 && nelem % 4 == 0 && (((uintptr_t)workspace | (uintptr_t)gw) & 15) == 0) {
        hipLaunchKernelGGL(odconv_wgrad_reduce4_kernel<T>, dim3((unsigned)((nelem / 4 + 255) / 256)), dim3(256),
                           sizeof(float) * B * nbanks, (hipStream_t)stream, workspace, (const T*)w, alpha, gw, galpha, B, nbanks, nelem);
      } else {
        const int epb = 256;       // one element per thread: the B x K wave reductions of a block stay short
        hipLaunchKernelGGL(odconv_wgrad_reduce_kernel<T>, dim3((unsigned)((nelem + epb - 1) / epb)), dim3(256),
                           sizeof(float) * B * nbanks, (hipStream_t)stream, workspace, (const T*)w, alpha, gw, galpha, B,
                           nbanks, nelem, epb);
      }
    }
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_odconv_wgrad_reduce(const float* gws, const void* w, const float* alpha, float* gw, float* galpha, int B, int K,
                                      long nelem, int dtype, void* stream) {
  MV_CHECK_ARG(gws && w && alpha && gw && galpha && B > 0 && K >= 1 && K <= 8 && nelem > 0);
  if (K <= 4 && nelem % 4 == 0 && (((uintptr_t)gws | (uintptr_t)gw) & 15) == 0) {
    MV_DISPATCH(dtype, hipLaunchKernelGGL(odconv_wgrad_reduce4_kernel<T>, dim3((unsigned)((nelem / 4 + 255) / 256)), dim3(256),
                                          sizeof(float) * B * K, (hipStream_t)stream, gws, (const T*)w, alpha, gw, galpha, B, K, nelem));
    MV_LAUNCH_CHECK();
    return MV_OK;
  }
  const int epb = 256;
  MV_DISPATCH(dtype, hipLaunchKernelGGL(odconv_wgrad_reduce_kernel<T>, dim3((unsigned)((nelem + epb - 1) / epb)), dim3(256),
                                        sizeof(float) * B * K, (hipStream_t)stream, gws, (const T*)w, alpha, gw, galpha, B, K, nelem, epb));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_bias_grad(const void* gy, const float* alpha, const void* bias, float* rowsum_ws, float* gbias,
                            float* galpha, int B, int C, int T_, int K, long g_bs, long g_cs, int dtype, void* stream) {
  MV_CHECK_ARG(gy && rowsum_ws && gbias && B > 0 && C > 0 && T_ > 0 && K >= 1 && (K == 1 || alpha));
  const long rows = (long)B * C;
  MV_DISPATCH(dtype, {
    hipLaunchKernelGGL(rowsum_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)gy,
                       rowsum_ws, C, T_, g_bs, g_cs, rows);
    const int n = K * C > B * K ? K * C : B * K;
    hipLaunchKernelGGL(bias_grad_kernel<T>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rowsum_ws, alpha,
                       (const T*)bias, gbias, (K > 1 && bias) ? galpha : nullptr, B, C, K);
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_odconv_attn_bwd(const float* alpha, const float* galpha, const float* pooled, const void* wa,
                                  float* gwa, float* gba, float* gm, int B, int C, int T_, int K, int dtype, void* stream) {
  MV_CHECK_ARG(alpha && galpha && pooled && wa && gwa && gba && gm && B > 0 && C > 0 && K > 0 && T_ > 0);
  const size_t lds = sizeof(float) * B * K;
  MV_CHECK_ARG(lds <= 64 * 1024);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(attn_bwd_kernel<T>, dim3(1), dim3(256), lds, (hipStream_t)stream, alpha, galpha,
                                        pooled, (const T*)wa, gwa, gba, gm, B, C, K, 1.f / (float)T_));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_add_rowconst(void* x, const float* v, long rows, int T_, int dtype, void* stream) {
  MV_CHECK_ARG(x && v && rows > 0 && rows <= 65535 && T_ > 0);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(add_rowconst_kernel<T>, dim3(grid_for(T_, 256, 64), (unsigned)rows), dim3(256), 0,
                                        (hipStream_t)stream, (T*)x, v, T_));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_groupnorm_bwd(const void* x, const void* gy, const float* mean, const float* rstd, const void* gw,
                                const void* gb, const uint8_t* mask, float mask_scale, int act, float slope,
                                void* gz_ws, float* ws, void* gx, float* dgw, float* dgb, int B, int C, int T_, int G,
                                long x_bs, long x_cs, long g_bs, long g_cs, int dtype, void* stream) {
  MV_CHECK_ARG(x && gy && mean && rstd && gz_ws && ws && gx && B > 0 && C > 0 && T_ > 0 && G > 0 && C % G == 0);
  MV_CHECK_ARG((long)B * C <= 65535);
  float* s12 = ws;                       // [B*G*2]
  float* dgw_bc = ws + (size_t)B * G * 2;  // [B*C]
  float* dgb_bc = dgw_bc + (size_t)B * C;  // [B*C]
  const long dense_bs = (long)C * T_, dense_cs = T_;
  MV_DISPATCH(dtype, {
    const bool vec = sizeof(T) == 2 && all_mult8(T_, x_bs, x_cs, g_bs, g_cs) &&
                     (((uintptr_t)x | (uintptr_t)gy | (uintptr_t)gz_ws | (uintptr_t)gx) & 15) == 0;
    dim3 ge(grid_for(vec ? cdiv(T_, 8) : T_, 256, 64), B * C);
    hipLaunchKernelGGL(gn_pre_bwd_kernel<T>, ge, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)gy, mean, rstd,
                       (const T*)gw, (const T*)gb, mask, mask_scale, (T*)gz_ws, C, T_, G, act, slope, x_bs, x_cs, g_bs, g_cs);
    hipLaunchKernelGGL(gn_bwd_stats_kernel<T>, dim3(B * G), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)gz_ws,
                       mean, rstd, (const T*)gw, s12, dgw_bc, dgb_bc, C, T_, G, x_bs, x_cs, dense_bs, dense_cs);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<T>, ge, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)gz_ws, mean, rstd,
                       (const T*)gw, s12, (T*)gx, C, T_, G, x_bs, x_cs, dense_bs, dense_cs);
    if (dgw) hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dgw_bc, dgw, B, C);
    if (dgb) hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dgb_bc, dgb, B, C);
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_film_bwd(const void* x, const void* gy, const void* proj, void* gx, float* gproj, int B, int C, int T_,
                           int F, int dtype, void* stream) {
  MV_CHECK_ARG(x && gy && proj && gx && gproj && B > 0 && C > 0 && T_ > 0 && F > 0);
  MV_HIP(mvi_zero_async(gproj, sizeof(float) * (size_t)B * 2 * F, (hipStream_t)stream));
  MV_DISPATCH(dtype, hipLaunchKernelGGL(film_bwd_kernel<T>, dim3(B * C), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                        (const T*)gy, (const T*)proj, (T*)gx, gproj, C, T_, F));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_linear_bwd(const void* x, const void* w, const float* gy, float* gx, float* gw, float* gb, int M, int N,
                             int Kd, int dtype, void* stream) {
  MV_CHECK_ARG(x && w && gy && gw && M > 0 && N > 0 && Kd > 0);
  long n = (long)M * Kd > (long)N * Kd ? (long)M * Kd : (long)N * Kd;
  if (n < N) n = N;
  MV_DISPATCH(dtype, hipLaunchKernelGGL(linear_bwd_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                                        (hipStream_t)stream, (const T*)x, (const T*)w, gy, gx, gw, gb, M, N, Kd));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_avgpool1d_bwd(const void* gy, void* gx, long rows, int T_, int s, int dtype, void* stream) {
  MV_CHECK_ARG(gy && gx && rows > 0 && rows <= 65535 && s > 0 && T_ >= s);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(grid_for(T_, 256, 64), (unsigned)rows), dim3(256), 0,
                                        (hipStream_t)stream, (const T*)gy, (T*)gx, T_, T_ / s, s));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_copy2d(const void* src, void* dst, int n, int rows_outer, int rows_inner, long s_os, long s_rs, long d_os,
                         long d_rs, int dtype, void* stream) {
  MV_CHECK_ARG(src && dst && n > 0 && rows_outer > 0 && rows_inner > 0 && (long)rows_outer * rows_inner <= 65535);
  const int epv = dtype == MV_F32 ? 4 : 8;
  const bool vec = n % epv == 0 && s_rs % epv == 0 && d_rs % epv == 0 && s_os % epv == 0 && d_os % epv == 0 &&
                   (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  MV_DISPATCH(dtype, hipLaunchKernelGGL(copy2d_kernel<T>, dim3(grid_for(vec ? n / epv : n, 256, 64), rows_outer * rows_inner), dim3(256), 0,
                                        (hipStream_t)stream, (const T*)src, (T*)dst, n, s_rs, d_rs, rows_inner, s_os, d_os));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_conv2d_flip_weights(const void* w, void* wt, int Cout, int Cin, int kh, int kw, int dtype, void* stream) {
  MV_CHECK_ARG(w && wt && Cout > 0 && Cin > 0 && kh > 0 && kw > 0);
  const int n = Cout * Cin * kh * kw;
  MV_DISPATCH(dtype, hipLaunchKernelGGL(conv2d_flip_kernel<T>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                                        (const T*)w, (T*)wt, Cout, Cin, kh, kw));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_conv2d_wgrad(const void* x, const void* gy, float* gw, int B, int Cin, int H, int W, int Cout, int kh,
                               int kw, int ph, int pw, int dtype, void* stream) {
  MV_CHECK_ARG(x && gy && gw && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && Cin <= 65535);
  const int Ho = H + 2 * ph - kh + 1, Wo = W + 2 * pw - kw + 1;
  MV_DISPATCH(dtype, hipLaunchKernelGGL(conv2d_wgrad_kernel<T>, dim3(Cout, Cin), dim3(256), 0, (hipStream_t)stream,
                                        (const T*)x, (const T*)gy, gw, B, Cin, H, W, Cout, Ho, Wo, kh, kw, ph, pw));
  MV_LAUNCH_CHECK();
  return MV_OK;
}
