// Generic-shape direct convolutions (any channel count / kernel / stride / dilation / groups), NCT layout.
// These are the "every shape works" path behind the drop-in modules and the fp32 reference the MFMA
// kernels are cross-checked against on the GPU; the C2/C3 hot shapes go through the fused MFMA kernels.
//
// Tiling: one workgroup = (sample b, CO_T=32 output channels of one group, TT=128 output steps).
// 256 threads, each owns a 4(co) x 4(t) register tile; the input window and the (alpha-aggregated)
// weights of CI_T input channels are staged in LDS per step.  ODConv (odconv.py:89-106) is the same
// kernel with nbanks=K: the per-sample kernel sum_k alpha[b,k] W[k] is formed while staging weights,
// so attention -> aggregate -> conv is one launch and the per-sample kernel never touches HBM.
#include "common.h"

namespace mv {

constexpr int TT = 128;    // output time steps per workgroup
constexpr int CO_T = 32;   // output channels per workgroup
constexpr int CI_T = 8;    // input channels staged per step
constexpr int MAX_BANKS = 8;

struct ConvP {
  int B, Cin, Tin, Cout, Tout, ks, stride, pad, dil, groups, nbanks, act;
  float slope;
  long x_bs, x_cs, y_bs, y_cs;
  int xw;  // staged input window width
};

template <typename T>
__global__ __launch_bounds__(256) void conv1d_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                         const T* __restrict__ bias, const float* __restrict__ alpha,
                                                         const T* __restrict__ res, T* __restrict__ y, ConvP p) {
  extern __shared__ __align__(16) float smem[];
  float* xs = smem;                       // [CI_T][xw]
  float* ws = smem + CI_T * p.xw;         // [CI_T][ks][CO_T]
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int b = blockIdx.z;
  const int cout_g = p.Cout / p.groups, cin_g = p.Cin / p.groups;
  const int co_tiles = cdiv(cout_g, CO_T);
  const int g = blockIdx.y / co_tiles;
  const int co0 = (blockIdx.y % co_tiles) * CO_T;  // within group
  const int t0 = blockIdx.x * TT;

  float al[MAX_BANKS];
#pragma unroll
  for (int k = 0; k < MAX_BANKS; ++k) al[k] = (k < p.nbanks) ? (alpha ? alpha[b * p.nbanks + k] : 1.f) : 0.f;

  float acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;

  const T* xb = x + (long)b * p.x_bs + (long)(g * cin_g) * p.x_cs;
  const long wbank = (long)p.Cout * cin_g * p.ks;
  const int tin0 = t0 * p.stride - p.pad;

  for (int ci0 = 0; ci0 < cin_g; ci0 += CI_T) {
    for (int idx = tid; idx < CI_T * p.xw; idx += 256) {
      const int ci = idx / p.xw, xi = idx - ci * p.xw;
      const int tin = tin0 + xi;
      float v = 0.f;
      if (ci0 + ci < cin_g && tin >= 0 && tin < p.Tin) v = ld<T>(xb + (long)(ci0 + ci) * p.x_cs + tin);
      xs[idx] = v;
    }
    const int nw = CO_T * CI_T * p.ks;
    for (int idx = tid; idx < nw; idx += 256) {
      const int k = idx % p.ks;
      const int ci = (idx / p.ks) % CI_T;
      const int co = idx / (p.ks * CI_T);
      float v = 0.f;
      if (co0 + co < cout_g && ci0 + ci < cin_g) {
        const long off = ((long)(g * cout_g + co0 + co) * cin_g + (ci0 + ci)) * p.ks + k;
        for (int kb = 0; kb < p.nbanks; ++kb) v += al[kb] * ld<T>(w + kb * wbank + off);
      }
      ws[(ci * p.ks + k) * CO_T + co] = v;
    }
    __syncthreads();
    for (int ci = 0; ci < CI_T; ++ci) {
      const float* xr = xs + ci * p.xw;
      for (int k = 0; k < p.ks; ++k) {
        const float4 wv = *reinterpret_cast<const float4*>(ws + (ci * p.ks + k) * CO_T + ty * 4);
        const int xo = k * p.dil;
        float xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = xr[(tx + 32 * i) * p.stride + xo];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[0][i] += wv.x * xv[i];
          acc[1][i] += wv.y * xv[i];
          acc[2][i] += wv.z * xv[i];
          acc[3][i] += wv.w * xv[i];
        }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = co0 + ty * 4 + j;
    if (co >= cout_g) continue;
    const int cog = g * cout_g + co;
    float bv = 0.f;
    if (bias)
      for (int kb = 0; kb < p.nbanks; ++kb) bv += al[kb] * ld<T>(bias + kb * p.Cout + cog);
    T* yr = y + (long)b * p.y_bs + (long)cog * p.y_cs;
    const T* rr = res ? res + (long)b * p.y_bs + (long)cog * p.y_cs : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = t0 + tx + 32 * i;
      if (t < p.Tout) {
        float v = apply_act(acc[j][i] + bv, p.act, p.slope);
        if (rr) v += ld<T>(rr + t);
        st<T>(yr + t, v);
      }
    }
  }
}

// Transposed conv, gather form: y[b,o,u] = sum_{c,j : (u+pad-j*dil) % stride == 0} w[c,o,j] x[b,c,(u+pad-j*dil)/stride]
template <typename T>
__global__ __launch_bounds__(256) void conv_transpose1d_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                                   const T* __restrict__ bias,
                                                                   const float* __restrict__ alpha, T* __restrict__ y,
                                                                   ConvP p, int tlo_pad) {
  extern __shared__ __align__(16) float smem[];
  float* xs = smem;                       // [CI_T][xw]
  float* ws = smem + CI_T * p.xw;         // [CI_T][ks][CO_T]
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int b = blockIdx.z;
  const int co0 = blockIdx.y * CO_T;
  const int u0 = blockIdx.x * TT;

  float al[MAX_BANKS];
#pragma unroll
  for (int k = 0; k < MAX_BANKS; ++k) al[k] = (k < p.nbanks) ? (alpha ? alpha[b * p.nbanks + k] : 1.f) : 0.f;

  float acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;

  // lowest input step any output of this tile can touch (floor division, may be negative)
  const int num_lo = u0 + p.pad - (p.ks - 1) * p.dil;
  const int tlo = (num_lo >= 0) ? num_lo / p.stride : -((-num_lo + p.stride - 1) / p.stride);
  const T* xb = x + (long)b * p.x_bs;
  const long wbank = (long)p.Cin * p.Cout * p.ks;

  for (int ci0 = 0; ci0 < p.Cin; ci0 += CI_T) {
    for (int idx = tid; idx < CI_T * p.xw; idx += 256) {
      const int ci = idx / p.xw, xi = idx - ci * p.xw;
      const int tin = tlo + xi;
      float v = 0.f;
      if (ci0 + ci < p.Cin && tin >= 0 && tin < p.Tin) v = ld<T>(xb + (long)(ci0 + ci) * p.x_cs + tin);
      xs[idx] = v;
    }
    const int nw = CI_T * CO_T * p.ks;
    for (int idx = tid; idx < nw; idx += 256) {
      const int k = idx % p.ks;
      const int co = (idx / p.ks) % CO_T;
      const int ci = idx / (p.ks * CO_T);
      float v = 0.f;
      if (co0 + co < p.Cout && ci0 + ci < p.Cin) {
        const long off = ((long)(ci0 + ci) * p.Cout + (co0 + co)) * p.ks + k;
        for (int kb = 0; kb < p.nbanks; ++kb) v += al[kb] * ld<T>(w + kb * wbank + off);
      }
      ws[(ci * p.ks + k) * CO_T + co] = v;
    }
    __syncthreads();
    for (int ci = 0; ci < CI_T; ++ci) {
      const float* xr = xs + ci * p.xw;
      for (int k = 0; k < p.ks; ++k) {
        const float4 wv = *reinterpret_cast<const float4*>(ws + (ci * p.ks + k) * CO_T + ty * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int num = u0 + tx + 32 * i + p.pad - k * p.dil;
          if (num >= 0 && (num % p.stride) == 0) {
            const int xi = num / p.stride - tlo;  // tin < Tin is enforced by the zero-filled staging
            if (xi < p.xw) {
              const float xv = xr[xi];
              acc[0][i] += wv.x * xv;
              acc[1][i] += wv.y * xv;
              acc[2][i] += wv.z * xv;
              acc[3][i] += wv.w * xv;
            }
          }
        }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = co0 + ty * 4 + j;
    if (co >= p.Cout) continue;
    float bv = 0.f;
    if (bias)
      for (int kb = 0; kb < p.nbanks; ++kb) bv += al[kb] * ld<T>(bias + kb * p.Cout + co);
    T* yr = y + (long)b * p.y_bs + (long)co * p.y_cs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = u0 + tx + 32 * i;
      if (u < p.Tout) st<T>(yr + u, apply_act(acc[j][i] + bv, p.act, p.slope));
    }
  }
}

// 2-D conv, stride 1.  Workgroup = (b*H' + h, co tile, w tile); stages kh input rows.
struct Conv2dP {
  int B, Cin, H, W, Cout, Ho, Wo, kh, kw, ph, pw, act;
  float slope;
  int xw;
};

template <typename T>
__global__ __launch_bounds__(256) void conv2d_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                         const T* __restrict__ bias, T* __restrict__ y, Conv2dP p) {
  extern __shared__ __align__(16) float smem[];
  const int kk = p.kh * p.kw;
  float* xs = smem;                            // [CI_T][kh][xw]
  float* ws = smem + CI_T * p.kh * p.xw;       // [CI_T][kh*kw][CO_T]
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int b = blockIdx.z / p.Ho, ho = blockIdx.z % p.Ho;
  const int co0 = blockIdx.y * CO_T;
  const int w0 = blockIdx.x * TT;

  float acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;

  const T* xb = x + (long)b * p.Cin * p.H * p.W;
  for (int ci0 = 0; ci0 < p.Cin; ci0 += CI_T) {
    const int nx = CI_T * p.kh * p.xw;
    for (int idx = tid; idx < nx; idx += 256) {
      const int xi = idx % p.xw;
      const int r = (idx / p.xw) % p.kh;
      const int ci = idx / (p.xw * p.kh);
      const int hi = ho - p.ph + r, wi = w0 - p.pw + xi;
      float v = 0.f;
      if (ci0 + ci < p.Cin && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
        v = ld<T>(xb + ((long)(ci0 + ci) * p.H + hi) * p.W + wi);
      xs[idx] = v;
    }
    const int nw = CO_T * CI_T * kk;
    for (int idx = tid; idx < nw; idx += 256) {
      const int k = idx % kk;
      const int ci = (idx / kk) % CI_T;
      const int co = idx / (kk * CI_T);
      float v = 0.f;
      if (co0 + co < p.Cout && ci0 + ci < p.Cin) v = ld<T>(w + ((long)(co0 + co) * p.Cin + (ci0 + ci)) * kk + k);
      ws[(ci * kk + k) * CO_T + co] = v;
    }
    __syncthreads();
    for (int ci = 0; ci < CI_T; ++ci) {
      for (int r = 0; r < p.kh; ++r) {
        const float* xr = xs + (ci * p.kh + r) * p.xw;
        for (int k = 0; k < p.kw; ++k) {
          const float4 wv = *reinterpret_cast<const float4*>(ws + (ci * kk + r * p.kw + k) * CO_T + ty * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float xv = xr[tx + 32 * i + k];
            acc[0][i] += wv.x * xv;
            acc[1][i] += wv.y * xv;
            acc[2][i] += wv.z * xv;
            acc[3][i] += wv.w * xv;
          }
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = co0 + ty * 4 + j;
    if (co >= p.Cout) continue;
    const float bv = bias ? ld<T>(bias + co) : 0.f;
    T* yr = y + (((long)b * p.Cout + co) * p.Ho + ho) * p.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int wo = w0 + tx + 32 * i;
      if (wo < p.Wo) st<T>(yr + wo, apply_act(acc[j][i] + bv, p.act, p.slope));
    }
  }
}

}  // namespace mv

using namespace mv;

extern "C" int mv_conv1d_fwd(const void* x, const void* w, const void* bias, const float* alpha, const void* res,
                             void* y, int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad,
                             int dil, int groups, int nbanks, int act, float slope, long x_bs, long x_cs, long y_bs,
                             long y_cs, int dtype, void* stream) {
  MV_CHECK_ARG(x && w && y && B > 0 && Cin > 0 && Cout > 0 && Tin > 0 && ks > 0 && stride > 0 && dil > 0 && pad >= 0);
  MV_CHECK_ARG(groups > 0 && Cin % groups == 0 && Cout % groups == 0);
  MV_CHECK_ARG(nbanks >= 1 && nbanks <= MAX_BANKS && (nbanks == 1 || alpha != nullptr));
  MV_CHECK_ARG(Tout == (Tin + 2 * pad - dil * (ks - 1) - 1) / stride + 1 && Tout > 0);
  ConvP p{B, Cin, Tin, Cout, Tout, ks, stride, pad, dil, groups, nbanks, act, slope, x_bs, x_cs, y_bs, y_cs, 0};
  p.xw = (TT - 1) * stride + (ks - 1) * dil + 1;
  const size_t lds = sizeof(float) * ((size_t)CI_T * p.xw + (size_t)CI_T * ks * CO_T);
  if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
  dim3 grid(cdiv(Tout, TT), groups * cdiv(Cout / groups, CO_T), B);
  MV_DISPATCH(dtype, {
    auto kern = conv1d_fwd_kernel<T>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, (const T*)x, (const T*)w, (const T*)bias, alpha,
                       (const T*)res, (T*)y, p);
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_conv_transpose1d_fwd(const void* x, const void* w, const void* bias, const float* alpha, void* y,
                                       int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad,
                                       int dil, int nbanks, int act, float slope, int dtype, void* stream) {
  MV_CHECK_ARG(x && w && y && B > 0 && Cin > 0 && Cout > 0 && Tin > 0 && ks > 0 && stride > 0 && dil > 0 && pad >= 0);
  MV_CHECK_ARG(nbanks >= 1 && nbanks <= MAX_BANKS && (nbanks == 1 || alpha != nullptr));
  const int full = (Tin - 1) * stride - 2 * pad + dil * (ks - 1) + 1;
  MV_CHECK_ARG(Tout >= full && Tout < full + stride && Tout > 0);  // output_padding < stride
  ConvP p{B, Cin, Tin, Cout, Tout, ks, stride, pad, dil, 1, nbanks, act, slope,
          (long)Cin * Tin, (long)Tin, (long)Cout * Tout, (long)Tout, 0};
  p.xw = (TT - 1 + (ks - 1) * dil) / stride + 2;
  const size_t lds = sizeof(float) * ((size_t)CI_T * p.xw + (size_t)CI_T * ks * CO_T);
  if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
  dim3 grid(cdiv(Tout, TT), cdiv(Cout, CO_T), B);
  MV_DISPATCH(dtype, {
    auto kern = conv_transpose1d_fwd_kernel<T>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, (const T*)x, (const T*)w, (const T*)bias, alpha,
                       (T*)y, p, 0);
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_conv2d_fwd(const void* x, const void* w, const void* bias, void* y, int B, int Cin, int H, int W,
                             int Cout, int kh, int kw, int ph, int pw, int act, float slope, int dtype, void* stream) {
  MV_CHECK_ARG(x && w && y && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && ph >= 0 && pw >= 0);
  const int Ho = H + 2 * ph - kh + 1, Wo = W + 2 * pw - kw + 1;
  MV_CHECK_ARG(Ho > 0 && Wo > 0 && (long)B * Ho <= 65535);
  Conv2dP p{B, Cin, H, W, Cout, Ho, Wo, kh, kw, ph, pw, act, slope, TT + kw - 1};
  const size_t lds = sizeof(float) * ((size_t)CI_T * kh * p.xw + (size_t)CI_T * kh * kw * CO_T);
  if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
  dim3 grid(cdiv(Wo, TT), cdiv(Cout, CO_T), B * Ho);
  MV_DISPATCH(dtype, {
    auto kern = conv2d_fwd_kernel<T>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, (const T*)x, (const T*)w, (const T*)bias,
                       (T*)y, p);
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}
