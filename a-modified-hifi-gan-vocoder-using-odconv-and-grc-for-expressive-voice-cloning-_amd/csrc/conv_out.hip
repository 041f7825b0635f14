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

// w_t[j][c] = w[0][c][j] as fp32 (tiny)
template <typename P>
__global__ void conv_out_pack_kernel(const P* __restrict__ w, float* __restrict__ wt, int C, int ks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C * ks) { const int j = i / C, c = i % C; wt[i] = ld<P>(w + c * ks + j); }
}

}  // namespace mv

using namespace mv;

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
