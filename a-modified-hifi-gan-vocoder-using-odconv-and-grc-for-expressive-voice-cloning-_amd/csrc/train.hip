// Training-side kernels: GRC weight-fold backward, fused GAN / L1 / hinge losses (value + gradient in one pass),
// log-mel spectrogram loss (forward + backward), flat-arena AdamW and the multi-tensor gradient gather.
//   losses:   hifigan_modified/complete_vocoder.py:89-184 (LSGAN + L1), conditioned_hifigan.py:254-267 (hinge)
//   AdamW:    conditioned_hifigan.py:219 (torch.optim.AdamW semantics: decoupled weight decay, bias correction)
//   mel loss: defined by this build (the reference only has placeholders)
#include "common.h"
#include <cstdlib>

namespace mv {

// ------------------------------------------------------------------------------------------- dropout keep-mask (Philox4x32-10)
// mask[i] = 1 with probability 1-p (nn.Dropout of grc_lora.py:151,162).  Counter-based: element block i/8 is counter (i/8, 0, 0, 0)
// under key (seed lo, seed hi), 16 random bits per element - the mask depends only on (seed, index), not on the launch geometry.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ mask, long n, uint32_t thresh16, uint32_t k0, uint32_t k1) {
  const long nblk = (n + 7) / 8;
  for (long blk = (long)blockIdx.x * blockDim.x + threadIdx.x; blk < nblk; blk += (long)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u, k0, k1, r);
    uint8_t m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = ((r[e >> 1] >> (16 * (e & 1))) & 0xffffu) >= thresh16 ? 1 : 0;   // keep iff u >= p
    if (blk * 8 + 8 <= n) *reinterpret_cast<uint2*>(mask + blk * 8) = *reinterpret_cast<const uint2*>(m);
    else for (int e = 0; blk * 8 + e < n; ++e) mask[blk * 8 + e] = m[e];
  }
}

// ------------------------------------------------------------------------------------------- GRC fold backward
// forward (norm_elem.hip grc_fold_kernel): comb[o'][c][j] = [c in grp(o')] Wc[o'][c_loc][j] + [j==mid] s (A B)[c][o']
//   w_eff[o][c][j] = sum_o' Wp[o][o'] comb[o'][c][j];  b_eff[o] = sum_o' Wp[o][o'] bc[o'] + bp[o]
template <typename P>
__global__ __launch_bounds__(1024) void grc_fold_bwd_kernel(const float* __restrict__ g_weff, const float* __restrict__ g_beff,
                                                           const P* __restrict__ conv_w, const P* __restrict__ conv_b,
                                                           const P* __restrict__ A, const P* __restrict__ Bm,
                                                           const P* __restrict__ scal, const P* __restrict__ proj_w,
                                                           float* __restrict__ g_conv_w, float* __restrict__ g_conv_b,
                                                           float* __restrict__ g_A, float* __restrict__ g_B,
                                                           float* __restrict__ g_s, float* __restrict__ g_proj_w,
                                                           float* __restrict__ g_proj_b, int Cin, int Cout, int ks,
                                                           int groups, int rank) {
  extern __shared__ float sm[];
  float* gcomb = sm;                         // [Cout][Cin][ks]
  float* L = gcomb + Cout * Cin * ks;        // [Cin][Cout]
  float* red = L + Cin * Cout;               // [32]
  float* gwl = red + 32;                     // [Cout][Cin][ks] copy of g_weff: every later loop re-reads it many times
  const int tid = threadIdx.x, nt = blockDim.x;
  const int cin_g = Cin / groups, cout_g = Cout / groups, mid = ks / 2;
  // parameter copies in LDS: the loops below are chains of dependent reads (192 per g_proj_w element) - out of global memory
  // each link was an L2 round trip (44 us for the whole kernel)
  float* Wl = gwl + Cout * Cin * ks;         // conv_w [Cout][cin_g][ks]
  float* Pl = Wl + Cout * cin_g * ks;        // proj_w [Cout][Cout]
  float* Al = Pl + Cout * Cout;              // A [Cin][rank]
  float* Bl = Al + Cin * rank;               // B [rank][Cout]
  const float s = ld<P>(scal);
  for (int i = tid; i < Cout * Cin * ks; i += nt) gwl[i] = g_weff[i];
  for (int i = tid; i < Cout * cin_g * ks; i += nt) Wl[i] = ld<P>(conv_w + i);
  for (int i = tid; i < Cout * Cout; i += nt) Pl[i] = ld<P>(proj_w + i);
  for (int i = tid; i < Cin * rank; i += nt) Al[i] = ld<P>(A + i);
  for (int i = tid; i < rank * Cout; i += nt) Bl[i] = ld<P>(Bm + i);
  __syncthreads();
  for (int i = tid; i < Cin * Cout; i += nt) {
    const int c = i / Cout, o = i % Cout;
    float l = 0.f;
    for (int r = 0; r < rank; ++r) l += Al[c * rank + r] * Bl[r * Cout + o];
    L[i] = l;
  }
  for (int i = tid; i < Cout * Cin * ks; i += nt) {
    const int j = i % ks, c = (i / ks) % Cin, op = i / (ks * Cin);
    float a = 0.f;
    for (int o = 0; o < Cout; ++o) a += Pl[o * Cout + op] * gwl[(o * Cin + c) * ks + j];
    gcomb[i] = a;
  }
  __syncthreads();
  // proj_w / proj_b / conv_b
  for (int i = tid; i < Cout * Cout; i += nt) {
    const int o = i / Cout, op = i % Cout;
    const int g = op / cout_g;
    float a = g_beff[o] * ld<P>(conv_b + op);
    for (int c = 0; c < Cin; ++c)
      for (int j = 0; j < ks; ++j) {
        float comb = 0.f;
        if (c / cin_g == g) comb = Wl[(op * cin_g + (c - g * cin_g)) * ks + j];
        if (j == mid) comb += s * L[c * Cout + op];
        a += gwl[(o * Cin + c) * ks + j] * comb;
      }
    g_proj_w[i] = a;
  }
  for (int o = tid; o < Cout; o += nt) {
    g_proj_b[o] = g_beff[o];
    float a = 0.f;
    for (int oo = 0; oo < Cout; ++oo) a += Pl[oo * Cout + o] * g_beff[oo];
    g_conv_b[o] = a;
  }
  for (int i = tid; i < Cout * cin_g * ks; i += nt) {
    const int j = i % ks, cl = (i / ks) % cin_g, op = i / (ks * cin_g);
    const int c = (op / cout_g) * cin_g + cl;
    g_conv_w[i] = gcomb[((long)op * Cin + c) * ks + j];
  }
  for (int i = tid; i < Cin * rank; i += nt) {
    const int c = i / rank, r = i % rank;
    float a = 0.f;
    for (int op = 0; op < Cout; ++op) a += s * gcomb[((long)op * Cin + c) * ks + mid] * Bl[r * Cout + op];
    g_A[i] = a;
  }
  for (int i = tid; i < rank * Cout; i += nt) {
    const int r = i / Cout, op = i % Cout;
    float a = 0.f;
    for (int c = 0; c < Cin; ++c) a += Al[c * rank + r] * s * gcomb[((long)op * Cin + c) * ks + mid];
    g_B[i] = a;
  }
  float part = 0.f;
  for (int i = tid; i < Cin * Cout; i += nt) {
    const int c = i / Cout, op = i % Cout;
    part += L[i] * gcomb[((long)op * Cin + c) * ks + mid];
  }
  part = block_sum(part, red);
  if (tid == 0) g_s[0] = part;
}

// ------------------------------------------------------------------------------------------- losses (value + gradient)
// kind 0: mean((x-c)^2)            (LSGAN, complete_vocoder.py:104,157-158)
// kind 1: mean(|x-y|)              (L1 / "feature matching", :118,127)          gy (optional) = -gx
// kind 2: mean(relu(1 - x))        (hinge generator, conditioned_hifigan.py:263)
// kind 3: mean(relu(1 + x))        (hinge discriminator-fake, :265)
// kind 4: mean((x-y)^2)            (MSE between tensors, conditioned_hifigan.py:238)
// loss_acc[0] += weight * value ; gx = weight * d value / dx
template <typename T>
__global__ __launch_bounds__(256) void loss_kernel(const T* __restrict__ x, const T* __restrict__ y, float c, float weight,
                                                   float* __restrict__ loss_acc, T* __restrict__ gx, T* __restrict__ gy,
                                                   long n, int kind) {
  __shared__ float red[32];
  const float inv = weight / (float)n;
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = ld<T>(x + i);
    float l, g;
    if (kind == 0) { const float d = v - c; l = d * d; g = 2.f * d; }
    else if (kind == 1) { const float d = v - ld<T>(y + i); l = fabsf(d); g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
    else if (kind == 2) { const float d = 1.f - v; l = d > 0.f ? d : 0.f; g = d > 0.f ? -1.f : 0.f; }
    else if (kind == 3) { const float d = 1.f + v; l = d > 0.f ? d : 0.f; g = d > 0.f ? 1.f : 0.f; }
    else { const float d = v - ld<T>(y + i); l = d * d; g = 2.f * d; }
    s += l;
    if (gx) st<T>(gx + i, g * inv);
    if (gy) st<T>(gy + i, -g * inv);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(loss_acc, s * inv);
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T* __restrict__ x, const float* __restrict__ factor_dev, float factor, long n) {
  const float f = factor_dev ? factor * factor_dev[0] : factor;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    st<T>(x + i, ld<T>(x + i) * f);
}

template <typename T>
__global__ __launch_bounds__(256) void scale_to_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ factor_dev,
                                                       float factor, long n) {
  const float f = factor_dev ? factor * factor_dev[0] : factor;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    st<T>(y + i, ld<T>(x + i) * f);
}

// ------------------------------------------------------------------------------------------- AdamW on a flat fp32 arena
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2, float gscale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gr = g[i] * gscale;
    float pi = p[i] * (1.f - lr * wd);                 // decoupled weight decay (torch.optim.AdamW)
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
  }
}

// the same with the step count read from device memory (a captured HIP graph must not bake the host's count into its launch arguments)
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                        float wd, const int* __restrict__ step, float gscale) {
  const float t = (float)(*step);
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gr = g[i] * gscale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
  }
}

// gather many gradient tensors (any storage type, given per tensor) into one flat fp32 buffer
struct GatherDesc { const void* src; long dst_off; long n; int dtype; int pad; };
__global__ __launch_bounds__(256) void multi_gather_kernel(const GatherDesc* __restrict__ descs, float* __restrict__ flat, int chunk) {
  const GatherDesc d = descs[blockIdx.y];
  const long beg = (long)blockIdx.x * chunk;
  if (beg >= d.n) return;
  const long end = beg + chunk < d.n ? beg + chunk : d.n;
  for (long i = beg + threadIdx.x; i < end; i += blockDim.x) {
    float v;
    if (d.src == nullptr) v = 0.f;
    else if (d.dtype == MV_F32) v = ((const float*)d.src)[i];
    else if (d.dtype == MV_BF16) v = ld<bf16>((const bf16*)d.src + i);
    else v = ld<f16>((const f16*)d.src + i);
    flat[d.dst_off + i] = v;
  }
}

// ------------------------------------------------------------------------------------------- log-mel spectrogram loss
// frames: reflect-pad (n_fft-hop)/2, hop, periodic Hann; |rFFT| (power-of-two n_fft: radix-2 FFT of the frame in LDS, forward and
// adjoint; otherwise a direct O(N^2) DFT with an exact (k*n mod N) twiddle table), mel = fb @ mag, logmel = log(max(mel, clamp)); loss = weight * mean |logmel - target|.
// One workgroup per (b, frame).  Backward recomputes the frame spectrum and scatters d wave with atomics (frames overlap 4x).
__device__ __forceinline__ int reflect_idx(int i, int Tn) {
  if (i < 0) i = -i;
  if (i >= Tn) i = 2 * (Tn - 1) - i;
  return i;
}

// In-LDS radix-2 FFT of N = 2^logn complex points (decimation in time; the caller stored its input in bit-reversed order).
// sgn = -1: forward transform e^{-2 pi i kn/N}; +1: the adjoint.  Twiddles come from the shared cos/sin tables (index j*N/m),
// N/2 butterflies per stage spread over the workgroup, one barrier per stage.
__device__ __forceinline__ void lds_fft(float* xr, float* xi, const float* ct, const float* stb, int logn, float sgn) {
  const int N = 1 << logn;
  for (int s = 1; s <= logn; ++s) {
    const int half = 1 << (s - 1), tws = N >> s;
    for (int j = threadIdx.x; j < N / 2; j += blockDim.x) {
      const int pos = j & (half - 1), i0 = ((j >> (s - 1)) << s) + pos, i1 = i0 + half;
      const float wr = ct[pos * tws], wi = sgn * stb[pos * tws];
      const float ar = xr[i1], ai = xi[i1];
      const float tr = wr * ar - wi * ai, ti = wr * ai + wi * ar;
      const float br = xr[i0], bi = xi[i0];
      xr[i1] = br - tr; xi[i1] = bi - ti;
      xr[i0] = br + tr; xi[i0] = bi + ti;
    }
    __syncthreads();
  }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void mel_loss_kernel(const T* __restrict__ wave, const float* __restrict__ fb,
                                                       const float* __restrict__ target, float* __restrict__ mel_out,
                                                       float* __restrict__ loss_acc, float* __restrict__ gwave,
                                                       int Tn, int n_fft, int hop, int n_mels, int n_frames, float clampv,
                                                       float weight, long n_total, int kind, int logn) {
  extern __shared__ float sm[];
  const int nb = n_fft / 2 + 1;
  // logn > 0 (n_fft a power of two): rFFT path - re / im hold all n_fft points of the in-LDS FFT, `scr` the adjoint's imaginary part
  float* fr = sm;                 // [n_fft] windowed frame
  float* ct = fr + n_fft;         // [n_fft] cos table
  float* stb = ct + n_fft;        // [n_fft] sin table
  float* re = stb + n_fft;        // [nb], FFT: [n_fft]
  float* im = re + (logn > 0 ? n_fft : nb);
  float* scr = im + (logn > 0 ? n_fft : nb);   // FFT: [n_fft]
  float* mag = scr + (logn > 0 ? n_fft : 0);   // [nb]
  float* gml = mag + nb;          // [n_mels] dL/dmel
  float* red = gml + n_mels;      // [32]
  const int b = blockIdx.y, f = blockIdx.x, tid = threadIdx.x;
  const int padn = (n_fft - hop) / 2;
  const float w0 = 6.283185307179586f / (float)n_fft;
  for (int n = tid; n < n_fft; n += blockDim.x) {
    const float win = 0.5f - 0.5f * cosf(w0 * n);
    const int ti = reflect_idx(f * hop - padn + n, Tn);
    fr[n] = ld<T>(wave + (long)b * Tn + ti) * win;
    ct[n] = cosf(w0 * n);
    stb[n] = sinf(w0 * n);
  }
  __syncthreads();
  if (logn > 0) {
    for (int n = tid; n < n_fft; n += blockDim.x) {
      const int r = (int)(__brev((unsigned)n) >> (32 - logn));
      re[r] = fr[n]; im[r] = 0.f;
    }
    __syncthreads();
    lds_fft(re, im, ct, stb, logn, -1.f);
    for (int k = tid; k < nb; k += blockDim.x) mag[k] = sqrtf(re[k] * re[k] + im[k] * im[k] + 1e-9f);
  } else {
    for (int k = tid; k < nb; k += blockDim.x) {
      float a = 0.f, c = 0.f;
      int idx = 0;
      for (int n = 0; n < n_fft; ++n) {
        a += fr[n] * ct[idx];
        c -= fr[n] * stb[idx];
        idx += k; if (idx >= n_fft) idx -= n_fft;
      }
      re[k] = a; im[k] = c;
      mag[k] = sqrtf(a * a + c * c + 1e-9f);
    }
  }
  __syncthreads();
  float lsum = 0.f;
  for (int m = tid; m < n_mels; m += blockDim.x) {
    float a = 0.f;
    for (int k = 0; k < nb; ++k) a += fb[(long)m * nb + k] * mag[k];
    const float lm = logf(fmaxf(a, clampv));
    if (mel_out) mel_out[((long)b * n_mels + m) * n_frames + f] = lm;
    float g = 0.f;
    if (target) {
      const float d = lm - target[((long)b * n_mels + m) * n_frames + f];
      lsum += kind == 0 ? fabsf(d) : d * d;
      g = (kind == 0 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d) * (weight / (float)n_total);
    }
    gml[m] = (a > clampv) ? g / a : 0.f;       // d/d mel through log(clamp)
  }
  if (loss_acc && target) {
    lsum = block_sum(lsum, red);
    if (tid == 0) atomicAdd(loss_acc, lsum * weight / (float)n_total);
  }
  if (!BWD) return;
  __syncthreads();
  // g_mag[k] = sum_m fb[m][k] gml[m] ; g_re = g_mag re/mag ; g_im = g_mag im/mag
  for (int k = tid; k < nb; k += blockDim.x) {
    float a = 0.f;
    for (int m = 0; m < n_mels; ++m) a += fb[(long)m * nb + k] * gml[m];
    const float inv = a / mag[k];
    re[k] *= inv; im[k] *= inv;    // now g_re, g_im
  }
  __syncthreads();
  // g_frame[n] = sum_k g_re cos(2 pi k n/N) - g_im sin(2 pi k n/N) ; d wave += g_frame * win
  if (logn > 0) {
    // = Re( sum_{k < nb} (g_re + i g_im) e^{+2 pi i kn/N} ): the adjoint FFT of the half spectrum, zero above Nyquist
    for (int k = tid; k < n_fft; k += blockDim.x) {
      const int r = (int)(__brev((unsigned)k) >> (32 - logn));
      fr[r] = k < nb ? re[k] : 0.f; scr[r] = k < nb ? im[k] : 0.f;
    }
    __syncthreads();
    lds_fft(fr, scr, ct, stb, logn, 1.f);
  }
  for (int n = tid; n < n_fft; n += blockDim.x) {
    float a = 0.f;
    if (logn > 0) a = fr[n];
    else {
      int idx = 0;
      for (int k = 0; k < nb; ++k) {
        a += re[k] * ct[idx] - im[k] * stb[idx];
        idx += n; if (idx >= n_fft) idx -= n_fft;
      }
    }
    const float win = 0.5f - 0.5f * ct[n];
    const int ti = reflect_idx(f * hop - padn + n, Tn);
    atomicAdd(gwave + (long)b * Tn + ti, a * win);
  }
}

}  // namespace mv

using namespace mv;

static inline int grid_for(long n, int block = 256, int cap = 2048) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

extern "C" int mv_dropout_mask(uint8_t* mask, long n, float p, long seed_, void* stream) {
  const uint64_t seed = (uint64_t)seed_;
  MV_CHECK_ARG(mask && n > 0 && p >= 0.f && p < 1.f && ((uintptr_t)mask & 7) == 0);
  const uint32_t thresh = (uint32_t)(p * 65536.f + 0.5f);
  const long nblk = (n + 7) / 8;
  const unsigned grid = (unsigned)((nblk + 255) / 256 > 4096 ? 4096 : (nblk + 255) / 256);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mask, n, thresh, (uint32_t)seed, (uint32_t)(seed >> 32));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_grc_fold_bwd(const float* g_weff, const float* g_beff, const void* conv_w, const void* conv_b,
                               const void* lora_A, const void* lora_B, const void* lora_scaling, const void* proj_w,
                               float* g_conv_w, float* g_conv_b, float* g_A, float* g_B, float* g_s, float* g_proj_w,
                               float* g_proj_b, int Cin, int Cout, int ks, int groups, int rank, int param_dtype,
                               void* stream) {
  MV_CHECK_ARG(g_weff && g_beff && conv_w && conv_b && lora_A && lora_B && lora_scaling && proj_w);
  MV_CHECK_ARG(g_conv_w && g_conv_b && g_A && g_B && g_s && g_proj_w && g_proj_b && (ks & 1) && Cin % groups == 0 && Cout % groups == 0);
  const size_t lds = sizeof(float) * (2 * (size_t)Cout * Cin * ks + (size_t)Cin * Cout + 32 + (size_t)Cout * (Cin / groups) * ks +
                                      (size_t)Cout * Cout + (size_t)Cin * rank + (size_t)rank * Cout);
  if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
#define GO(P) { auto kern = grc_fold_bwd_kernel<P>; \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, (hipStream_t)stream, g_weff, g_beff, (const P*)conv_w, (const P*)conv_b, \
      (const P*)lora_A, (const P*)lora_B, (const P*)lora_scaling, (const P*)proj_w, g_conv_w, g_conv_b, g_A, g_B, g_s, g_proj_w, \
      g_proj_b, Cin, Cout, ks, groups, rank); }
  switch (param_dtype) {
    case MV_F32: GO(float); break;
    case MV_BF16: GO(bf16); break;
    case MV_F16: GO(f16); break;
    default: return MV_ERR_DTYPE;
  }
#undef GO
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_loss_fwd_bwd(const void* x, const void* y, float c, float weight, float* loss_acc, void* gx, void* gy,
                               long n, int kind, int dtype, void* stream) {
  MV_CHECK_ARG(x && loss_acc && n > 0 && kind >= 0 && kind <= 4 && ((kind != 1 && kind != 4) || y));
  MV_DISPATCH(dtype, hipLaunchKernelGGL(loss_kernel<T>, dim3(grid_for(n, 256, 512)), dim3(256), 0, (hipStream_t)stream,
                                        (const T*)x, (const T*)y, c, weight, loss_acc, (T*)gx, (T*)gy, n, kind));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_scale(void* x, const float* factor_dev, float factor, long n, int dtype, void* stream) {
  MV_CHECK_ARG(x && n > 0);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(scale_kernel<T>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (T*)x,
                                        factor_dev, factor, n));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_scale_to(const void* x, void* y, const float* factor_dev, float factor, long n, int dtype, void* stream) {
  MV_CHECK_ARG(x && y && n > 0);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(scale_to_kernel<T>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y,
                                        factor_dev, factor, n));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_adamw_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, float grad_scale, void* stream) {
  MV_CHECK_ARG(p && g && m && v && n > 0 && step >= 1);
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2, grad_scale);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_adamw_flat_dev(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, const int* step_dev, float grad_scale, void* stream) {
  MV_CHECK_ARG(p && g && m && v && n > 0 && step_dev);
  hipLaunchKernelGGL(adamw_dev_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, step_dev, grad_scale);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_multi_gather(const void* descs_dev, int n_tensors, long max_len, float* flat, void* stream) {
  MV_CHECK_ARG(descs_dev && flat && n_tensors > 0 && n_tensors <= 65535 && max_len > 0);
  const int chunk = 16384;
  dim3 grid((unsigned)((max_len + chunk - 1) / chunk), n_tensors);
  hipLaunchKernelGGL(multi_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const GatherDesc*)descs_dev, flat, chunk);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_mel_loss(const void* wave, const float* fb, const float* target, float* mel_out, float* loss_acc,
                           float* gwave, int B, int T_, int n_fft, int hop, int n_mels, float clampv, float weight,
                           int kind, int backward, int dtype, void* stream) {
  MV_CHECK_ARG(wave && fb && B > 0 && T_ > 0 && n_fft > 0 && hop > 0 && n_mels > 0 && T_ % hop == 0 && n_fft >= hop);
  MV_CHECK_ARG(!backward || (gwave && target));
  MV_CHECK_ARG(kind == 0 || kind == 1);
  MV_CHECK_ARG((n_fft - hop) / 2 < T_);
  const int n_frames = T_ / hop, nb = n_fft / 2 + 1;
  int logn = 0;                                       // rFFT path for power-of-two n_fft (env MV_MEL_DFT=1: the direct DFT, for comparison)
  {
    static int force_dft = -1;
    if (force_dft < 0) { const char* e = getenv("MV_MEL_DFT"); force_dft = e ? atoi(e) : 0; }
    if (!force_dft && n_fft >= 32 && (n_fft & (n_fft - 1)) == 0) while ((1 << logn) < n_fft) ++logn;
  }
  const size_t lds = sizeof(float) * ((logn > 0 ? 6 * (size_t)n_fft + nb : 3 * (size_t)n_fft + 3 * (size_t)nb) + n_mels + 32);
  if (lds > 160 * 1024 || B > 65535) return MV_ERR_UNSUPPORTED;
  const long n_total = (long)B * n_mels * n_frames;
  dim3 grid(n_frames, B);
  MV_DISPATCH(dtype, {
    if (backward) {
      auto kern = mel_loss_kernel<T, true>;
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, (const T*)wave, fb, target, mel_out,
                         loss_acc, gwave, T_, n_fft, hop, n_mels, n_frames, clampv, weight, n_total, kind, logn);
    } else {
      auto kern = mel_loss_kernel<T, false>;
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, (const T*)wave, fb, target, mel_out,
                         loss_acc, gwave, T_, n_fft, hop, n_mels, n_frames, clampv, weight, n_total, kind, logn);
    }
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}
