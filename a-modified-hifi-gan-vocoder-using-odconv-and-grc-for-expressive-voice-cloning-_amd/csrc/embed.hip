// Conditioning producers: the kernels of the ECAPA-TDNN speaker encoder and the Emotion2Vec
// emotion encoder (reference embedding_extractors.py:13-284) that are not plain convolutions.  Activations are channels-last
// [B][T][C]; every Conv1d / Linear over them runs on the MFMA implicit-GEMM kernels of disc_fused.hip (BatchNorm folded into
// the packed weights on the host side, ReLU / tanh in the conv epilogue).  Here:
//   mha_kernel            multi-head self-attention (flash form, online softmax, S^T = K Q^T and O^T = V^T P^T on MFMA: the
//                         accumulator layout of S^T IS the B-operand layout of P^T, so the probabilities never leave registers)
//   add_layernorm_kernel  y = LayerNorm(x + r) (post-norm transformer layer)
//   mean_t_kernel         mean over time -> fp32 [B][C]                     (SE squeeze, utterance pooling)
//   se_gate_kernel        sigmoid(W2 relu(W1 m + b1) + b2)                  (SE excitation, embedding_extractors.py:159-170)
//   scale_add_kernel      y = x * gate[b][c] + r                            (SE scale + block residual, :147-149)
//   res2_glue_kernel      cat[:, i] = y_i ; next input = u[:, i+1] + y_i    (Res2Net chain, :135-143)
//   asp_* kernels         softmax over CHANNELS (as the reference has it, :44), attended mean / unbiased std over time (:76-84)
//   l2norm_rows_kernel    F.normalize(p=2, dim=1)                           (:90, :245)
#include "common.h"
#include "mfma.h"

namespace mv {

template <typename T> __device__ __forceinline__ void load8(const T* p, float* o) {
  if constexpr (sizeof(T) == 4) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  } else {
    load8f<T>(p, o);
  }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* v) {
  if constexpr (sizeof(T) == 4) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    store8f<T>(p, v);
  }
}

// ------------------------------------------------------------------------------------------- y = LN(x + r) * g + b
// One wave per row; C % 8 == 0; the row (<= 8 x 8 x 64 = 4096 channels) stays in registers between the two passes.
template <typename T, int NV>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const T* __restrict__ x, const T* __restrict__ r, const float* __restrict__ gam,
                                                            const float* __restrict__ bet, T* __restrict__ y, long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[NV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < C) {
      load8<T>(x + row * C + c, v[i]);
      if (r) { float t[8]; load8<T>(r + row * C + c, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] += t[e]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c < C) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rstd * gam[c + e] + bet[c + e];
      store8<T>(y + row * C + c, o);
    }
  }
}

// ------------------------------------------------------------------------------------------- mean over time
// x [B][T][C] -> out fp32 [B][C].  grid (C/64, B, TS): thread = (channel, time slice); TS slices combined by atomics when > 1.
template <typename T>
__global__ __launch_bounds__(256) void mean_t_kernel(const T* __restrict__ x, float* __restrict__ out, int T_, int C, float inv_t) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sl = threadIdx.x >> 6;                       // 4 time slices per block
  const int b = blockIdx.y;
  __shared__ float red[4][64];
  float s = 0.f;
  if (c < C)
    for (int t = sl; t < T_; t += 4) s += ld<T>(x + ((long)b * T_ + t) * C + c);
  red[sl][threadIdx.x & 63] = s;
  __syncthreads();
  if (sl == 0 && c < C) out[(long)b * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) * inv_t;
}

// ------------------------------------------------------------------------------------------- SE excitation
// gate[b][c] = sigmoid(W2[c][:] . relu(W1 m_b + b1) + b2[c]);  W1 [R][C], W2 [C][R] fp32 masters; one block per sample.
__global__ __launch_bounds__(1024) void se_gate_kernel(const float* __restrict__ m, const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ gate,
                                                       int C, int R) {
  extern __shared__ float sm[];                          // m_b [C] | h [R]
  float* ml = sm; float* hl = sm + C;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < C; c += 1024) ml[c] = m[(long)b * C + c];
  __syncthreads();
  for (int r = wid; r < R; r += 16) {                    // one wave per hidden unit, 16 units in flight
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += w1[(long)r * C + c] * ml[c];
    s = wave_sum(s);
    if (lane == 0) hl[r] = fmaxf(s + b1[r], 0.f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 1024) {
    float s = b2[c];
    const float* wr = w2 + (long)c * R;
    if ((R & 3) == 0) {
      for (int r = 0; r < R; r += 4) { const float4 w = *reinterpret_cast<const float4*>(wr + r); s += w.x * hl[r] + w.y * hl[r + 1] + w.z * hl[r + 2] + w.w * hl[r + 3]; }
    } else {
      for (int r = 0; r < R; ++r) s += wr[r] * hl[r];
    }
    gate[(long)b * C + c] = 1.f / (1.f + __expf(-s));
  }
}

// ------------------------------------------------------------------------------------------- y = x * gate[b][c] + r
template <typename T>
__global__ __launch_bounds__(256) void scale_add_kernel(const T* __restrict__ x, const float* __restrict__ gate, const T* __restrict__ r,
                                                        T* __restrict__ y, long n8, int T_, int C) {
  const int c8 = C / 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c8) * 8;
    const long b = i / ((long)c8 * T_);
    float v[8], rr[8];
    load8<T>(x + i * 8, v);
    load8<T>(r + i * 8, rr);
    const float* g = gate + b * C + c;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * g[e] + rr[e];
    store8<T>(y + i * 8, v);
  }
}

// ------------------------------------------------------------------------------------------- Res2Net chain glue
// cat[row][dst_off .. +cs) = src[row][0..cs)  (src = u slice 0 at src_stride C for the first step, else the conv output);
// nxt[row][0..cs) = u[row][nxt_off .. +cs) + src[row][..]   when nxt != nullptr.   cs % 8 == 0.
template <typename T>
__global__ __launch_bounds__(256) void res2_glue_kernel(const T* __restrict__ src, int src_stride, const T* __restrict__ u, T* __restrict__ cat,
                                                        T* __restrict__ nxt, long rows, int C, int cs, int dst_off, int nxt_off) {
  const int p8 = cs / 8;
  const long n = rows * p8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long row = i / p8;
    const int c = (int)(i % p8) * 8;
    float v[8];
    load8<T>(src + row * src_stride + c, v);
    store8<T>(cat + row * C + dst_off + c, v);
    if (nxt) {
      float w[8];
      load8<T>(u + row * C + nxt_off + c, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] += v[e];
      store8<T>(nxt + row * cs + c, w);
    }
  }
}

// ------------------------------------------------------------------------------------------- attentive statistics pooling
// Phase 1: per (b,t) row, softmax statistics of the logits over the C channels: stat[row] = (max, 1/sum exp).
template <typename T>
__global__ __launch_bounds__(256) void asp_rowstat_kernel(const T* __restrict__ lg, float2* __restrict__ stat, long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float m = -3.0e38f;
  for (int c = lane * 8; c < C; c += 512) { float v[8]; load8<T>(lg + row * C + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, v[e]); }
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane * 8; c < C; c += 512) { float v[8]; load8<T>(lg + row * C + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += __expf(v[e] - m); }
  s = wave_sum(s);
  if (lane == 0) stat[row] = make_float2(m, 1.f / s);
}
// Phase 2: per (b,c): a_t = x * softmax weight; pooled[b][c] = mean_t a, pooled[b][C + c] = unbiased std_t a (two passes over t).
template <typename T>
__global__ __launch_bounds__(256) void asp_pool_kernel(const T* __restrict__ x, const T* __restrict__ lg, const float2* __restrict__ stat,
                                                       float* __restrict__ pooled, int T_, int C) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6, b = blockIdx.y;
  __shared__ float red[4][64];
  const bool ok = c < C;
  float s = 0.f;
  if (ok)
    for (int t = sl; t < T_; t += 4) {
      const long row = (long)b * T_ + t;
      const float2 st_ = stat[row];
      s += ld<T>(x + row * C + c) * __expf(ld<T>(lg + row * C + c) - st_.x) * st_.y;
    }
  red[sl][threadIdx.x & 63] = s;
  __syncthreads();
  const float mean = (red[0][threadIdx.x & 63] + red[1][threadIdx.x & 63] + red[2][threadIdx.x & 63] + red[3][threadIdx.x & 63]) / (float)T_;
  __syncthreads();
  float q = 0.f;
  if (ok)
    for (int t = sl; t < T_; t += 4) {
      const long row = (long)b * T_ + t;
      const float2 st_ = stat[row];
      const float d = ld<T>(x + row * C + c) * __expf(ld<T>(lg + row * C + c) - st_.x) * st_.y - mean;
      q += d * d;
    }
  red[sl][threadIdx.x & 63] = q;
  __syncthreads();
  if (sl == 0 && ok) {
    const float var = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)(T_ - 1);
    pooled[(long)b * 2 * C + c] = mean;
    pooled[(long)b * 2 * C + C + c] = sqrtf(var);
  }
}

// ------------------------------------------------------------------------------------------- rows / max(||row||, eps)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, T* __restrict__ y, int C, float eps) {
  __shared__ float red[16];
  const long b = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) { const float v = x[b * C + c]; s += v * v; }
  s = block_sum(s, red);
  const float inv = 1.f / fmaxf(sqrtf(s), eps);
  for (int c = threadIdx.x; c < C; c += 256) st<T>(y + b * C + c, x[b * C + c] * inv);
}

// ------------------------------------------------------------------------------------------- multi-head self-attention
// qkv [B][T][3*H] (q | k | v, head h at h*HD inside each, nn.MultiheadAttention's in_proj order) -> out [B][T][H].
// grid (ceil(T/64), heads, B), 4 waves x 16 queries.  Keys stream through LDS in blocks of 64: K row-major [key][HDP] and V
// transposed [d][key].  Per wave and 32 keys: S^T = K Q^T (rows = keys, cols = queries) -> online softmax per column (= per lane,
// max across the 4 lane groups by two xor-shuffles) -> P^T stays in the accumulator registers and is the B operand of
// O^T += V^T P^T (from_acc: contraction slot j of lane group g <-> key 4g+j of the first 16 keys, 16+4g+(j-4) of the second;
// the V^T fragment is read with the same permutation, two 8-byte LDS reads).
template <typename T, int HD>
__global__ __launch_bounds__(256) void mha_kernel(const T* __restrict__ qkv, T* __restrict__ out, int T_, int H, float scale_log2e) {
  using M = Mma<T>;
  using V = typename M::V;
  constexpr int HDP = HD < 32 ? 32 : HD;                 // contraction width of S (zero-padded for HD 16)
  constexpr int KS = HDP / 32;                           // k-steps of S
  constexpr int DT = HD / 16;                            // 16-row tiles of O^T
  constexpr int KB = 64;                                 // keys per LDS block
  constexpr int KRS = HDP * 2 + 16;                      // K row stride (bytes)
  constexpr int VRS = KB * 2 + 8;                        // V^T row stride (bytes)
  __shared__ __attribute__((aligned(16))) char kl[KB * KRS];
  __shared__ __attribute__((aligned(16))) char vl[HD * VRS];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const long rs = 3L * H;                                // qkv row stride (elements)
  const T* base = qkv + (long)b * T_ * rs + (long)h * HD;
  const int q0 = blockIdx.x * 64 + wid * 16;
  // Q^T operand: B[k = d][col = query l15]
  V qf[KS];
  {
    int qi = q0 + l15; if (qi >= T_) qi = T_ - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d = ks * 32 + g * 8;
      if (d < HD) qf[ks] = M::load_b(base + (long)qi * rs + d);
      else { u32x4 z = {0u, 0u, 0u, 0u}; qf[ks].v = __builtin_bit_cast(decltype(qf[ks].v), z); }
    }
  }
  f32x4 acc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrun = -3.0e38f, lrun = 0.f;

  // K / V pieces of a key block travel global -> registers one block ahead of their use (unconditional, row-clamped loads: the
  // fetch of block kb+1 is in flight under the MFMAs of block kb), registers -> LDS at the top of the block
  constexpr int PCS = HD / 8;                            // 16-byte pieces per key row
  constexpr int NPT = (KB * PCS + 255) / 256;            // pieces per thread
  uint4 kreg[NPT], vreg[NPT];
  auto fetch = [&](int kb) {
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int i = tid + j * 256;
      const int key = (i / PCS) % KB, pc = i % PCS;
      int kr = kb + key; if (kr >= T_) kr = T_ - 1;
      const T* rowp = base + (long)kr * rs + pc * 8;
      kreg[j] = *reinterpret_cast<const uint4*>(rowp + H);
      vreg[j] = *reinterpret_cast<const uint4*>(rowp + 2 * H);
    }
  };
  fetch(0);
  for (int kb = 0; kb < T_; kb += KB) {
    __syncthreads();
    // stage K (row-major, zero rows past T, zero pad columns) and V^T
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int i = tid + j * 256;
      if (i < KB * PCS) {
        const int key = i / PCS, pc = i % PCS;
        const bool ok = kb + key < T_;
        const uint4 kv = ok ? kreg[j] : make_uint4(0, 0, 0, 0), vv = ok ? vreg[j] : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(kl + key * KRS + pc * 16) = kv;
        if (HD < HDP && pc == 0) {
#pragma unroll
          for (int z = HD / 8; z < HDP / 8; ++z) *reinterpret_cast<uint4*>(kl + key * KRS + z * 16) = make_uint4(0, 0, 0, 0);
        }
        const uint16_t* ve = reinterpret_cast<const uint16_t*>(&vv);
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<uint16_t*>(vl + (pc * 8 + e) * VRS + key * 2) = ve[e];
      }
    }
    fetch(kb + KB < T_ ? kb + KB : kb);
    __syncthreads();
#pragma unroll
    for (int sb = 0; sb < KB; sb += 32) {
      if (kb + sb >= T_) break;
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const V ka = M::load_b(kl + (sb + l15) * KRS + (ks * 32 + g * 8) * 2);
        const V kb2 = M::load_b(kl + (sb + 16 + l15) * KRS + (ks * 32 + g * 8) * 2);
        s0 = M::mma(ka, qf[ks], s0);
        s1 = M::mma(kb2, qf[ks], s1);
      }
      // lane: query l15, keys kb+sb+4g+r (s0) and kb+sb+16+4g+r (s1)
      float mx = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = (kb + sb + 4 * g + r < T_) ? s0[r] * scale_log2e : -3.0e38f;
        s1[r] = (kb + sb + 16 + 4 * g + r < T_) ? s1[r] * scale_log2e : -3.0e38f;
        mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun, mx);
      const float corr = exp2f(mrun - mnew);
      mrun = mnew;
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = exp2f(s0[r] - mnew); s1[r] = exp2f(s1[r] - mnew);
        ps += s0[r] + s1[r];
      }
      lrun = lrun * corr + ps;
      const V pf = M::from_acc(s0, s1);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const char* vp = vl + (dt * 16 + l15) * VRS + (sb + 4 * g) * 2;
        const u32x2 lo = *reinterpret_cast<const u32x2*>(vp), hi = *reinterpret_cast<const u32x2*>(vp + 32);
        u32x4 u = {lo[0], lo[1], hi[0], hi[1]};
        V va; va.v = __builtin_bit_cast(decltype(va.v), u);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[dt][r] *= corr;
        acc[dt] = M::mma(va, pf, acc[dt]);
      }
    }
  }
  float l = lrun;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
  const int qi = q0 + l15;
  if (qi < T_) {
    T* op = out + ((long)b * T_ + qi) * H + (long)h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      float o[4] = {acc[dt][0] * inv, acc[dt][1] * inv, acc[dt][2] * inv, acc[dt][3] * inv};
      M::store4(op + dt * 16 + 4 * g, o);
    }
  }
}

// fp32-storage (parity-grade) and odd head sizes: one thread per (b, h, query), two passes over the keys.
template <typename T>
__global__ __launch_bounds__(64) void mha_naive_kernel(const T* __restrict__ qkv, T* __restrict__ out, int T_, int H, int HD, float scale) {
  const int qi = blockIdx.x * 64 + threadIdx.x, h = blockIdx.y, b = blockIdx.z;
  if (qi >= T_) return;
  const long rs = 3L * H;
  const T* base = qkv + (long)b * T_ * rs + (long)h * HD;
  const T* qp = base + (long)qi * rs;
  float m = -3.0e38f;
  for (int k = 0; k < T_; ++k) {
    float s = 0.f;
    for (int d = 0; d < HD; ++d) s += ld<T>(qp + d) * ld<T>(base + (long)k * rs + H + d);
    m = fmaxf(m, s * scale);
  }
  float l = 0.f;
  float o[128];
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  for (int k = 0; k < T_; ++k) {
    float s = 0.f;
    for (int d = 0; d < HD; ++d) s += ld<T>(qp + d) * ld<T>(base + (long)k * rs + H + d);
    const float p = __expf(s * scale - m);
    l += p;
    for (int d = 0; d < HD; ++d) o[d] += p * ld<T>(base + (long)k * rs + 2 * H + d);
  }
  T* op = out + ((long)b * T_ + qi) * H + (long)h * HD;
  for (int d = 0; d < HD; ++d) st<T>(op + d, o[d] / l);
}


// ------------------------------------------------------------------------------------------- skinny 1x1 GEMM (few positions)
// y[M][N] = act(x[M][K] W^T + bias) for the conditioning producers' short sequences (M = B*T ~ 1e3 rows): the tiled conv kernels
// of disc_fused.hip are latency-bound there (one LDS round trip per 128-channel chunk, <= 1 workgroup per CU).  Here a workgroup
// owns 32 positions x 64 output channels and its 8 waves SPLIT K: wave w takes k-steps [w*PER, (w+1)*PER), issues ALL its operand
// loads up front (weights from the packed A-fragment image of mv_dconv_pack, activations straight from global as B operands - no
// LDS staging, no barrier before the MFMAs), so a launch costs about one memory latency.  Partial tiles are summed through LDS,
// wave f finishing fragment f (bias + activation + 8-byte channels-last stores).
template <typename T, int PER, typename ActF>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const T* __restrict__ x, const T* __restrict__ wp, const T* __restrict__ bias,
                                                          T* __restrict__ y, int Mrows, int K, int Nc, ActF actf) {
  using M = Mma<T>;
  using V = typename M::V;
  constexpr int MT = 4, NP = 2;
  __shared__ __attribute__((aligned(16))) float red[8][MT * NP][64][4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  const int p0 = blockIdx.x * (16 * NP), mt0 = blockIdx.y * MT;
  const int ksteps = K / 32;
  V a[PER][MT], b[PER][NP];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ks = wid * PER + i;
#pragma unroll
    for (int m = 0; m < MT; ++m) a[i][m] = M::load_b(wp + (((long)(mt0 + m) * ksteps + ks) * 64 + lane) * 8);
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      int pos = p0 + 16 * n + l15; if (pos >= Mrows) pos = Mrows - 1;
      b[i][n] = M::load_b(x + (long)pos * K + ks * 32 + g * 8);
    }
  }
  f32x4 acc[MT][NP];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NP; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < PER; ++i)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NP; ++n) acc[m][n] = M::mma(a[i][m], b[i][n], acc[m][n]);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NP; ++n) *reinterpret_cast<f32x4*>(&red[wid][m * NP + n][lane][0]) = acc[m][n];
  __syncthreads();
  f32x4 sum = *reinterpret_cast<const f32x4*>(&red[0][wid][lane][0]);
#pragma unroll
  for (int w = 1; w < 8; ++w) { const f32x4 t = *reinterpret_cast<const f32x4*>(&red[w][wid][lane][0]); sum += t; }
  const int m = wid / NP, n = wid % NP;
  const int pos = p0 + 16 * n + l15, co = (mt0 + m) * 16 + 4 * g;
  if (pos < Mrows) {
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) M::load4(bias + co, bv);
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = actf(sum[r] + bv[r]);
    M::store4(y + (long)pos * Nc + co, o);
  }
}


// ------------------------------------------------------------------------------------------- fused Res2Net chain
// The 7 dependent k=3 dilated convs of SE_Res2Block (embedding_extractors.py:135-143) in ONE launch instead of 7 convs + 8 glue
// launches: ys[0] = xs[0]; ys[i] = conv_i(xs[i] + ys[i-1]); cat = concat(ys).  A workgroup owns 64 output positions of one sample and
// recomputes the halo: step i is evaluated on [t0 - (7-i)d, t0 + 64 + (7-i)d), its input xs[i] + ys[i-1] lives in LDS (two
// ping-pong tiles, channels-last rows), output channels on MFMA rows so a lane owns 4 consecutive channels of one position
// (8-byte stores into the next tile and into cat).  Weights: the 7 packed A-fragment images of mv_dconv_pack back to back.
template <typename T, int CS>
__global__ __launch_bounds__(256) void res2_chain_kernel(const T* __restrict__ u, const T* __restrict__ wp, const T* __restrict__ bias,
                                                         T* __restrict__ cat, int T_, int C, int d) {
  using M = Mma<T>;
  using V = typename M::V;
  constexpr int TT = 64, MT = CS / 16, KH = CS / 32, KS = 3 * KH, RS = CS * 2 + 16, PCS = CS / 8;
  extern __shared__ __align__(16) char lds[];
  const int P = TT + 14 * d + 16;
  char* cur = lds;
  char* nxt = lds + P * RS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * TT, base = t0 - 7 * d;
  const T* ub = u + (long)b * T_ * C;
  T* cb = cat + (long)b * T_ * C;
  for (int i = tid; i < P * PCS; i += 256) {           // in_1 = xs[1] + xs[0] (zero outside the sequence); cat[:, 0] = xs[0]
    const int row = i / PCS, pc = i % PCS, p = base + row;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p >= 0 && p < T_) {
      float a[8], c[8];
      load8<T>(ub + (long)p * C + pc * 8, a);
      load8<T>(ub + (long)p * C + CS + pc * 8, c);
      if (p >= t0 && p < t0 + TT) store8<T>(cb + (long)p * C + pc * 8, a);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = a[e] + c[e];
    }
    store8<T>(reinterpret_cast<T*>(cur + row * RS + pc * 16), v);
  }
#pragma unroll 1
  for (int i = 1; i <= 7; ++i) {
    V a[MT][KS];
    const T* wi = wp + (long)(i - 1) * CS * CS * 3;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[mt][ks] = M::load_b(wi + ((long)(mt * KS + ks) * 64 + lane) * 8);
    float bv[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) M::load4(bias + (i - 1) * CS + 16 * mt + 4 * g, bv[mt]);
    __syncthreads();                                    // in_i complete (staging / previous step), previous reads of `nxt` done
    const int r0 = i * d, L = TT + 2 * (7 - i) * d, nct = (L + 15) / 16;
    for (int ct = wid; ct < nct; ct += 4) {
      const int row = r0 + 16 * ct + l15, p = base + row, pt0 = base + r0 + 16 * ct;
      const bool live = pt0 < T_ && pt0 + 16 > 0;       // wave-uniform: tiles wholly outside the sequence are zeros
      f32x4 acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live) {
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
          for (int kh = 0; kh < KH; ++kh) {
            const V bf = M::load_b(cur + (row + (tap - 1) * d) * RS + (kh * 32 + g * 8) * 2);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = M::mma(a[mt][tap * KH + kh], bf, acc[mt]);
          }
      }
      const bool inreg = 16 * ct + l15 < L;
      const bool valid = inreg && p >= 0 && p < T_;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = valid ? acc[mt][r] + bv[mt][r] : 0.f;
        const int co = 16 * mt + 4 * g;
        if (valid && p >= t0 && p < t0 + TT) M::store4(cb + (long)p * C + i * CS + co, o);
        if (i < 7 && inreg) {
          float nx[4] = {0.f, 0.f, 0.f, 0.f};
          if (valid) {
            float uu[4];
            M::load4(ub + (long)p * C + (i + 1) * CS + co, uu);
#pragma unroll
            for (int r = 0; r < 4; ++r) nx[r] = uu[r] + o[r];
          }
          M::store4(nxt + row * RS + co * 2, nx);
        }
      }
    }
    char* t = cur; cur = nxt; nxt = t;
  }
}

}  // namespace mv

using namespace mv;

static inline unsigned grid_for(long n, int per_block, unsigned cap = 65535u * 16u) {
  long g = (n + per_block - 1) / per_block;
  return (unsigned)(g < 1 ? 1 : (g > (long)cap ? cap : g));
}

extern "C" int mv_add_layernorm(const void* x, const void* res, const float* gamma, const float* beta, void* y, long rows, int C, float eps,
                                int dtype, void* stream) {
  MV_CHECK_ARG(x && gamma && beta && y && rows > 0 && C > 0 && C % 8 == 0 && C <= 4096);
  MV_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0);
  const dim3 grid((unsigned)((rows + 3) / 4));
#define MV_LN(NV_) MV_DISPATCH(dtype, hipLaunchKernelGGL((add_layernorm_kernel<T, NV_>), grid, dim3(256), 0, (hipStream_t)stream, \
                               (const T*)x, (const T*)res, gamma, beta, (T*)y, rows, C, eps))
  if (C <= 512) { MV_LN(1); } else if (C <= 1024) { MV_LN(2); } else if (C <= 2048) { MV_LN(4); } else { MV_LN(8); }
#undef MV_LN
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_mean_t_cl(const void* x, float* out, int B, int T_, int C, int dtype, void* stream) {
  MV_CHECK_ARG(x && out && B > 0 && T_ > 0 && C > 0 && B <= 65535);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(mean_t_kernel<T>, dim3(cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream, (const T*)x, out, T_, C,
                                        1.f / (float)T_));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_se_gate(const float* mean, const float* w1, const float* b1, const float* w2, const float* b2, float* gate, int B, int C,
                          int R, void* stream) {
  MV_CHECK_ARG(mean && w1 && b1 && w2 && b2 && gate && B > 0 && C > 0 && R > 0 && (size_t)(C + R) * 4 <= 64 * 1024);
  hipLaunchKernelGGL(se_gate_kernel, dim3(B), dim3(1024), (size_t)(C + R) * 4, (hipStream_t)stream, mean, w1, b1, w2, b2, gate, C, R);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_scale_add_cl(const void* x, const float* gate, const void* res, void* y, int B, int T_, int C, int dtype, void* stream) {
  MV_CHECK_ARG(x && gate && res && y && B > 0 && T_ > 0 && C > 0 && C % 8 == 0);
  MV_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0);
  const long n8 = (long)B * T_ * (C / 8);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(scale_add_kernel<T>, dim3(grid_for(n8, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)x, gate,
                                        (const T*)res, (T*)y, n8, T_, C));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_res2_glue(const void* src, int src_stride, const void* u, void* cat, void* nxt, long rows, int C, int cs, int dst_off,
                            int nxt_off, int dtype, void* stream) {
  MV_CHECK_ARG(src && cat && rows > 0 && C > 0 && cs > 0 && cs % 8 == 0 && C % 8 == 0 && src_stride % 8 == 0 && dst_off % 8 == 0);
  MV_CHECK_ARG(dst_off >= 0 && dst_off + cs <= C && (!nxt || (u && nxt_off % 8 == 0 && nxt_off >= 0 && nxt_off + cs <= C)));
  MV_CHECK_ARG((((uintptr_t)src | (uintptr_t)u | (uintptr_t)cat | (uintptr_t)nxt) & 15) == 0);
  const long n = rows * (cs / 8);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(res2_glue_kernel<T>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const T*)src,
                                        src_stride, (const T*)u, (T*)cat, (T*)nxt, rows, C, cs, dst_off, nxt_off));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" size_t mv_asp_workspace_bytes(int B, int T_) { return (size_t)B * T_ * sizeof(float2); }

extern "C" int mv_asp_pool(const void* x, const void* logits, void* workspace, float* pooled, int B, int T_, int C, int dtype, void* stream) {
  MV_CHECK_ARG(x && logits && workspace && pooled && B > 0 && T_ > 1 && C > 0 && C % 8 == 0 && B <= 65535);
  MV_CHECK_ARG((((uintptr_t)x | (uintptr_t)logits | (uintptr_t)workspace) & 15) == 0);
  const long rows = (long)B * T_;
  MV_DISPATCH(dtype, hipLaunchKernelGGL(asp_rowstat_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                                        (const T*)logits, (float2*)workspace, rows, C));
  MV_LAUNCH_CHECK();
  MV_DISPATCH(dtype, hipLaunchKernelGGL(asp_pool_kernel<T>, dim3(cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                        (const T*)logits, (const float2*)workspace, pooled, T_, C));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_l2norm_rows(const float* x, void* y, int B, int C, float eps, int dtype, void* stream) {
  MV_CHECK_ARG(x && y && B > 0 && C > 0);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(l2norm_rows_kernel<T>, dim3(B), dim3(256), 0, (hipStream_t)stream, x, (T*)y, C, eps));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_mha_fwd(const void* qkv, void* out, int B, int T_, int nheads, int head_dim, int dtype, void* stream) {
  MV_CHECK_ARG(qkv && out && B > 0 && T_ > 0 && nheads > 0 && head_dim > 0 && head_dim <= 128 && B <= 65535 && nheads <= 65535);
  const int H = nheads * head_dim;
  const float scale = 1.f / sqrtf((float)head_dim);
  hipStream_t st_ = (hipStream_t)stream;
  const bool mfma_ok = dtype != MV_F32 && (head_dim == 16 || head_dim == 32 || head_dim == 64) &&
                       (((uintptr_t)qkv | (uintptr_t)out) & 15) == 0;
  if (mfma_ok) {
    const dim3 grid(cdiv(T_, 64), nheads, B);
    const float sl2 = scale * 1.44269504088896f;
#define MV_MHA(TT, HD_) hipLaunchKernelGGL((mha_kernel<TT, HD_>), grid, dim3(256), 0, st_, (const TT*)qkv, (TT*)out, T_, H, sl2)
#define MV_MHA_T(TT) do { if (head_dim == 64) MV_MHA(TT, 64); else if (head_dim == 32) MV_MHA(TT, 32); else MV_MHA(TT, 16); } while (0)
    if (dtype == MV_BF16) MV_MHA_T(bf16); else if (dtype == MV_F16) MV_MHA_T(f16); else return MV_ERR_DTYPE;
#undef MV_MHA_T
#undef MV_MHA
  } else {
    const dim3 grid(cdiv(T_, 64), nheads, B);
    MV_DISPATCH(dtype, hipLaunchKernelGGL(mha_naive_kernel<T>, grid, dim3(64), 0, st_, (const T*)qkv, (T*)out, T_, H, head_dim, scale));
  }
  MV_LAUNCH_CHECK();
  return MV_OK;
}

template <typename T, typename ActF>
static int skinny_launch(const void* x, const void* wp, const void* bias, void* y, int M, int K, int N, ActF actf, hipStream_t st_) {
  const dim3 grid(cdiv(M, 32), N / 64);
#define MV_SK(PER_) hipLaunchKernelGGL((gemm_skinny_kernel<T, PER_, ActF>), grid, dim3(512), 0, st_, (const T*)x, (const T*)wp, (const T*)bias, (T*)y, M, K, N, actf)
  switch (K / 256) {
    case 1: MV_SK(1); break;
    case 2: MV_SK(2); break;
    case 3: MV_SK(3); break;
    case 4: MV_SK(4); break;
    case 6: MV_SK(6); break;
    case 8: MV_SK(8); break;
    default: return MV_ERR_UNSUPPORTED;
  }
#undef MV_SK
  return MV_OK;
}

extern "C" int mv_gemm_cl_skinny(const void* x, const void* packed, const void* bias, void* y, int M, int K, int N, int act, float slope,
                                 int dtype, void* stream) {
  MV_CHECK_ARG(x && packed && y && M > 0 && K > 0 && N > 0);
  MV_CHECK_ARG((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)y | (uintptr_t)bias) & 15) == 0);
  if (dtype == MV_F32 || K % 256 != 0 || N % 64 != 0 || N / 64 > 65535) return MV_ERR_UNSUPPORTED;
  hipStream_t st_ = (hipStream_t)stream;
  int rc;
  if (act == ACT_TANH || act == ACT_SILU) {
    const ActAny f{act, slope};
    rc = dtype == MV_BF16 ? skinny_launch<bf16>(x, packed, bias, y, M, K, N, f, st_) : skinny_launch<f16>(x, packed, bias, y, M, K, N, f, st_);
  } else {
    const ActLrelu f{act == ACT_NONE ? 1.f : slope};
    rc = dtype == MV_BF16 ? skinny_launch<bf16>(x, packed, bias, y, M, K, N, f, st_) : skinny_launch<f16>(x, packed, bias, y, M, K, N, f, st_);
  }
  if (rc != MV_OK) return rc;
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_res2_chain(const void* u, const void* packed, const void* bias, void* cat, int B, int T_, int C, int cs, int dil, int dtype,
                             void* stream) {
  MV_CHECK_ARG(u && packed && bias && cat && B > 0 && T_ > 0 && C > 0 && cs > 0 && dil >= 1 && B <= 65535);
  MV_CHECK_ARG((((uintptr_t)u | (uintptr_t)packed | (uintptr_t)bias | (uintptr_t)cat) & 15) == 0);
  if (dtype == MV_F32 || C != 8 * cs || (cs != 32 && cs != 64) || dil > 4) return MV_ERR_UNSUPPORTED;
  const dim3 grid(cdiv(T_, 64), B);
  const size_t ldsb = 2 * (size_t)(64 + 14 * dil + 16) * (cs * 2 + 16);
  hipStream_t st_ = (hipStream_t)stream;
#define MV_R2(TT, CS_) hipLaunchKernelGGL((res2_chain_kernel<TT, CS_>), grid, dim3(256), ldsb, st_, (const TT*)u, (const TT*)packed, (const TT*)bias, (TT*)cat, T_, C, dil)
  if (dtype == MV_BF16) { if (cs == 64) MV_R2(bf16, 64); else MV_R2(bf16, 32); }
  else if (dtype == MV_F16) { if (cs == 64) MV_R2(f16, 64); else MV_R2(f16, 32); }
  else return MV_ERR_DTYPE;
#undef MV_R2
  MV_LAUNCH_CHECK();
  return MV_OK;
}
