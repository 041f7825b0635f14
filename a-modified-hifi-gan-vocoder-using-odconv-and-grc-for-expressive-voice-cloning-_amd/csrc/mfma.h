// MFMA operand/accumulator helpers for gfx950 (v_mfma_f32_16x16x32_{bf16,f16}).
//
// Fragment conventions used by every fused kernel (MI355X guide §3):
//   A[row = lane&15][k = 8*(lane>>4) + j]   B[k = 8*(lane>>4) + j][col = lane&15]   j = 0..7
//   D[row = 4*(lane>>4) + r][col = lane&15]                                       r = 0..3
// We always put OUTPUT CHANNELS on D rows (A = weights) and TIME on D columns (B = activations from a
// channels-last tile), so one lane owns 4 consecutive output channels of one time step: 8/16-byte
// channels-last stores, and GroupNorm groups of 4 channels live inside one lane.
//
// Storage type T -> operand type:
//   bf16 -> bf16 MFMA;  f16 -> f16 MFMA;
//   float -> "bf16x3": every operand is split hi = bf16(x), lo = bf16(x - hi) and a product is
//            hi*hi + hi*lo + lo*hi (fp32 accumulate), ~2^-16 relative - the fp32-grade path.
#pragma once
#include "common.h"

namespace mv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  // RNE via the hardware convert (v_cvt_pk_bf16_f32)
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  f32x2_t f = {a, b};
  bf16x2_t h = __builtin_convertvector(f, bf16x2_t);
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
  f16x2_t h = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

template <typename T> struct Mma;

template <> struct Mma<bf16> {
  static constexpr int NSETS = 1;          // operand images per weight fragment
  static constexpr int ES = 2;             // storage element size
  using ST = bf16;                         // storage type in HBM
  struct V { bf16x8_t v; };
  using VA = V; using VB = V;              // weight (A) / activation (B) operand types
  static __device__ __forceinline__ f32x4 mma(const V& a, const V& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
  }
  // 8 consecutive storage elements at p (16-byte aligned, LDS or global) -> operand
  static __device__ __forceinline__ V load_b(const void* p) {
    V r; r.v = *reinterpret_cast<const bf16x8_t*>(p); return r;
  }
  // packed weight fragment: base of this lane's 16 bytes in image 0 (images are `img_stride` bytes apart)
  static __device__ __forceinline__ V load_a(const char* p, int) { return load_b(p); }
  // operand from a pre-split LDS tile (16-bit storage: the tile IS the operand; `plane` unused)
  static __device__ __forceinline__ V load_bp(const void* p, int) { return load_b(p); }
  static __device__ __forceinline__ V from_acc(const f32x4& x, const f32x4& y) {
    u32x4 u = {pack_bf16(x[0], x[1]), pack_bf16(x[2], x[3]), pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3])};
    V r; r.v = __builtin_bit_cast(bf16x8_t, u); return r;
  }
  static __device__ __forceinline__ void load4(const void* p, float* o) {   // 4 consecutive storage elements
    const u32x2 u = *reinterpret_cast<const u32x2*>(p);
    o[0] = bf16_lo(u[0]); o[1] = bf16_hi(u[0]); o[2] = bf16_lo(u[1]); o[3] = bf16_hi(u[1]);
  }
  static __device__ __forceinline__ void store4(void* p, const float* v) {
    u32x2 u = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    *reinterpret_cast<u32x2*>(p) = u;
  }
  static __device__ __forceinline__ float round_store(float v) { return rnd<bf16>(v); }
};

template <> struct Mma<f16> {
  static constexpr int NSETS = 1;
  static constexpr int ES = 2;
  using ST = f16;
  struct V { f16x8_t v; };
  using VA = V; using VB = V;
  static __device__ __forceinline__ f32x4 mma(const V& a, const V& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c, 0, 0, 0);
  }
  static __device__ __forceinline__ V load_b(const void* p) {
    V r; r.v = *reinterpret_cast<const f16x8_t*>(p); return r;
  }
  static __device__ __forceinline__ V load_a(const char* p, int) { return load_b(p); }
  // operand from a pre-split LDS tile (16-bit storage: the tile IS the operand; `plane` unused)
  static __device__ __forceinline__ V load_bp(const void* p, int) { return load_b(p); }
  static __device__ __forceinline__ V from_acc(const f32x4& x, const f32x4& y) {
    u32x4 u = {pack_f16(x[0], x[1]), pack_f16(x[2], x[3]), pack_f16(y[0], y[1]), pack_f16(y[2], y[3])};
    V r; r.v = __builtin_bit_cast(f16x8_t, u); return r;
  }
  static __device__ __forceinline__ void load4(const void* p, float* o) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    const f16x4_t h = *reinterpret_cast<const f16x4_t*>(p);
    o[0] = (float)h[0]; o[1] = (float)h[1]; o[2] = (float)h[2]; o[3] = (float)h[3];
  }
  static __device__ __forceinline__ void store4(void* p, const float* v) {
    u32x2 u = {pack_f16(v[0], v[1]), pack_f16(v[2], v[3])};
    *reinterpret_cast<u32x2*>(p) = u;
  }
  static __device__ __forceinline__ float round_store(float v) { return rnd<f16>(v); }
};

template <> struct Mma<float> {
  static constexpr int NSETS = 2;          // hi image + lo image
  static constexpr int ES = 4;
  using ST = float;
  struct V { bf16x8_t hi, lo; };
  using VA = V; using VB = V;
  static __device__ __forceinline__ void load4p(const void* p, float* o) { Mma<bf16>::load4(p, o); }   // 4 elements of one split plane
  static __device__ __forceinline__ f32x4 mma(const V& a, const V& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, c, 0, 0, 0);
  }
  static __device__ __forceinline__ V split(const float* f) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h[i] = pack_bf16(f[2 * i], f[2 * i + 1]);
      l[i] = pack_bf16(f[2 * i] - bf16_lo(h[i]), f[2 * i + 1] - bf16_hi(h[i]));
    }
    u32x4 uh = {h[0], h[1], h[2], h[3]}, ul = {l[0], l[1], l[2], l[3]};
    V r; r.hi = __builtin_bit_cast(bf16x8_t, uh); r.lo = __builtin_bit_cast(bf16x8_t, ul); return r;
  }
  static __device__ __forceinline__ V load_b(const void* p) {
    const f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
    const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return split(f);
  }
  static __device__ __forceinline__ V load_a(const char* p, int img_stride) {
    V r;
    r.hi = *reinterpret_cast<const bf16x8_t*>(p);
    r.lo = *reinterpret_cast<const bf16x8_t*>(p + img_stride);
    return r;
  }
  // operand from a PRE-SPLIT LDS tile: hi plane at p, lo plane `plane` bytes further (both bf16).  The split happened once,
  // when the tile was committed to LDS - load_b() above splits on every read, i.e. once per (tap, k-step, column tile).
  static __device__ __forceinline__ V load_bp(const void* p, int plane) {
    V r;
    r.hi = *reinterpret_cast<const bf16x8_t*>(p);
    r.lo = *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const char*>(p) + plane);
    return r;
  }
  // 4 fp32 values -> 4 hi + 4 lo bf16 (8 bytes each)
  static __device__ __forceinline__ void split4(const f32x4& a, u32x2& hi, u32x2& lo) {
    hi[0] = pack_bf16(a[0], a[1]); hi[1] = pack_bf16(a[2], a[3]);
    lo[0] = pack_bf16(a[0] - bf16_lo(hi[0]), a[1] - bf16_hi(hi[0]));
    lo[1] = pack_bf16(a[2] - bf16_lo(hi[1]), a[3] - bf16_hi(hi[1]));
  }
  static __device__ __forceinline__ V from_acc(const f32x4& x, const f32x4& y) {
    const float f[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    return split(f);
  }
  static __device__ __forceinline__ void load4(const void* p, float* o) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
  }
  static __device__ __forceinline__ void store4(void* p, const float* v) {
    f32x4 a = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = a;
  }
  static __device__ __forceinline__ float round_store(float v) { return v; }
};

// fp32 storage, TWO products per MAC: activations split hi = f16(x), lo = f16(x - hi) (22 significant bits), weights a SINGLE f16
// (the f16 packed image): a.w * b.hi + a.w * b.lo.  One third fewer MFMAs and half the weight bytes of the bf16x3 mode; the price is the
// weights' rounding (2^-12 relative; tools/error_budget.py: +1.1e-4 in quadrature on the C2 waveform for the three MRF blocks).
struct f32w16 {};
template <> struct Mma<f32w16> {
  static constexpr int NSETS = 1;
  static constexpr int ES = 4;
  using ST = float;
  struct VA { f16x8_t v; };
  struct VB { f16x8_t hi, lo; };
  static __device__ __forceinline__ f32x4 mma(const VA& a, const VB& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.hi, c, 0, 0, 0);
  }
  static __device__ __forceinline__ VA load_a(const char* p, int) {
    VA r; r.v = *reinterpret_cast<const f16x8_t*>(p); return r;
  }
  static __device__ __forceinline__ VB load_bp(const void* p, int plane) {
    VB r;
    r.hi = *reinterpret_cast<const f16x8_t*>(p);
    r.lo = *reinterpret_cast<const f16x8_t*>(reinterpret_cast<const char*>(p) + plane);
    return r;
  }
  // hi by round-to-nearest-even (v_cvt_f16_f32), lo = f16(x - hi): exact difference, 11 more bits
  static __device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    hi = pack_f16((float)ha, (float)hb);
    lo = pack_f16(a - (float)ha, b - (float)hb);
  }
  static __device__ __forceinline__ void split4(const f32x4& a, u32x2& hi, u32x2& lo) {
    uint32_t h0, h1, l0, l1;
    split2(a[0], a[1], h0, l0);
    split2(a[2], a[3], h1, l1);
    hi = u32x2{h0, h1}; lo = u32x2{l0, l1};
  }
  static __device__ __forceinline__ VB from_acc(const f32x4& x, const f32x4& y) {
    uint32_t h[4], l[4];
    split2(x[0], x[1], h[0], l[0]); split2(x[2], x[3], h[1], l[1]);
    split2(y[0], y[1], h[2], l[2]); split2(y[2], y[3], h[3], l[3]);
    const u32x4 uh = {h[0], h[1], h[2], h[3]}, ul = {l[0], l[1], l[2], l[3]};
    VB r; r.hi = __builtin_bit_cast(f16x8_t, uh); r.lo = __builtin_bit_cast(f16x8_t, ul); return r;
  }
  static __device__ __forceinline__ void load4(const void* p, float* o) { Mma<float>::load4(p, o); }
  static __device__ __forceinline__ void store4(void* p, const float* v) { Mma<float>::store4(p, v); }
  static __device__ __forceinline__ void load4p(const void* p, float* o) { Mma<f16>::load4(p, o); }
  static __device__ __forceinline__ float round_store(float v) { return v; }
};
template <typename T> using StT = typename Mma<T>::ST;

// store one weight value into a packed A-fragment image set (pack kernels)
template <typename T> struct PackW;
template <> struct PackW<bf16> {
  static __device__ __forceinline__ void put(char* frag, int img_stride, int lane, int j, float v) {
    reinterpret_cast<bf16*>(frag + lane * 16)[j] = __float2bfloat16(v);
  }
};
template <> struct PackW<f16> {
  static __device__ __forceinline__ void put(char* frag, int img_stride, int lane, int j, float v) {
    reinterpret_cast<f16*>(frag + lane * 16)[j] = (f16)v;
  }
};
template <> struct PackW<f32w16> : PackW<f16> {};
template <> struct PackW<float> {
  static __device__ __forceinline__ void put(char* frag, int img_stride, int lane, int j, float v) {
    const bf16 h = __float2bfloat16(v);
    reinterpret_cast<bf16*>(frag + lane * 16)[j] = h;
    reinterpret_cast<bf16*>(frag + img_stride + lane * 16)[j] = __float2bfloat16(v - __bfloat162float(h));
  }
};

constexpr int FRAG_BYTES = 1024;  // one 16x32 operand image: 64 lanes x 16 bytes

// LDS row stride (bytes) for a channels-last operand tile read with ds_read_b128 by lane (row = lane&15, 16-byte chunk =
// lane>>4): brute force over the b128 lane groups shows the read is conflict-free iff stride % 128 is 32 or 96
// (16-bit storage); +16 is 2-way.  fp32 rows (two b128 per lane) stay at +16.
__host__ __device__ inline int lds_row_stride(int row_bytes, int elem_size) {
  if (elem_size != 2) return row_bytes + 16;
  for (int pad = 16; pad <= 128; pad += 16) {
    const int m = (row_bytes + pad) % 128;
    if (m == 32 || m == 96) return row_bytes + pad;
  }
  return row_bytes + 16;
}

// Global -> LDS staging with UB 16-byte loads per thread in flight.  A plain `load; store` loop makes the compiler wait
// for every load before the LDS store, i.e. one full memory latency per 16 bytes and thread.  f(i, src, dst) maps the
// linear piece index i to its source pointer (nullptr = zero fill) and its LDS byte offset.
template <int UB, int NT, typename F>
__device__ __forceinline__ void stage_batched(int tid, int total, char* lds_base, F f) {
  for (int i0 = tid; i0 < total; i0 += NT * UB) {
    u32x4 v[UB];
    int d[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int i = i0 + u * NT;
      const void* src = nullptr;
      d[u] = -1;
      if (i < total) f(i, src, d[u]);
      v[u] = u32x4{0u, 0u, 0u, 0u};
      if (src) v[u] = *reinterpret_cast<const u32x4*>(src);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (d[u] >= 0) *reinterpret_cast<u32x4*>(lds_base + d[u]) = v[u];
  }
}

// "hl8" rows (the block-to-block streams x' and f when HL is on): 192 bytes per time step - a hi plane of 64 f16 values (channel c
// at byte 2c) and a lo plane of 64 e4m3 bytes (channel c at byte 128 + c) holding (v - hi) * 2048: 15 significant bits, the pair
// row with its lo halves cut to 4.  tools/error_budget.py prices a 2^-15 rounding of both streams at ~1e-4 in quadrature on the
// waveform (a single f16, 2^-12, is what does not fit); 5 transfers of 67 MB per block become 5 of 50 MB.
constexpr int SR_HLB = 192;
constexpr float SR_LOS = 2048.f, SR_LOI = 1.f / 2048.f;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
struct Hl8 { u32x4 hi; u32x2 lo; };
__device__ __forceinline__ uint32_t hl8_pack4(float a, float b, float c, float d) {       // four residuals -> four e4m3 bytes
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -448.f, 448.f), __builtin_amdgcn_fmed3f(b, -448.f, 448.f), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c, -448.f, 448.f), __builtin_amdgcn_fmed3f(d, -448.f, 448.f), w, true);
  return (uint32_t)w;
}
__device__ __forceinline__ Hl8 hl8_encode(const float* v) {                               // 8 consecutive channels
  Hl8 r;
  float d[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 h0 = (_Float16)v[2 * i], h1 = (_Float16)v[2 * i + 1];
    const f16x2_t hp = {h0, h1};
    r.hi[i] = __builtin_bit_cast(uint32_t, hp);
    d[2 * i] = (v[2 * i] - (float)h0) * SR_LOS;
    d[2 * i + 1] = (v[2 * i + 1] - (float)h1) * SR_LOS;
  }
  r.lo[0] = hl8_pack4(d[0], d[1], d[2], d[3]);
  r.lo[1] = hl8_pack4(d[4], d[5], d[6], d[7]);
  return r;
}
__device__ __forceinline__ void hl8_lo_floats(u32x2 lo, float* o) {                       // the 8 residuals, unscaled
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo[i], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo[i], true);
    o[4 * i] = a[0] * SR_LOI; o[4 * i + 1] = a[1] * SR_LOI; o[4 * i + 2] = b[0] * SR_LOI; o[4 * i + 3] = b[1] * SR_LOI;
  }
}
__device__ __forceinline__ void hl8_decode(u32x4 hi, u32x2 lo, float* o) {
  hl8_lo_floats(lo, o);
  const f16x8_t h = __builtin_bit_cast(f16x8_t, hi);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] += (float)h[e];
}
__device__ __forceinline__ u32x4 hl8_lo_f16(u32x2 lo) {                                   // the residuals as the LDS image's f16 lo half (exact)
  float o[8];
  hl8_lo_floats(lo, o);
  return u32x4{pack_f16(o[0], o[1]), pack_f16(o[2], o[3]), pack_f16(o[4], o[5]), pack_f16(o[6], o[7])};
}

}  // namespace mv
