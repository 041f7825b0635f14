// Channels-last MFMA kernels for the discriminator stacks (the 99 % of a training step):
//   Discriminator2D: 5x Conv2d 3x3 pad 1 (1-32-64-128-256-1) + LeakyReLU(0.1)   hifigan_modified/discriminators.py:56-66
//   Discriminator1D: 5x Conv1d k15 pad 7 (same widths)      + LeakyReLU(0.1)   hifigan_modified/discriminators.py:97-107
// Activations are [B][H][W][C] (H = period for MPD, 1 for MSD), stride 1, "same" padding.
//
//  dconv_cl_kernel   implicit GEMM  D[o][w] = sum_{(dh,dw,c)} Wp[o][(dh,dw,c)] X[h+dh][w+dw][c]  for one output row
//                    segment; A = packed weights straight from L2 (prefetched one k-step ahead), B = kh input planes
//                    staged in LDS.  Forward: + bias, LeakyReLU.  Data gradient: the same kernel on flipped/transposed
//                    weights, its epilogue multiplies by LeakyReLU'(saved activation) so the result is already the
//                    gradient w.r.t. the previous layer's pre-activation.
//  dconv_head_*      the 256 -> 1 layer (forward dot, data-gradient outer product, weight-gradient reduction).
//  dconv_wgrad_kernel  gW[o][(tap,c)] = sum_pos g[pos][o] x[pos+tap][c]: a GEMM whose contraction index is TIME, so both
//                    operands are read from channels-last LDS tiles with ds_read_b64_tr_b16 (hardware transpose);
//                    position chunks are spread over workgroups and summed with fp32 atomics.
#include "mfma.h"
#include <cstdlib>

namespace mv {

#ifdef MV_DC_TIMING
__device__ long long* dc_dbg = nullptr;
#endif
struct DcP {
  int B, H, W, Cin, Cout, kh, kw, act;
  float slope;
  int ksteps, cin32;
  int dil;                                        // dilation along W (1 for the discriminators; GRC convs use 1/3/5)
  int cchunk;                                     // wide variant: input channels staged in LDS at a time
  // fused 256 -> 1 head (forward only; both null otherwise): after the epilogue tile is complete, z[tap][pos] = sum_c w[c][tap] * y[pos][c]
  // for this workgroup's positions - dhead_z_kernel's result without reading y back from HBM (needs all 256 output rows in the workgroup)
  const void* head_pf;                            // dhead_pack_kernel's forward operator [C/32][lane][8]
  float* head_z;                                  // [16][B*H*W] per-tap partial sums
  int head_taps;
};

// packed[mt][kstep][lane][8]: row o = 16*mt + lane&15, k-chunk = 4*kstep + lane>>4 -> tap = chunk / (Cin/8), c = 8*(chunk % (Cin/8)) + j
// flip == 0: forward weights  w[o][c][ih][iw]  (tap = ih*kw + iw)
// flip == 1: data-gradient weights: rows = c (previous-layer channels), k runs over (flipped tap, o)
template <typename T, typename P>
__global__ __launch_bounds__(256) void dconv_pack_kernel(const P* __restrict__ w, T* __restrict__ out, int Cout, int Cin,
                                                         int kh, int kw, int flip, int Coutp, int Cinp) {
  // w is [Cout][Cin][taps]; the packed operator is Coutp x Cinp (zero rows / columns beyond the real channel counts)
  const int M = flip ? Cinp : Coutp, Kc = flip ? Coutp : Cinp;
  const int taps = kh * kw, cpc = Kc / 8, ksteps = taps * (Kc / 32);
  const long total = (long)(M / 16) * ksteps * 512;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = idx % 8, lane = (idx / 8) % 64;
    const long fr = idx / 512;
    const int kstep = fr % ksteps, mt = fr / ksteps;
    const int row = 16 * mt + (lane & 15), chunk = 4 * kstep + (lane >> 4);
    const int tap = chunk / cpc, c = 8 * (chunk % cpc) + j;
    float v = 0.f;
    if (!flip) { if (row < Cout && c < Cin) v = ld<P>(w + ((long)row * Cin + c) * taps + tap); }
    else if (c < Cout && row < Cin) v = ld<P>(w + ((long)c * Cin + row) * taps + (taps - 1 - tap));   // w[o=c][cin=row][flipped tap]
    st<T>(out + idx, v);
  }
}

// Many weight tensors packed by ONE launch (after an optimizer step every conv weight of the stepped module changes at once):
// blockIdx.y selects the tensor, the body is dconv_pack_kernel's for fp32 masters with no channel padding.
struct MultiPackDesc { const float* src; void* dst; int Cout, Cin, kh, kw, flip, dtype; };
__global__ __launch_bounds__(256) void dconv_multi_pack_kernel(const MultiPackDesc* __restrict__ descs) {
  const MultiPackDesc d = descs[blockIdx.y];
  const int M = d.flip ? d.Cin : d.Cout, Kc = d.flip ? d.Cout : d.Cin;
  const int taps = d.kh * d.kw, cpc = Kc / 8, ksteps = taps * (Kc / 32);
  const long total = (long)(M / 16) * ksteps * 512;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = idx % 8, lane = (idx / 8) % 64;
    const long fr = idx / 512;
    const int kstep = fr % ksteps, mt = fr / ksteps;
    const int row = 16 * mt + (lane & 15), chunk = 4 * kstep + (lane >> 4);
    const int tap = chunk / cpc, c = 8 * (chunk % cpc) + j;
    const float v = !d.flip ? d.src[((long)row * d.Cin + c) * taps + tap] : d.src[((long)c * d.Cin + row) * taps + (taps - 1 - tap)];
    if (d.dtype == MV_BF16) st<bf16>((bf16*)d.dst + idx, v); else st<f16>((f16*)d.dst + idx, v);
  }
}

// XCD-aware tile order for the [B][H][W] convs.  Workgroups are dealt to the 8 XCDs round-robin by linear id, and every XCD has its own
// L2; with the plain (W tile, row tile, b*H + h) grid the three workgroups that read the same input row (h-1, h, h+1 of a 3x3 conv)
// sit gridDim.x ids apart - on different XCDs - and the row comes out of HBM / MALL once per XCD (PMC: the 256->128 data gradient
// fetched 4.8x its input).  Here XCD k walks a contiguous range of the order (b, row tile, W tile, h) with h FASTEST, so the rows a
// workgroup shares with its predecessor are in that XCD's L2.  The map is a bijection for any grid size (guide: section 5, XCD swizzle).
struct TileId { int wx, ry, b, h; };
__device__ __forceinline__ TileId xcd_tile(int H) {
  const unsigned nx = gridDim.x, ny = gridDim.y, nwg = nx * ny * gridDim.z;
  const unsigned L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
  const unsigned xcd = L & 7, j = L >> 3, q = nwg >> 3, r = nwg & 7;
  unsigned LL = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  TileId t;
  t.h = LL % H; LL /= H;
  t.wx = LL % nx; LL /= nx;
  t.ry = LL % ny;
  t.b = LL / ny;
  return t;
}

// the fused head pass of a conv epilogue: `tile` = LDS epilogue tile [NPOS positions][256 channels] (row stride ORS bytes), already activated
template <typename T, int NPOS, int NWVS>
__device__ __forceinline__ void dhead_from_tile(const char* tile, int ORS, const DcP& p, long rowbase, int w0, int wid, int lane) {
  using M = Mma<T>;
  using V = typename M::V;
  constexpr int ES = M::ES, KS = 256 / 32;
  const int col = lane & 15, g = lane >> 4;
  const long npos = (long)p.B * p.H * p.W;
  V a[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a[ks] = M::load_b(reinterpret_cast<const char*>(p.head_pf) + ((long)ks * 64 + lane) * 16);
  for (int pb = wid; pb < NPOS / 16; pb += NWVS) {
    const char* xr = tile + (long)(pb * 16 + col) * ORS + 8 * g * ES;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = M::mma(a[ks], M::load_b(xr + ks * 32 * ES), acc);
    const int ww = w0 + pb * 16 + col;
    if (ww < p.W) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r < p.head_taps) p.head_z[(long)(4 * g + r) * npos + rowbase + ww] = acc[r];
    }
  }
}

template <typename T, int NWV, int MW, int NB>
__global__ __launch_bounds__(NWV * 64, 2) void dconv_cl_kernel(const T* __restrict__ x, const T* __restrict__ wp,
                                                       const T* __restrict__ bias, const T* __restrict__ actsave,
                                                       T* __restrict__ y, DcP p) {
  using M = Mma<T>;
  using V = typename M::V;
  constexpr int ES = M::ES;
  extern __shared__ __align__(16) char lds[];
  const int RS = lds_row_stride(p.Cin * ES, ES);
  const int prow = NB * 16 + (p.kw - 1) * p.dil;  // staged columns per plane
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const TileId tl = xcd_tile(p.H);
  const int w0 = tl.wx * NB * 16;
  const int mt0 = (tl.ry * NWV + wid) * MW;
  const int b = tl.b, h = tl.h;
  const int n_mt = p.Cout / 16;
  const int ph = p.kh / 2, pw = (p.kw / 2) * p.dil;

  // ---- stage kh input planes: rows h-ph..h+ph, columns w0-pw .. w0+NB*16+pw-1, zero outside the image.
  //      UB global loads per thread are in flight together (a load -> wait -> LDS-store loop costs one full memory
  //      latency per 16 bytes and dominated the 3x3 layers)
  {
    constexpr int UB = 8;
    const int cpr = p.Cin * ES / 16;             // 16-byte pieces per row; a thread keeps its piece and walks rows
    const int rpp = (NWV * 64) / cpr;            // rows covered by one sweep of the workgroup (threads beyond rpp*cpr idle)
    const int ch = tid % cpr;
    const int nrow = tid < rpp * cpr ? p.kh * prow : 0;
    const char* xb = reinterpret_cast<const char*>(x + (long)b * p.H * p.W * p.Cin) + ch * 16;
    for (int R0 = tid / cpr; R0 < nrow; R0 += rpp * UB) {
      u32x4 v[UB];
      int dsto[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int R = R0 + u * rpp;
        int pl = 0, r = R;
        while (r >= prow) { r -= prow; ++pl; }   // kh <= 3 planes
        const int hh = h - ph + pl, ww = w0 - pw + r;
        v[u] = u32x4{0u, 0u, 0u, 0u};
        dsto[u] = R < nrow ? R * RS + ch * 16 : -1;
        if (R < nrow && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
          v[u] = *reinterpret_cast<const u32x4*>(xb + ((long)hh * p.W + ww) * p.Cin * ES);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (dsto[u] >= 0) *reinterpret_cast<u32x4*>(lds + dsto[u]) = v[u];
    }
  }
  __syncthreads();

  f32x4 acc[MW][NB];
#pragma unroll
  for (int mw = 0; mw < MW; ++mw)
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[mw][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* wlane = reinterpret_cast<const char*>(wp) + (long)lane * 16;
  // weight fragments stream L2 -> registers two k-steps ahead of their use (the x tile is already in LDS)
  V a0[MW], a1[MW], a2[MW];
  auto wfetch = [&](int kstep, V (&dst)[MW]) {
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) {
      const int mt = (mt0 + mw) < n_mt ? (mt0 + mw) : (n_mt - 1);
      dst[mw] = M::load_b(wlane + ((long)mt * p.ksteps + kstep) * 1024);
    }
  };
  // unconditional fetches (clamped to the last k-step): a branch around the loads would force vmcnt(0) at every k-step
  const int klast = p.ksteps - 1;
  wfetch(0, a0);
  wfetch(klast < 1 ? klast : 1, a1);
  int tap = 0, c32 = 0;
  for (int kstep = 0; kstep < p.ksteps; ++kstep) {
    wfetch(kstep + 2 < p.ksteps ? kstep + 2 : klast, a2);
    const int ih = tap / p.kw, iw = tap % p.kw;
    const char* bbase = lds + ((long)ih * prow + iw * p.dil + col) * RS + (c32 * 32 + 8 * g) * ES;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const V bf = M::load_b(bbase + (long)(n * 16) * RS);
#pragma unroll
      for (int mw = 0; mw < MW; ++mw) acc[mw][n] = M::mma(a0[mw], bf, acc[mw][n]);
    }
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) { a0[mw] = a1[mw]; a1[mw] = a2[mw]; }
    if (++c32 == p.cin32) { c32 = 0; ++tap; }
  }

  // ---- epilogue through LDS: [w][rows of this workgroup] -> whole-row stores
  __syncthreads();
  constexpr int RW = NWV * MW * 16;
  constexpr int ORS = RW * ES + 16;
  const int R0 = tl.ry * RW;
  auto epi = [&](auto actf) {       // one body per activation kind: a run-time `act` in the element loop is a branch tree per element
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) {
      const int mt = mt0 + mw;
      if (mt < n_mt) {
        const int row = 16 * mt + 4 * g;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) M::load4(bias + row, bv);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          float ov[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) ov[i] = actf(acc[mw][n][i] + bv[i]);
          M::store4(lds + (long)(n * 16 + col) * ORS + (row - R0) * ES, ov);
        }
      }
    }
  };
  if (p.act <= ACT_LRELU) epi(ActLrelu{p.act == ACT_NONE ? 1.f : p.slope}); else epi(ActAny{p.act, p.slope});
  __syncthreads();
  if constexpr (RW == 256 && ES == 2) {
    if (p.head_pf) dhead_from_tile<T, NB * 16, NWV>(lds, ORS, p, ((long)b * p.H + h) * p.W, w0, wid, lane);
  }
  {
    constexpr int EPC = 16 / ES;
    constexpr int CPR = RW / EPC;
    T* yrow = y + (((long)b * p.H + h) * p.W) * p.Cout;
    const T* srow = actsave ? actsave + (((long)b * p.H + h) * p.W) * p.Cout : nullptr;
    for (int i = tid; i < NB * 16 * CPR; i += NWV * 64) {
      const int ch = i % CPR, wi = i / CPR;
      const int ww = w0 + wi, row = R0 + ch * EPC;
      if (ww >= p.W || row >= p.Cout) continue;
      u32x4 val = *reinterpret_cast<const u32x4*>(lds + (long)wi * ORS + ch * 16);
      if (srow) {   // data-gradient mode: multiply by LeakyReLU'(pre-activation) = (saved activation >= 0 ? 1 : slope)
        float gvals[16 / ES], svals[16 / ES];
        T tmp[16 / ES];
        *reinterpret_cast<u32x4*>(tmp) = val;
        const u32x4 sv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(srow + (long)ww * p.Cout + row));
        T stmp[16 / ES];
        *reinterpret_cast<u32x4*>(stmp) = sv;
#pragma unroll
        for (int e = 0; e < 16 / ES; ++e) {
          gvals[e] = ld<T>(tmp + e);
          svals[e] = ld<T>(stmp + e);
          st<T>(tmp + e, svals[e] >= 0.f ? gvals[e] : gvals[e] * p.slope);
        }
        val = *reinterpret_cast<u32x4*>(tmp);
      }
      *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(yrow + (long)ww * p.Cout + row)) = val;
    }
  }
}

// Wide-input variant (Cin = 256: the data gradient of the 128->256 layer).  The whole-Cin tile of dconv_cl_kernel leaves room
// for 4 column blocks and 4 waves per CU (one wave per SIMD, nothing to hide latency behind).  Here the input is staged in
// 128-channel chunks, the tile keeps 128 positions, and 8 waves split it as 2 position halves x 4 row groups (2 M-tiles x 4
// column blocks each): two waves per SIMD on the same LDS budget.  Weight k-steps are fetched two ahead, across chunks.
template <typename T, int PS, int MW, int NB, int CCH, int NWV = 8>
__global__ __launch_bounds__(NWV * 64, 2) void dconv_cl_wide_kernel(const T* __restrict__ x, const T* __restrict__ wp,
                                                            const T* __restrict__ bias, const T* __restrict__ actsave,
                                                            T* __restrict__ y, DcP p) {
  using M = Mma<T>;
  using V = typename M::V;
  constexpr int ES = M::ES, RG = NWV / PS, NPOS = PS * NB * 16, NT = NWV * 64;   // PS position slices x RG row groups = NWV waves; CCH channels staged at a time
  extern __shared__ __align__(16) char lds[];
  const int RS = lds_row_stride(CCH * ES, ES);
  const int prow = NPOS + (p.kw - 1) * p.dil;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int ps = wid / RG, rg = wid % RG;                // position slice, row group
  const TileId tl = xcd_tile(p.H);
  const int w0 = tl.wx * NPOS;
  const int mt0 = (tl.ry * RG + rg) * MW;
  const int b = tl.b, h = tl.h;
  const int n_mt = p.Cout / 16;
  const int ph = p.kh / 2, pw = (p.kw / 2) * p.dil;
  const int taps = p.kh * p.kw, nchunks = p.Cin / CCH;

  auto stage = [&](int ck) {
    constexpr int UB = 8;
    constexpr int cpr = CCH * ES / 16, rpp = NT / cpr;
    const int ch = tid % cpr;
    const int nrow = p.kh * prow;
    const char* xb = reinterpret_cast<const char*>(x + (long)b * p.H * p.W * p.Cin + (long)ck * CCH) + ch * 16;
    for (int R0 = tid / cpr; R0 < nrow; R0 += rpp * UB) {
      u32x4 v[UB];
      int dsto[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int R = R0 + u * rpp;
        int pl = 0, r = R;
        while (r >= prow) { r -= prow; ++pl; }
        const int hh = h - ph + pl, ww = w0 - pw + r;
        v[u] = u32x4{0u, 0u, 0u, 0u};
        dsto[u] = R < nrow ? R * RS + ch * 16 : -1;
        if (R < nrow && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
          v[u] = *reinterpret_cast<const u32x4*>(xb + ((long)hh * p.W + ww) * p.Cin * ES);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (dsto[u] >= 0) *reinterpret_cast<u32x4*>(lds + dsto[u]) = v[u];
    }
  };

  f32x4 acc[MW][NB];
#pragma unroll
  for (int mw = 0; mw < MW; ++mw)
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[mw][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* wlane = reinterpret_cast<const char*>(wp) + (long)lane * 16;
  V a0[MW], a1[MW], a2[MW];
  auto wfetch = [&](int kid, V (&dst)[MW]) {
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) {
      const int mt = (mt0 + mw) < n_mt ? (mt0 + mw) : (n_mt - 1);
      dst[mw] = M::load_b(wlane + ((long)mt * p.ksteps + kid) * 1024);
    }
  };
  // k-steps in execution order: chunk-major, then tap, then the 4 channel blocks of the chunk
  constexpr int cc32 = CCH / 32;
  int fck = 0, ftap = 0, fj = 0, fetched = 0;
  // The fetch is UNCONDITIONAL (past the last k-step it re-reads the last fragment): a branch around the loads makes the
  // compiler wait for vmcnt(0) at every k-step, i.e. for the fragments it has just requested - the prefetch distance is lost.
  auto fetch_next = [&](V (&dst)[MW]) {
    wfetch(ftap * p.cin32 + fck * cc32 + fj, dst);
    if (++fetched < p.ksteps) {                       // scalar cursor update only; the last k-step id stays put
      if (++fj == cc32) { fj = 0; if (++ftap == taps) { ftap = 0; ++fck; } }
    }
  };
  fetch_next(a0);
  fetch_next(a1);
  for (int ck = 0; ck < nchunks; ++ck) {
    if (ck) __syncthreads();
    stage(ck);
    __syncthreads();
    for (int tap = 0; tap < taps; ++tap) {
      const int ih = tap / p.kw, iw = tap - ih * p.kw;
      const char* brow = lds + ((long)ih * prow + iw * p.dil + ps * NB * 16 + col) * RS + 8 * g * ES;
#pragma unroll
      for (int j = 0; j < cc32; ++j) {
        fetch_next(a2);
        V bf[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) bf[n] = M::load_b(brow + j * 32 * ES + (long)(n * 16) * RS);
        // all NB operand reads in flight before the first MFMA: left alone, the scheduler issues them two at a time, each pair
        // right in front of the 2 x MW MFMAs that use it (rr wait MMMM wait MMMM rr ...), one LDS round trip per 8 MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int mw = 0; mw < MW; ++mw) acc[mw][n] = M::mma(a0[mw], bf[n], acc[mw][n]);
#pragma unroll
        for (int mw = 0; mw < MW; ++mw) { a0[mw] = a1[mw]; a1[mw] = a2[mw]; }
      }
    }
  }

  // ---- epilogue through LDS: [position][128 rows of this workgroup] -> whole-row stores
  __syncthreads();
  constexpr int RW = RG * MW * 16;
  constexpr int ORS = RW * ES + 16;
  const int R0 = tl.ry * RW;
  auto epi = [&](auto actf) {
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) {
      const int mt = mt0 + mw;
      if (mt < n_mt) {
        const int row = 16 * mt + 4 * g;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) M::load4(bias + row, bv);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          float ov[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) ov[i] = actf(acc[mw][n][i] + bv[i]);
          M::store4(lds + (long)(ps * NB * 16 + n * 16 + col) * ORS + (row - R0) * ES, ov);
        }
      }
    }
  };
  if (p.act <= ACT_LRELU) epi(ActLrelu{p.act == ACT_NONE ? 1.f : p.slope}); else epi(ActAny{p.act, p.slope});
  __syncthreads();
  if constexpr (RW == 256 && ES == 2) {
    if (p.head_pf) dhead_from_tile<T, NPOS, NWV>(lds, ORS, p, ((long)b * p.H + h) * p.W, w0, wid, lane);
  }
  {
    constexpr int EPC = 16 / ES;
    constexpr int CPR = RW / EPC;
    T* yrow = y + (((long)b * p.H + h) * p.W) * p.Cout;
    const T* srow = actsave ? actsave + (((long)b * p.H + h) * p.W) * p.Cout : nullptr;
    for (int i = tid; i < NPOS * CPR; i += NT) {
      const int ch = i % CPR, wi = i / CPR;
      const int ww = w0 + wi, row = R0 + ch * EPC;
      if (ww >= p.W || row >= p.Cout) continue;
      u32x4 val = *reinterpret_cast<const u32x4*>(lds + (long)wi * ORS + ch * 16);
      if (srow) {
        alignas(16) T tmp[16 / ES];
        alignas(16) T stmp[16 / ES];
        *reinterpret_cast<u32x4*>(tmp) = val;
        *reinterpret_cast<u32x4*>(stmp) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(srow + (long)ww * p.Cout + row));
#pragma unroll
        for (int e = 0; e < 16 / ES; ++e) {
          const float gv = ld<T>(tmp + e);
          st<T>(tmp + e, ld<T>(stmp + e) >= 0.f ? gv : gv * p.slope);
        }
        val = *reinterpret_cast<u32x4*>(tmp);
      }
      *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(yrow + (long)ww * p.Cout + row)) = val;
    }
  }
}

// ------------------------------------------------------------------------------------------------ 256 -> 1 head
// The head has ONE output channel, so the MFMA rows carry the TAPS instead:
//   forward   z[tap][pos] = sum_c w[c][tap] x[pos][c]      (a 1x1 "conv" with 16 output rows = taps, K = C; x read once,
//             straight from HBM as the B operand - no shifted windows), then  y[pos] = b + sum_tap z[tap][pos + off(tap)]
//   dgrad     gx[pos][c]  = lrelu'(save[pos][c]) * sum_tap w[c][tap] gy[pos - off(tap)]   (rows = channels, K = taps <= 32:
//             one MFMA per 16 channels x 16 positions; rows are permuted so a lane owns C/4 CONTIGUOUS channels)
// Both are HBM-bound: x / gx / save are touched exactly once.
constexpr int DH_C = 256;                           // head input width of both discriminator families

// packed forward operator [C/32][lane][8]: row = tap (lane&15), k = 32*ks + 8*(lane>>4) + j = channel
// packed dgrad operator   [C/16][lane][8]: m-tile mt, row i = lane&15 -> channel (i>>2)*(C/4) + 4*mt + (i&3); k = tap
template <typename T, typename P>
__global__ __launch_bounds__(256) void dhead_pack_kernel(const P* __restrict__ w, T* __restrict__ pf, T* __restrict__ pd, int C, int taps) {
  const int nf = (C / 32) * 512, nd = (C / 16) * 512;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nf + nd; idx += gridDim.x * blockDim.x) {
    if (idx < nf) {
      const int j = idx % 8, lane = (idx / 8) % 64, ks = idx / 512;
      const int tap = lane & 15, c = 32 * ks + 8 * (lane >> 4) + j;
      st<T>(pf + idx, tap < taps ? ld<P>(w + (long)c * taps + tap) : 0.f);
    } else {
      const int i2 = idx - nf;
      const int j = i2 % 8, lane = (i2 / 8) % 64, mt = i2 / 512;
      const int i = lane & 15, c = (i >> 2) * (C / 4) + 4 * mt + (i & 3), tap = 8 * (lane >> 4) + j;
      st<T>(pd + i2, tap < taps ? ld<P>(w + (long)c * taps + tap) : 0.f);
    }
  }
}

template <typename T, int C>
__global__ __launch_bounds__(256) void dhead_z_kernel(const T* __restrict__ x, const T* __restrict__ pf, float* __restrict__ z,
                                                      long npos, int taps) {
  using M = Mma<T>;
  using V = typename M::V;
  constexpr int KS = C / 32;
  const int lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4;
  V a[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a[ks] = M::load_b(reinterpret_cast<const char*>(pf) + ((long)ks * 64 + lane) * 16);
  const long ngroups = (npos + 15) / 16;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  for (long grp = wave; grp < ngroups; grp += nwaves) {
    const long pos = grp * 16 + col;
    const long pc = pos < npos ? pos : npos - 1;
    const char* xr = reinterpret_cast<const char*>(x + pc * C) + g * 16;
    V bfr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bfr[ks] = M::load_b(xr + ks * 64);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = M::mma(a[ks], bfr[ks], acc);
    if (pos < npos) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r < taps) z[(long)(4 * g + r) * npos + pos] = acc[r];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dhead_sum_kernel(const float* __restrict__ z, const T* __restrict__ bias, T* __restrict__ y,
                                                        int B, int H, int W, int kh, int kw, int flip) {
  const long npos = (long)B * H * W;
  const int ph = kh / 2, pw = kw / 2;
  const float bv = bias ? ld<T>(bias) : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npos; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W), h = (int)((i / W) % H);
    float acc = bv;
    for (int ih = 0; ih < kh; ++ih) {
      const int dh = flip ? -(ih - ph) : ih - ph, hh = h + dh;     // flip: the adjoint (data gradient) reaches pos - off(tap)
      if (hh < 0 || hh >= H) continue;
      for (int iw = 0; iw < kw; ++iw) {
        const int dw = flip ? -(iw - pw) : iw - pw, ww = w + dw;
        if (ww < 0 || ww >= W) continue;
        acc += z[(long)(ih * kw + iw) * npos + i + (long)dh * W + dw];
      }
    }
    st<T>(y + i, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dhead_dgrad_kernel(const T* __restrict__ gy, const T* __restrict__ pd, const T* __restrict__ save,
                                                          T* __restrict__ gx, int B, int H, int W, int kh, int kw, float slope) {
  using M = Mma<T>;
  using V = typename M::V;
  constexpr int NMT = DH_C / 16;
  const long npos = (long)B * H * W;
  const int lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4;
  const int ph = kh / 2, pw = kw / 2, taps = kh * kw;
  const long ngroups = (npos + 15) / 16;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const char* pa = reinterpret_cast<const char*>(pd) + (long)lane * 16;
  for (long grp = wave; grp < ngroups; grp += nwaves) {
    const long pos = grp * 16 + col;
    const bool ok = pos < npos;
    const long pc = ok ? pos : npos - 1;
    const int w = (int)(pc % W), h = (int)((pc / W) % H);
    // B[k = tap = 8g + j][col]: gy at the position this tap reaches (gy is tiny and cache-resident).  All 8 gathers are
    // unconditional loads from clamped addresses (independent, in flight together); validity is applied afterwards.
    float gval[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tap = 8 * g + j;
      const int ih = tap / kw, iw = tap - ih * kw;
      const int hh = h - (ih - ph), ww = w - (iw - pw);
      const bool valid = tap < taps && hh >= 0 && hh < H && ww >= 0 && ww < W;
      const long idx = valid ? pc - (long)(ih - ph) * W - (iw - pw) : pc;
      const float raw = ld<T>(gy + idx);
      gval[j] = valid ? raw : 0.f;
    }
    // this lane's C/4 contiguous channels: g*(C/4) + 4*mt + r; the saved activations are fetched before the MFMAs
    const T* srow = save + pc * DH_C + g * (DH_C / 4);
    T* orow = gx + pc * DH_C + g * (DH_C / 4);
    u32x4 svv[NMT / 2];
#pragma unroll
    for (int m2 = 0; m2 < NMT; m2 += 2) svv[m2 / 2] = *reinterpret_cast<const u32x4*>(srow + 4 * m2);
    alignas(16) T bt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) st<T>(bt + j, gval[j]);
    const V bfr = M::load_b(bt);
#pragma unroll
    for (int m2 = 0; m2 < NMT; m2 += 2) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      const f32x4 d0 = M::mma(M::load_b(pa + (long)m2 * 1024), bfr, zero);
      const f32x4 d1 = M::mma(M::load_b(pa + (long)(m2 + 1) * 1024), bfr, zero);
      alignas(16) T sv[8];
      alignas(16) T ov[8];
      *reinterpret_cast<u32x4*>(sv) = svv[m2 / 2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st<T>(ov + r, ld<T>(sv + r) >= 0.f ? d0[r] : d0[r] * slope);
        st<T>(ov + 4 + r, ld<T>(sv + 4 + r) >= 0.f ? d1[r] : d1[r] * slope);
      }
      if (ok) *reinterpret_cast<u32x4*>(orow + 4 * m2) = *reinterpret_cast<u32x4*>(ov);
    }
  }
}

// weight gradient of the head: gw[tap][c] = sum_{b,pos} g[pos] x[pos+tap][c]; gb = sum g.
// One workgroup = one output row (b,h) x a chunk of columns; thread = channel.  Every x element is read ONCE (all taps that
// use it are accumulated from the LDS-staged g window), results leave with fp32 atomics.
constexpr int DH_MAXTAPS = 16;
template <typename T>
__global__ __launch_bounds__(256) void dhead_wgrad_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                          float* __restrict__ gw, float* __restrict__ gb, int H, int W,
                                                          int C, int kh, int kw, int chunk) {
  extern __shared__ float gs[];                     // [kh][chunk + kw - 1] g rows h-ph..h+ph (as seen from the x row)
  const int b = blockIdx.z, hx = blockIdx.y;        // hx: row of x being read
  const int ph = kh / 2, pw = kw / 2;
  const int w0 = blockIdx.x * chunk;
  const int gwid = chunk + kw - 1;
  // x[hx][wx] contributes to gw[ih][iw] with g[hx - (ih-ph)][wx - (iw-pw)]
  for (int i = threadIdx.x; i < kh * gwid; i += blockDim.x) {
    const int ih = i / gwid, j = i % gwid;
    const int hg = hx - (ih - ph), wg = w0 - pw + j;   // j indexes wg from w0-pw .. w0+chunk+pw-1
    gs[i] = (hg >= 0 && hg < H && wg >= 0 && wg < W) ? ld<T>(g + ((long)b * H + hg) * W + wg) : 0.f;
  }
  __syncthreads();
  float acc[DH_MAXTAPS];
#pragma unroll
  for (int t = 0; t < DH_MAXTAPS; ++t) acc[t] = 0.f;
  const int wend = (w0 + chunk < W) ? w0 + chunk : W;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    for (int ih = 0; ih < kh; ++ih) {
#pragma unroll
      for (int t = 0; t < DH_MAXTAPS; ++t) acc[t] = 0.f;
      for (int wx = w0; wx < wend; ++wx) {
        const float xv = ld<T>(x + (((long)b * H + hx) * W + wx) * C + c);
        const float* gr = gs + ih * gwid + (wx - w0) + 2 * pw;   // g at wx - (iw - pw) = index (wx - w0 + pw) + (pw - iw)
#pragma unroll
        for (int iw = 0; iw < DH_MAXTAPS; ++iw)
          if (iw < kw) acc[iw] += gr[-iw] * xv;
      }
#pragma unroll
      for (int iw = 0; iw < DH_MAXTAPS; ++iw)
        if (iw < kw) atomicAdd(gw + (ih * kw + iw) * C + c, acc[iw]);
    }
  }
  if (gb && hx < H && threadIdx.x == 0) {            // bias: sum of g over this (row, chunk) - rows of g == rows of x
    float s = 0.f;
    for (int wx = w0; wx < wend; ++wx) s += gs[ph * gwid + (wx - w0) + pw];
    atomicAdd(gb, s);
  }
}

// ------------------------------------------------------------------------------------------------ one-channel weight gradients
// Both one-channel layers reduce a channels-last VECTOR tensor against shifted copies of a SCALAR tensor:
//   first layer (flip = 0): gw[o][tap] = sum_pos g1[pos][o] * x0[pos + tap - pad]     vec = g1 [..][C1], sc = x0
//   head        (flip = 1): gw[tap][c] = sum_pos x[pos][c]  * g[pos - (tap - pad)]    vec = x  [..][C],  sc = g
// out index = c * so + tap * st.  A thread owns 8 channels (one 16-byte load per position) and all taps in registers; the
// scalar window of the chunk sits in LDS.  Workgroups loop over row chunks, reduce across their threads through LDS and
// add one value per (channel, tap) to HBM.  bias_mode 1: gb[c] = sum_pos vec[pos][c];  2: gb[0] = sum_pos sc[pos].
template <typename T, int KH, int KW>
__global__ __launch_bounds__(256) void dtap_wgrad_kernel(const T* __restrict__ vec, const T* __restrict__ sc, float* __restrict__ gw,
                                                         float* __restrict__ gb, int H, int W, int C, int flip, int so, int st_,
                                                         int bias_mode, int WT, int wsplit, int nchunks, int cpw) {
  constexpr int TAPS = KH * KW, PH = KH / 2, PW = KW / 2;
  extern __shared__ float sm[];
  const int gwid = WT + KW - 1;
  float* win = sm;                                  // [KH][gwid]
  float* red = sm + KH * gwid;                      // [256][8]
  const int tid = threadIdx.x;
  const int tpr = C / 8, cg = tid % tpr, part = tid / tpr, nparts = 256 / tpr;
  float acc[TAPS][8];
  float bacc[8];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) bacc[e] = 0.f;
  float bsum = 0.f;
  const int cbeg = blockIdx.x * cpw;
  int cend = cbeg + cpw; if (cend > nchunks) cend = nchunks;
  for (int chunk = cbeg; chunk < cend; ++chunk) {
    const int bh = chunk / wsplit, wc = chunk - bh * wsplit;
    const int b = bh / H, h = bh - b * H, w0 = wc * WT;
    if (chunk > cbeg) __syncthreads();
    for (int i = tid; i < KH * gwid; i += 256) {
      const int ih = i / gwid, j = i - ih * gwid;
      const int hh = flip ? h - (ih - PH) : h + (ih - PH), ww = w0 - PW + j;
      win[i] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? ld<T>(sc + ((long)b * H + hh) * W + ww) : 0.f;
    }
    __syncthreads();
    const int wend = (w0 + WT < W) ? w0 + WT : W;
    const T* vrow = vec + (((long)b * H + h) * W) * C + cg * 8;
    // two positions per iteration, software-pipelined: the loads of the NEXT pair are in flight during this pair's FMAs
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto ldpos = [&](int wq) { return wq < wend ? *reinterpret_cast<const u32x4*>(vrow + (long)wq * C) : zero4; };
    u32x4 nx = ldpos(w0 + part), nx2 = ldpos(w0 + part + nparts);
    for (int w = w0 + part; w < wend; w += 2 * nparts) {
      const int w2 = w + nparts;
      const bool two = w2 < wend;
      alignas(16) T tmp[8];
      alignas(16) T tmp2[8];
      *reinterpret_cast<u32x4*>(tmp) = nx;
      *reinterpret_cast<u32x4*>(tmp2) = nx2;
      nx = ldpos(w + 2 * nparts);
      nx2 = ldpos(w2 + 2 * nparts);
      float v[8], v2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] = ld<T>(tmp + e); v2[e] = ld<T>(tmp2 + e); }
      const float* wp = win + (w - w0) + (flip ? 2 * PW : 0);
      const float* wp2 = two ? wp + nparts : wp;          // v2 is zero when the second position does not exist
#pragma unroll
      for (int ih = 0; ih < KH; ++ih)
#pragma unroll
        for (int iw = 0; iw < KW; ++iw) {
          const float sv = flip ? wp[ih * gwid - iw] : wp[ih * gwid + iw];
          const float sv2 = flip ? wp2[ih * gwid - iw] : wp2[ih * gwid + iw];
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[ih * KW + iw][e] += v[e] * sv + v2[e] * sv2;
        }
      if (bias_mode == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bacc[e] += v[e] + v2[e];
      } else if (bias_mode == 2 && cg == 0) {
        bsum += win[PH * gwid + (w - w0) + PW] + (two ? win[PH * gwid + (w2 - w0) + PW] : 0.f);
      }
    }
  }
  // workgroup reduction, one (tap) slice at a time; threads 0..C-1 own the final sums
  const int myc = tid;                              // channel index for the final sum
#pragma unroll
  for (int t = 0; t <= TAPS; ++t) {
    if (t == TAPS && bias_mode != 1) break;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = (t < TAPS) ? acc[t < TAPS ? t : 0][e] : bacc[e];
    __syncthreads();
    if (myc < C) {
      const int g8 = myc >> 3, e = myc & 7;
      float s = 0.f;
      for (int q = 0; q < nparts; ++q) s += red[(q * tpr + g8) * 8 + e];
      if (t < TAPS) atomicAdd(gw + (long)myc * so + (long)t * st_, s); else atomicAdd(gb + myc, s);
    }
  }
  if (bias_mode == 2) {
    __syncthreads();
    const float s = block_sum(bsum, red);
    if (tid == 0) atomicAdd(gb, s);
  }
}

template <typename T>
static int dtap_launch(const void* vec, const void* sc, float* gw, float* gb, int B, int H, int W, int C, int kh, int kw, int flip,
                       int so, int st_, int bias_mode, hipStream_t s) {
  if (C % 8 != 0 || C > 256 || 256 % (C / 8) != 0) return MV_ERR_UNSUPPORTED;
  const int WT = 512;
  const int wsplit = cdiv(W, WT);
  const long nchunks = (long)B * H * wsplit;
  if (nchunks > (1L << 30)) return MV_ERR_UNSUPPORTED;
  int groups = nchunks < 1024 ? (int)nchunks : 1024;
  const int cpw = (int)((nchunks + groups - 1) / groups);
  groups = (int)((nchunks + cpw - 1) / cpw);
  const size_t lds = sizeof(float) * ((size_t)kh * (WT + kw - 1) + 256 * 8);
  if (kh == 3 && kw == 3)
    hipLaunchKernelGGL((dtap_wgrad_kernel<T, 3, 3>), dim3(groups), dim3(256), lds, s, (const T*)vec, (const T*)sc, gw, gb, H, W, C, flip,
                       so, st_, bias_mode, WT, wsplit, (int)nchunks, cpw);
  else if (kh == 1 && kw == 15)
    hipLaunchKernelGGL((dtap_wgrad_kernel<T, 1, 15>), dim3(groups), dim3(256), lds, s, (const T*)vec, (const T*)sc, gw, gb, H, W, C, flip,
                       so, st_, bias_mode, WT, wsplit, (int)nchunks, cpw);
  else return MV_ERR_UNSUPPORTED;
  return MV_OK;
}

// ------------------------------------------------------------------------------------------------ MFMA weight gradient
// gws[tap][o][c] (fp32, zero on entry) += sum over this workgroup's positions of g[pos][o] * x[pos+tap][c]
// workgroup = 8 waves arranged 2 (tap halves) x 2 (o) x 2 (c): wave (th, wm, wn) owns output channels o0 + 32*wm .. +31
// (2 M-tiles), input channels c0 + 32*wn .. +31 (2 N-tiles) and taps th*TPW .. th*TPW+TPW-1, so one staged (g, x) chunk
// feeds ALL taps of the 64x64 channel tile.  Chunks (WT positions of one image row + halo) are double-buffered in LDS:
// the global loads of chunk i+1 are in flight (registers) while chunk i is multiplied.  Both operands are read with the
// hardware-transposed ds_read_b64_tr_b16; the contraction index (position) may be permuted freely as long as A and B
// agree, so a lane group reads rows 4*grp+q / 16+4*grp+q - with a 160-byte row stride that is bank-conflict-free.
// Accumulators stay in registers across all chunks of the workgroup; only the final tiles go to HBM (fp32 atomics into
// the tap-major workspace, lanes along c -> 64-byte segments), then dconv_wgrad_reorder_kernel writes [o][c][tap].
template <typename T, int TAPS_H, int TAPS_W, int WT>
__global__ __launch_bounds__(512) void dconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                          float* __restrict__ gws, float* __restrict__ gb, int B, int H, int W,
                                                          int Cin, int Cout, int wsplit, int nchunks, int chunks_per_wg, int dil,
                                                          int xW, long sample_stride) {
  // xW: rows per image line of x (= W, or W + 1 for the even-tap ODConv adjoint whose x operand is one row longer);
  // sample_stride > 0: per-sample tiles - a workgroup's chunks all belong to sample (first chunk) / (H * wsplit)
  static_assert(sizeof(T) == 2, "MFMA weight gradient needs 16-bit storage");
  using M = Mma<T>;
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  constexpr int TAPS = TAPS_H * TAPS_W, TPW = (TAPS + 1) / 2;
  constexpr int PH = TAPS_H / 2;
  constexpr int RS = 160;                                   // 64 channels x 2 bytes + 32: conflict-free transposed reads
  constexpr int GROWS = WT;
  constexpr int MAXHALO = TAPS_H == 1 ? 64 : TAPS_W - 1;    // (TAPS_W - 1) * dil, checked by the launcher
  constexpr int NLD = ((GROWS + TAPS_H * (WT + MAXHALO)) * 8 + 511) / 512;   // 16-byte pieces per thread per chunk
  const int PW = (TAPS_W & 1) ? (TAPS_W / 2) * dil : 0;    // even tap counts: offsets 0 .. TAPS_W-1 (no centring)
  const int XCOLS = WT + (TAPS_W - 1) * dil;
  const int ROWS = GROWS + TAPS_H * XCOLS;
  const int BUF = ROWS * RS;
  extern __shared__ __align__(16) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int th = wid >> 2, wm = (wid >> 1) & 1, wn = wid & 1;
  const int grp = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;   // transposed-read lane roles (guide T10)
  const int o0 = blockIdx.y * 64, c0 = blockIdx.x * 64;

  f32x4 acc[TPW][2][2];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[t][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  int toff[TPW];                                             // wave-uniform LDS offset of each tap's shifted x window
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = (th * TPW + t) < TAPS ? (th * TPW + t) : (TAPS - 1);
    toff[t] = ((tap / TAPS_W) * XCOLS + (tap % TAPS_W) * dil) * RS;
  }
  const int ga = (4 * grp + q) * RS + (wm * 32 + 4 * pp) * 2;
  const int xa = GROWS * RS + (4 * grp + q) * RS + (wn * 32 + 4 * pp) * 2;
  auto trload = [&](const char* p0) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0 + 16 * RS));
    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    typename M::V r;
    r.v = __builtin_bit_cast(decltype(r.v), both);
    return r;
  };

  u32x4 pre[NLD];
  auto issue = [&](int chunk) {
    const int bh = chunk / wsplit, wc = chunk - bh * wsplit;
    const int b = bh / H, h = bh - b * H;
    const int w0 = wc * WT;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * 512;
      const int r = idx >> 3, ch = idx & 7;
      pre[i] = u32x4{0u, 0u, 0u, 0u};
      if (r < GROWS) {
        const int ww = w0 + r;
        if (ww < W && o0 + ch * 8 < Cout) pre[i] = *reinterpret_cast<const u32x4*>(g + (((long)b * H + h) * W + ww) * Cout + o0 + ch * 8);
      } else if (r < ROWS) {
        const int rr = r - GROWS;
        const int pl = TAPS_H == 1 ? 0 : rr / XCOLS, col = rr - pl * XCOLS;
        const int hh = h - PH + pl, ww = w0 - PW + col;
        if (hh >= 0 && hh < H && ww >= 0 && ww < xW && c0 + ch * 8 < Cin)
          pre[i] = *reinterpret_cast<const u32x4*>(x + (((long)b * H + hh) * xW + ww) * Cin + c0 + ch * 8);
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * 512;
      if (idx < ROWS * 8) *reinterpret_cast<u32x4*>(lds + buf * BUF + (idx >> 3) * RS + (idx & 7) * 16) = pre[i];
    }
  };

  // bias gradient gb[o] = sum_pos g[pos][o] rides along: the g tile is already in LDS (first Cin tile's workgroups only)
  const bool do_gb = gb != nullptr && blockIdx.x == 0;
  float bsum = 0.f;
  const int cbeg = blockIdx.z * chunks_per_wg;
  int cend = cbeg + chunks_per_wg; if (cend > nchunks) cend = nchunks;
  if (cbeg < cend) { issue(cbeg); commit(0); }
  __syncthreads();
  int buf = 0;
  for (int chunk = cbeg; chunk < cend; ++chunk) {
    const bool more = chunk + 1 < cend;                 // uniform across the workgroup
    if (more) issue(chunk + 1);
    const char* base = lds + buf * BUF;
    if (do_gb) {
      const T* gcol = reinterpret_cast<const T*>(base) + (tid & 63);
#pragma unroll
      for (int i = 0; i < WT / 8; ++i)
        bsum += ld<T>(reinterpret_cast<const T*>(reinterpret_cast<const char*>(gcol) + ((tid >> 6) + 8 * i) * RS));
    }
#pragma unroll 1
    for (int k0 = 0; k0 < WT; k0 += 32) {
      typename M::V a[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = trload(base + ga + k0 * RS + m * 32);
      // a batch of transposed B fragments first (independent LDS reads in flight), then their MFMAs
      constexpr int BT = TPW > 5 ? 4 : TPW;
#pragma unroll
      for (int tb = 0; tb < TPW; tb += BT) {
        typename M::V bfr[BT][2];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            if (tb + t < TPW) bfr[t][n] = trload(base + xa + toff[tb + t] + k0 * RS + n * 32);
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m)
              if (tb + t < TPW) acc[tb + t][m][n] = M::mma(a[m], bfr[t][n], acc[tb + t][m][n]);
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (do_gb) {                                              // the 8 row-slice sums of a channel -> one atomic per channel and workgroup
    float* red = reinterpret_cast<float*>(lds);            // (the loop's last barrier is behind us)
    red[tid] = bsum;
    __syncthreads();
    if (tid < 64 && o0 + tid < Cout) {
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sacc += red[r * 64 + tid];
      atomicAdd(gb + o0 + tid, sacc);
    }
  }
  if (sample_stride > 0) gws += (long)(cbeg / (H * wsplit)) * sample_stride;
  // D[row = o (4*grp + r)][col = c (li)] -> gws[tap][o][c]
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = th * TPW + t;
    if (tap >= TAPS) continue;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int o = o0 + wm * 32 + m * 16 + 4 * grp, c = c0 + wn * 32 + n * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (o + r < Cout && c < Cin) atomicAdd(gws + ((long)tap * Cout + o + r) * Cin + c, acc[t][m][n][r]);
      }
  }
}

// 3x3 (dilation 1) weight gradient, second form.  The kernel above stages, per 64 positions of an image row, the three x rows
// h-1, h, h+1 and a 64 x 64 channel tile: 33 KB of loads and 197 KB of LDS reads for 40 MFMAs per wave - both paths saturate
// long before the matrix pipe (128->256 layer: 23 % of the MFMA peak).  Here
//   * a workgroup walks its positions ROW AFTER ROW of one 64-column strip, so x rows stay in LDS: a ring of 8 row slots indexed
//     by the "virtual row" v = strip * (H + 2) + h + 1 (every strip carries a zero row above and below); a step brings ONE new
//     x row (two at a strip change) instead of three,
//   * a wave owns MO = 4 M-tiles (64 output channels): LDS reads per MFMA 0.6 -> 0.35, g/x loads per MFMA 2.7x lower,
//   * the tap split is 5 + 4 (the second wave half skips its fifth tap instead of multiplying a clamped copy).
// Layout of the result and the final atomics are those of the kernel above.
template <typename T, int MO>
__global__ __launch_bounds__(512, 2) void dconv_wgrad3_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                              float* __restrict__ gws, float* __restrict__ gb, int B, int H, int W,
                                                              int Cin, int Cout, int wsplit, int nchunks, int chunks_per_wg) {
  static_assert(sizeof(T) == 2, "MFMA weight gradient needs 16-bit storage");
  using M = Mma<T>;
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  constexpr int WT = 64, XC = WT + 2, TAPS = 9, TPW = 5, OT = 32 * MO;
  constexpr int RS = 160, RSG = OT * 2 + 32;                 // row strides: conflict-free transposed reads (stride = 32 mod 128)
  constexpr int OPC = OT / 8, GP = WT * OPC, PP = XC * 8;    // 16-byte pieces: per g row, per g tile, per x row slot
  constexpr int NLD = (GP + 2 * PP + 511) / 512;
  constexpr int GBUF = WT * RSG, PBUF = XC * RS, XBASE = 2 * GBUF;
  extern __shared__ __align__(16) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int th = wid >> 2, wm = (wid >> 1) & 1, wn = wid & 1;
  const int grp = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int o0 = blockIdx.y * OT, c0 = blockIdx.x * 64;
  const int ntap = th ? TAPS - TPW : TPW;                    // 5 + 4
  const int HV = H + 2;

  f32x4 acc[TPW][MO][2];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int m = 0; m < MO; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[t][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ga = (4 * grp + q) * RSG + (wm * MO * 16 + 4 * pp) * 2;
  const int xa = (4 * grp + q) * RS + (wn * 32 + 4 * pp) * 2;
  auto trload = [&](const char* p0, int stride) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0 + 16 * stride));
    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    typename M::V r;
    r.v = __builtin_bit_cast(decltype(r.v), both);
    return r;
  };
  // x rows live in slots 0..7 (slot = virtual row & 7); slot 8 is a row of zeros that stands for every padding row
  constexpr int ZSLOT = 8;
  u32x4 pre[NLD];
  // loads of: the g tile at (gb_, gh, gw0) when with_g, and up to two x rows (row index r0 / r1 of sample pb, strip column pw0;
  // a negative row = none).  The piece -> (row, 16-byte column) maps are compile-time shifts; nothing here divides.
  auto issue = [&](bool with_g, int gb_, int gh, int gw0, int pb, int pw0, int r0, int r1) {
    const T* grow = g + (((long)gb_ * H + gh) * W + gw0) * Cout + o0;
    const T* xrow0 = x + (((long)pb * H + (r0 < 0 ? 0 : r0)) * W + pw0 - 1) * Cin + c0;
    const T* xrow1 = x + (((long)pb * H + (r1 < 0 ? 0 : r1)) * W + pw0 - 1) * Cin + c0;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * 512;
      pre[i] = u32x4{0u, 0u, 0u, 0u};
      if (idx < GP) {
        const int r = idx / OPC, ch = idx - r * OPC;
        if (with_g && gw0 + r < W && o0 + ch * 8 < Cout) pre[i] = *reinterpret_cast<const u32x4*>(grow + r * Cout + ch * 8);
      } else if (idx < GP + 2 * PP) {
        const int j = idx - GP;
        const int sel = j >= PP ? 1 : 0, rr = j - sel * PP;
        const int col = rr >> 3, ch = rr & 7;
        const int ww = pw0 - 1 + col;
        if ((sel ? r1 : r0) >= 0 && ww >= 0 && ww < W && c0 + ch * 8 < Cin)
          pre[i] = *reinterpret_cast<const u32x4*>((sel ? xrow1 : xrow0) + col * Cin + ch * 8);
      }
    }
  };
  auto commit = [&](bool with_g, int gbuf, int s0, int s1) {      // s0 / s1: destination slots of the two rows (negative = none)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * 512;
      if (idx < GP) {
        if (with_g) { const int r = idx / OPC, ch = idx - r * OPC; *reinterpret_cast<u32x4*>(lds + gbuf * GBUF + r * RSG + ch * 16) = pre[i]; }
      } else if (idx < GP + 2 * PP) {
        const int j = idx - GP;
        const int sel = j >= PP ? 1 : 0, rr = j - sel * PP;
        const int sl = sel ? s1 : s0;
        if (sl >= 0) *reinterpret_cast<u32x4*>(lds + XBASE + sl * PBUF + (rr >> 3) * RS + (rr & 7) * 16) = pre[i];
      }
    }
  };

  const bool do_gb = gb != nullptr && blockIdx.x == 0;
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
  const int cbeg = blockIdx.z * chunks_per_wg;
  int cend = cbeg + chunks_per_wg; if (cend > nchunks) cend = nchunks;
  if (cbeg >= cend) return;
  // chunk = (sample b, strip wc, row h), rows fastest; v = virtual row of h (each strip: zero row, H rows, zero row)
  int b, wc, h, v;
  {
    const int st = cbeg / H;
    h = cbeg - st * H;
    b = st / wsplit;
    wc = st - b * wsplit;
    v = st * HV + h + 1;
    for (int k = tid; k < PP; k += 512) *reinterpret_cast<u32x4*>(lds + XBASE + ZSLOT * PBUF + (k >> 3) * RS + (k & 7) * 16) = u32x4{0u, 0u, 0u, 0u};
    issue(true, b, h, wc * WT, b, wc * WT, h > 0 ? h - 1 : -1, h);
    commit(true, 0, h > 0 ? ((v - 1) & 7) : -1, v & 7);
    issue(false, b, h, wc * WT, b, wc * WT, h + 1 < H ? h + 1 : -1, -1);
    commit(false, 0, h + 1 < H ? ((v + 1) & 7) : -1, -1);
  }
  __syncthreads();
  int par = 0;
#ifdef MV_DC_TIMING
  long long tacc[5] = {0, 0, 0, 0, 0}, tprev = clock64(), tnow;
#define W3_TM(i) do { tnow = clock64(); tacc[i] += tnow - tprev; tprev = tnow; } while (0)
#else
#define W3_TM(i) do {} while (0)
#endif
  for (int chunk = cbeg; chunk < cend; ++chunk) {
    const bool more = chunk + 1 < cend;                     // uniform across the workgroup
    // the next chunk and the x rows it adds: one row below in the same strip, or rows 0 and 1 of the next strip
    int nb = b, nwc = wc, nh = h + 1, nv = v + 1, r0, r1, s0, s1;
    if (nh == H) { nh = 0; nv = v + 3; if (++nwc == wsplit) { nwc = 0; ++nb; } r0 = 0; r1 = H > 1 ? 1 : -1; s0 = nv & 7; s1 = H > 1 ? ((nv + 1) & 7) : -1; }
    else { r0 = nh + 1 < H ? nh + 1 : -1; r1 = -1; s0 = r0 >= 0 ? ((nv + 1) & 7) : -1; s1 = -1; }
    if (more) issue(true, nb, nh, nwc * WT, nb, nwc * WT, r0, r1);
    int toff[TPW];                                          // wave-uniform LDS offset of each tap's shifted x window
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tap = (th * TPW + t) < TAPS ? (th * TPW + t) : (TAPS - 1);
      const int ih = tap / 3, iw = tap - 3 * ih;
      const int hh = h - 1 + ih;
      toff[t] = XBASE + ((hh >= 0 && hh < H) ? ((v - 1 + ih) & 7) : ZSLOT) * PBUF + iw * RS;
    }
    const char* gbase = lds + par * GBUF;
    W3_TM(0);
    if (do_gb) {                                            // bias gradient from the staged g tile: 8 channels x 2 rows per thread
      constexpr int RPP = 512 / OPC;
#pragma unroll
      for (int i = 0; i < WT / RPP; ++i) {
        alignas(16) T tmp[8];
        *reinterpret_cast<u32x4*>(tmp) = *reinterpret_cast<const u32x4*>(gbase + ((tid / OPC) + RPP * i) * RSG + (tid % OPC) * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += ld<T>(tmp + e);
      }
    }
    W3_TM(1);
#pragma unroll 1
    for (int k0 = 0; k0 < WT; k0 += 32) {
      typename M::V a[MO];
#pragma unroll
      for (int m = 0; m < MO; ++m) a[m] = trload(gbase + ga + k0 * RSG + m * 32, RSG);
      constexpr int BT = 2;
#pragma unroll
      for (int tb = 0; tb < TPW; tb += BT) {
        if (tb < ntap) {
          typename M::V bfr[BT][2];
#pragma unroll
          for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n)
              if (tb + t < TPW) bfr[t][n] = trload(lds + toff[tb + t] + xa + k0 * RS + n * 32, RS);
#pragma unroll
          for (int t = 0; t < BT; ++t)
            if (tb + t < TPW && tb + t < ntap) {
#pragma unroll
              for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < MO; ++m) acc[tb + t][m][n] = M::mma(a[m], bfr[t][n], acc[tb + t][m][n]);
            }
        }
      }
    }
    W3_TM(2);
    if (more) commit(true, par ^ 1, s0, s1);
    W3_TM(3);
    __syncthreads();
    W3_TM(4);
    par ^= 1;
    b = nb; wc = nwc; h = nh; v = nv;
  }
#ifdef MV_DC_TIMING
  if (tid == 0 && dc_dbg) {
    const long wgid = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wgid < 4096) { for (int i = 0; i < 5; ++i) dc_dbg[wgid * 8 + i] = tacc[i] / (cend - cbeg); dc_dbg[wgid * 8 + 5] = cend - cbeg; dc_dbg[wgid * 8 + 6] = dc_dbg[wgid * 8 + 7] = -1; }
  }
#endif
  if (do_gb) {                                              // row-slice sums -> one atomic per channel and workgroup
    float* red = reinterpret_cast<float*>(lds);            // [512 / OPC slices][OT] (the loop's last barrier is behind us)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid / OPC) * OT + (tid % OPC) * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < OT && o0 + tid < Cout) {
      float sacc = 0.f;
      for (int r = 0; r < 512 / OPC; ++r) sacc += red[r * OT + tid];
      atomicAdd(gb + o0 + tid, sacc);
    }
  }
  // D[row = o (4*grp + r)][col = c (li)] -> gws[tap][o][c]
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = th * TPW + t;
    if (tap >= TAPS) continue;
#pragma unroll
    for (int m = 0; m < MO; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int o = o0 + wm * MO * 16 + m * 16 + 4 * grp, c = c0 + wn * 32 + n * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (o + r < Cout && c < Cin) atomicAdd(gws + ((long)tap * Cout + o + r) * Cin + c, acc[t][m][n][r]);
      }
  }
}

// gw[o][c][tap] (reference layout [Cout][Cin][kh][kw]) = gws[tap][o][c]
__global__ __launch_bounds__(256) void dconv_wgrad_reorder_kernel(const float* __restrict__ gws, float* __restrict__ gw,
                                                                  int OC, int taps) {
  const long total = (long)OC * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const long oc = i / taps;
    gw[i] = gws[(long)tap * OC + oc];
  }
}

// The same reorder for a workspace that STAYS zero between calls: every element read is cleared (each is read exactly once), the
// bias sums behind the weights are handed out and cleared too - the next weight gradient needs no fill launch.
__global__ __launch_bounds__(256) void dconv_wgrad_finalize_kernel(float* __restrict__ gws, float* __restrict__ gw, float* __restrict__ gb,
                                                                   int OC, int taps, int Cout) {
  const long total = (long)OC * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total + Cout; i += (long)gridDim.x * blockDim.x) {
    if (i < total) {
      const int tap = (int)(i % taps);
      const long src = (long)tap * OC + i / taps;
      gw[i] = gws[src];
      gws[src] = 0.f;
    } else {
      if (gb) gb[i - total] = gws[i];
      gws[i] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------ first layer (1 -> C1 channels)
// forward: a1[pos][o] = lrelu(b[o] + sum_tap w[o][tap] x0[pos+tap]);  x0 [B][H][W] (one channel), a1 [B][H][W][C1]
template <typename T>
__global__ __launch_bounds__(256) void dfirst_fwd_kernel(const T* __restrict__ x0, const T* __restrict__ w, const T* __restrict__ bias,
                                                         T* __restrict__ y, int H, int W, int C1, int kh, int kw, float slope) {
  const int b = blockIdx.y, ph = kh / 2, pw = kw / 2;
  const long npos = (long)H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npos * C1; i += (long)gridDim.x * blockDim.x) {
    const int o = (int)(i % C1);
    const long pos = i / C1;
    const int h = (int)(pos / W), ww0 = (int)(pos % W);
    float acc = bias ? ld<T>(bias + o) : 0.f;
    for (int ih = 0; ih < kh; ++ih) {
      const int hh = h + ih - ph;
      if (hh < 0 || hh >= H) continue;
      for (int iw = 0; iw < kw; ++iw) {
        const int ww = ww0 + iw - pw;
        if (ww < 0 || ww >= W) continue;
        acc += ld<T>(w + (long)o * kh * kw + ih * kw + iw) * ld<T>(x0 + (long)b * npos + (long)hh * W + ww);
      }
    }
    st<T>(y + ((long)b * npos + pos) * C1 + o, acc >= 0.f ? acc : acc * slope);
  }
}

// register-tiled forward: a thread owns 8 output channels (their KH*KW weights live in registers) and walks the positions
// of the workgroup's row chunk; the one-channel input window sits in LDS; 16-byte channels-last stores
template <typename T, int KH, int KW>
__global__ __launch_bounds__(256) void dfirst_fwd_tiled_kernel(const T* __restrict__ x0, const T* __restrict__ w, const T* __restrict__ bias,
                                                               T* __restrict__ y, int H, int W, int C1, float slope, int WT, int wsplit) {
  constexpr int TAPS = KH * KW, PH = KH / 2, PW = KW / 2;
  extern __shared__ float win[];                     // [KH][WT + KW - 1]
  const int gwid = WT + KW - 1;
  const int tid = threadIdx.x;
  const int tpr = C1 / 8, cg = tid % tpr, part = tid / tpr, nparts = 256 / tpr;
  const int chunk = blockIdx.x, bh = chunk / wsplit, wc = chunk - bh * wsplit;
  const int b = bh / H, h = bh - b * H, w0 = wc * WT;
  for (int i = tid; i < KH * gwid; i += 256) {
    const int ih = i / gwid, j = i - ih * gwid;
    const int hh = h + ih - PH, ww = w0 - PW + j;
    win[i] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? ld<T>(x0 + ((long)b * H + hh) * W + ww) : 0.f;
  }
  float wr[TAPS][8], bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bv[e] = bias ? ld<T>(bias + cg * 8 + e) : 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) wr[t][e] = ld<T>(w + (long)(cg * 8 + e) * TAPS + t);
  }
  __syncthreads();
  const int wend = (w0 + WT < W) ? w0 + WT : W;
  T* yrow = y + (((long)b * H + h) * W) * C1 + cg * 8;
  for (int wx = w0 + part; wx < wend; wx += nparts) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bv[e];
    const float* wp = win + (wx - w0);
#pragma unroll
    for (int ih = 0; ih < KH; ++ih)
#pragma unroll
      for (int iw = 0; iw < KW; ++iw) {
        const float xv = wp[ih * gwid + iw];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += wr[ih * KW + iw][e] * xv;
      }
    alignas(16) T ov[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) st<T>(ov + e, acc[e] >= 0.f ? acc[e] : acc[e] * slope);
    *reinterpret_cast<u32x4*>(yrow + (long)wx * C1) = *reinterpret_cast<u32x4*>(ov);
  }
}

// data-gradient operator of the first layer for the tap-row z-GEMM (dhead_z_kernel<T, 32>): row = tap, k = channel
template <typename T>
__global__ void dfirst_dpack_kernel(const T* __restrict__ w, T* __restrict__ pf, int C1, int taps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (C1 / 32) * 512) return;
  const int j = idx % 8, lane = (idx / 8) % 64, ks = idx / 512;
  const int tap = lane & 15, c = 32 * ks + 8 * (lane >> 4) + j;
  pf[idx] = tap < taps ? w[(long)c * taps + tap] : T(0.f);
}

// data gradient: gx0[pos] = sum_{o,tap} w[o][tap] g1[pos - tap][o]   (g1 = d/d pre-activation, [B][H][W][C1]); one wave per position
template <typename T>
__global__ __launch_bounds__(256) void dfirst_dgrad_kernel(const T* __restrict__ g1, const T* __restrict__ w, T* __restrict__ gx0,
                                                           int H, int W, int C1, int kh, int kw) {
  const int lane = threadIdx.x & 63, b = blockIdx.y, ph = kh / 2, pw = kw / 2;
  const long npos = (long)H * W;
  const long pos = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pos >= npos) return;
  const int h = (int)(pos / W), ww0 = (int)(pos % W);
  float acc = 0.f;
  const int taps = kh * kw;
  for (int e = lane; e < taps * C1; e += 64) {
    const int o = e % C1, tap = e / C1, ih = tap / kw, iw = tap % kw;
    const int hh = h - (ih - ph), ww = ww0 - (iw - pw);
    if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
    acc += ld<T>(w + (long)o * taps + tap) * ld<T>(g1 + ((long)b * npos + (long)hh * W + ww) * C1 + o);
  }
  acc = wave_sum(acc);
  if (lane == 0) st<T>(gx0 + (long)b * npos + pos, acc);
}

// weight gradient: gw[o][tap] += sum_pos g1[pos][o] x0[pos+tap]; gb[o] += sum_pos g1[pos][o]   (atomics across workgroups)
template <typename T>
__global__ __launch_bounds__(256) void dfirst_wgrad_kernel(const T* __restrict__ g1, const T* __restrict__ x0, float* __restrict__ gw,
                                                           float* __restrict__ gb, int H, int W, int C1, int kh, int kw, int chunk) {
  __shared__ float red[256 * 4];
  const int b = blockIdx.y, ph = kh / 2, pw = kw / 2, taps = kh * kw;
  const long npos = (long)H * W;
  const int o = threadIdx.x % C1, part = threadIdx.x / C1, nparts = blockDim.x / C1;
  const long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk < npos ? p0 + chunk : npos;
  for (int tap = -1; tap < taps; ++tap) {       // tap -1 = bias
    const int ih = tap < 0 ? 0 : tap / kw, iw = tap < 0 ? 0 : tap % kw;
    float acc = 0.f;
    for (long pos = p0 + part; pos < p1; pos += nparts) {
      const float gv = ld<T>(g1 + ((long)b * npos + pos) * C1 + o);
      if (tap < 0) { acc += gv; continue; }
      const int h = (int)(pos / W), ww0 = (int)(pos % W);
      const int hh = h + ih - ph, ww = ww0 + iw - pw;
      if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
      acc += gv * ld<T>(x0 + (long)b * npos + (long)hh * W + ww);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (part == 0) {
      float s = 0.f;
      for (int q = 0; q < nparts; ++q) s += red[q * C1 + o];
      if (tap < 0) atomicAdd(gb + o, s); else atomicAdd(gw + (long)o * taps + tap, s);
    }
    __syncthreads();
  }
}

// out[c] += sum_rows x[row][c]   (bias gradients of channels-last tensors); one thread = 16 bytes of a row
template <typename T>
__global__ __launch_bounds__(256) void colsum_cl_kernel(const T* __restrict__ x, float* __restrict__ out, long rows, int C, int chunk) {
  constexpr int EPC = 16 / sizeof(T);
  __shared__ float red[256 * EPC];
  const int tpr = C / EPC;                           // threads per row
  const int cg = threadIdx.x % tpr, part = threadIdx.x / tpr, nparts = blockDim.x / tpr;
  const long r0 = (long)blockIdx.x * chunk, r1 = r0 + chunk < rows ? r0 + chunk : rows;
  float acc[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
  for (long r = r0 + part; r < r1; r += nparts) {
    T tmp[EPC];
    *reinterpret_cast<u32x4*>(tmp) = *reinterpret_cast<const u32x4*>(x + r * C + cg * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] += ld<T>(tmp + e);
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) red[threadIdx.x * EPC + e] = acc[e];
  __syncthreads();
  if (part == 0) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float sum = 0.f;
      for (int q = 0; q < nparts; ++q) sum += red[(q * tpr + cg) * EPC + e];
      atomicAdd(out + cg * EPC + e, sum);
    }
  }
}

// out[row] = x[row][c]  /  y[row][:] = 0, y[row][c] = g[row]   (the C -> 1 head rides the MFMA conv with zero-padded channels)
template <typename T>
__global__ __launch_bounds__(256) void take_channel_kernel(const T* __restrict__ x, T* __restrict__ out, long rows, int C, int c) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) out[r] = x[r * C + c];
}
template <typename T>
__global__ __launch_bounds__(256) void put_channel_kernel(const T* __restrict__ g, T* __restrict__ y, long rows, int C, int c) {
  const long n = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int ch = (int)(i % C);
    T z; st<T>(&z, 0.f);
    y[i] = (ch == c) ? g[r] : z;
  }
}

}  // namespace mv

using namespace mv;

extern "C" size_t mv_dconv_packed_bytes(int Cout, int Cin, int kh, int kw, int dtype) {
  return (size_t)Cout * Cin * kh * kw * (dtype == MV_F32 ? 4 : 2);
}

extern "C" int mv_dconv_pack_pad(const void* w, int param_dtype, void* packed, int Cout, int Cin, int kh, int kw, int Coutp,
                                 int Cinp, int flip, int dtype, void* stream) {
  MV_CHECK_ARG(w && packed && Cout > 0 && Cin > 0 && Coutp >= Cout && Cinp >= Cin && kh > 0 && kw > 0);
  const int Mr = flip ? Cinp : Coutp, Kc = flip ? Coutp : Cinp;
  MV_CHECK_ARG(Mr % 16 == 0 && Kc % 32 == 0);
  const long total = (long)Coutp * Cinp * kh * kw;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  MV_DISPATCH(dtype, {
    switch (param_dtype) {
      case MV_F32: hipLaunchKernelGGL((dconv_pack_kernel<T, float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)w, (T*)packed, Cout, Cin, kh, kw, flip, Coutp, Cinp); break;
      case MV_BF16: hipLaunchKernelGGL((dconv_pack_kernel<T, bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)w, (T*)packed, Cout, Cin, kh, kw, flip, Coutp, Cinp); break;
      case MV_F16: hipLaunchKernelGGL((dconv_pack_kernel<T, f16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16*)w, (T*)packed, Cout, Cin, kh, kw, flip, Coutp, Cinp); break;
      default: return MV_ERR_DTYPE;
    }
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_dconv_pack(const void* w, int param_dtype, void* packed, int Cout, int Cin, int kh, int kw, int flip,
                             int dtype, void* stream) {
  MV_CHECK_ARG(Cout % 16 == 0 && Cin % 32 == 0);
  return mv_dconv_pack_pad(w, param_dtype, packed, Cout, Cin, kh, kw, Cout, Cin, flip, dtype, stream);
}

extern "C" int mv_dconv_multi_pack(const void* descs_dev, int n, void* stream) {
  MV_CHECK_ARG(descs_dev && n > 0 && n <= 65535);
  hipLaunchKernelGGL(dconv_multi_pack_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, (const MultiPackDesc*)descs_dev);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

template <typename T, int NWV, int MW, int NB>
static int dconv_launch(const void* x, const void* wp, const void* bias, const void* actsave, void* y, DcP p, hipStream_t s) {
  using M = Mma<T>;
  const int prow = NB * 16 + (p.kw - 1) * p.dil;
  const size_t xb = (size_t)p.kh * prow * lds_row_stride(p.Cin * M::ES, M::ES);
  const size_t ob = (size_t)NB * 16 * (NWV * MW * 16 * M::ES + 16);
  const size_t lds = xb > ob ? xb : ob;
  if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
  const int cpr = p.Cin * M::ES / 16;
  if (cpr > NWV * 64) return MV_ERR_UNSUPPORTED;   // staging: a thread keeps its 16-byte piece of every row it copies
  auto kern = dconv_cl_kernel<T, NWV, MW, NB>;
  static size_t lds_set = 0;
  if (lds > lds_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); lds_set = lds; }
  dim3 grid(cdiv(p.W, NB * 16), cdiv(p.Cout / 16, NWV * MW), p.B * p.H);
  if (grid.z > 65535) return MV_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), lds, s, (const T*)x, (const T*)wp, (const T*)bias, (const T*)actsave, (T*)y, p);
  return MV_OK;
}

static int dconv_cl_fwd_impl(const void* x, const void* packed, const void* bias, const void* act_save, void* y,
                             int B, int H, int W, int Cin, int Cout, int kh, int kw, int dil_w, int act, float slope, int dtype,
                             void* stream, const void* head_pf, float* head_z, int head_taps) {
  MV_CHECK_ARG(x && packed && y && B > 0 && H > 0 && W > 0 && Cin % 32 == 0 && Cout % 16 == 0 && (kh & 1) && (kw & 1));
  MV_CHECK_ARG(dil_w >= 1 && (kw - 1) * dil_w <= 128);
  MV_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)packed & 15) == 0 && Cout % 8 == 0);
  if (head_pf && (Cout != 256 || dtype == MV_F32 || act_save || B * H > 65535)) return MV_ERR_UNSUPPORTED;   // needs the 256-row workgroups
  DcP p{B, H, W, Cin, Cout, kh, kw, act, slope, kh * kw * (Cin / 32), Cin / 32, dil_w, Cin, head_pf, head_z, head_taps};
  int rc = MV_ERR_DTYPE;
  if (dtype != MV_F32 && B * H <= 65535) {
    // 8-wave chunked variant: picked where it measured faster than the whole-Cin tile (tools/bench_dconv.py, B=32):
    // every wide input (Cin > 128), and narrow outputs of 3x3 layers, where the whole-Cin kernel leaves waves idle
    static int force = -2;
    if (force == -2) { const char* e = getenv("MV_DCONV_WIDE"); force = e ? atoi(e) : -1; }
    const size_t full = (size_t)kh * (128 + (kw - 1) * dil_w) * lds_row_stride(Cin * 2, 2);
    const bool big3x3 = kh == 3 && Cin == 128 && Cout >= 256;      // 256 rows per workgroup: 4 M-tiles per B fragment
    bool use = full > 160 * 1024 || Cin > 128 || (kh == 3 && Cin >= 64 && Cout <= 128) || (kh == 1 && Cin == 64 && Cout == 32) || big3x3;
    if (force >= 0) use = force != 0;
    if (head_pf && !(big3x3 || (kh == 3 && Cin % 64 == 0 && Cin >= 128 && W >= 256))) use = false;   // the chunked tiles below have 256 rows only there
    // 3x3 layers with >= 128 channels on both sides: 256-position tiles (8 column blocks per wave: one weight fragment feeds
    // 8 MFMAs) with 64-channel chunks measured 6-13 % faster than 128-position tiles; the k15 layers measured slower
    const bool nb8 = kh == 3 && Cin % 64 == 0 && Cin >= 128 && Cout >= 128 && W >= 256;
    // 4-wave form of the 256-row tile (MV_DCONV_W4=0 switches it off): 256 rows x 128 positions, 64-channel chunks, TWO workgroups per CU - one's
    // staging and epilogue run under the other's MFMAs; every wave keeps the 64 x 128 tile of the 8-wave kernel
    static int w4 = -1;
    if (w4 < 0) { const char* e = getenv("MV_DCONV_W4"); w4 = e ? atoi(e) : 1; }
    if (w4 && kh == 3 && Cin % 64 == 0 && Cout % 256 == 0 && W >= 128 && force != 0) {
      p.cchunk = 64;
      const size_t xb = (size_t)kh * (128 + (kw - 1) * dil_w) * lds_row_stride(64 * 2, 2);
      const size_t ob = (size_t)128 * (256 * 2 + 16);
      const size_t ldsb = xb > ob ? xb : ob;
      dim3 grid(cdiv(W, 128), Cout / 256, B * H);
#define MV_W4(TT) do { auto kern = dconv_cl_wide_kernel<TT, 1, 4, 8, 64, 4>; static size_t lds_set_4 = 0; \
        if (ldsb > lds_set_4) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); lds_set_4 = ldsb; } \
        hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, (hipStream_t)stream, (const TT*)x, (const TT*)packed, (const TT*)bias, (const TT*)act_save, (TT*)y, p); } while (0)
      if (dtype == MV_BF16) MV_W4(bf16); else MV_W4(f16);
#undef MV_W4
      MV_LAUNCH_CHECK();
      return MV_OK;
    }
    // 128-row outputs of the big 3x3 layers (the data gradient of 128->256): 2 waves per workgroup, each with the 64 x 128 tile;
    // 32-channel chunks keep the tile at 37 KB, so FOUR workgroups share a CU (445 -> 425 us; 64 input channels measured slower).
    // MV_DCONV_W2 = smallest Cin that takes this path (0: never)
    static int w2 = -1;
    if (w2 < 0) { const char* e = getenv("MV_DCONV_W2"); w2 = e ? atoi(e) : 128; }
    if (w2 && kh == 3 && Cin % 32 == 0 && Cin >= w2 && Cout == 128 && W >= 128 && force != 0) {
      p.cchunk = 32;
      const size_t xb = (size_t)kh * (128 + (kw - 1) * dil_w) * lds_row_stride(32 * 2, 2);
      const size_t ob = (size_t)128 * (128 * 2 + 16);
      const size_t ldsb = xb > ob ? xb : ob;
      dim3 grid(cdiv(W, 128), 1, B * H);
#define MV_W2(TT) do { auto kern = dconv_cl_wide_kernel<TT, 1, 4, 8, 32, 2>; static size_t lds_set_2 = 0; \
        if (ldsb > lds_set_2) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); lds_set_2 = ldsb; } \
        hipLaunchKernelGGL(kern, grid, dim3(128), ldsb, (hipStream_t)stream, (const TT*)x, (const TT*)packed, (const TT*)bias, (const TT*)act_save, (TT*)y, p); } while (0)
      if (dtype == MV_BF16) MV_W2(bf16); else MV_W2(f16);
#undef MV_W2
      MV_LAUNCH_CHECK();
      return MV_OK;
    }
    const int npos = nb8 ? 256 : 128;
    p.cchunk = nb8 ? 64 : (Cin > 128 ? 128 : Cin);
    if (use && Cin % p.cchunk == 0 && (p.cchunk == 128 || p.cchunk == 64 || p.cchunk == 32)) {
      const size_t xb = (size_t)kh * (npos + (kw - 1) * dil_w) * lds_row_stride(p.cchunk * 2, 2);
      int rows = (big3x3 || (nb8 && Cout >= 256)) ? 256 : Cout >= 128 ? 128 : (Cout >= 64 ? 64 : 32);
      // short sequences (the conditioning producers: a few hundred positions in all): narrower row tiles until the grid covers the chip
      if (!nb8 && !big3x3)
        while (rows > 32 && (long)cdiv(W, npos) * cdiv(Cout, rows) * B * H < 256) rows >>= 1;
      const size_t ob = (size_t)npos * (rows * 2 + 16);
      const size_t ldsb = xb > ob ? xb : ob;
      if (ldsb <= 160 * 1024) {
        hipStream_t st_ = (hipStream_t)stream;
        dim3 grid(cdiv(W, npos), cdiv(Cout, rows), B * H);
#define MV_WIDE(TT, PS_, MW_, NB_, CC_) do { \
          auto kern = dconv_cl_wide_kernel<TT, PS_, MW_, NB_, CC_>; \
          static size_t lds_set_w = 0; \
          if (ldsb > lds_set_w) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); lds_set_w = ldsb; } \
          hipLaunchKernelGGL(kern, grid, dim3(512), ldsb, st_, (const TT*)x, (const TT*)packed, (const TT*)bias, (const TT*)act_save, (TT*)y, p); } while (0)
#define MV_WIDE_R(TT, CC_) do { if (rows == 256) MV_WIDE(TT, 2, 4, 4, CC_); else if (rows == 128) MV_WIDE(TT, 2, 2, 4, CC_); else if (rows == 64) MV_WIDE(TT, 4, 2, 2, CC_); else MV_WIDE(TT, 4, 1, 2, CC_); } while (0)
#define MV_WIDE_T(TT) do { if (nb8) { if (rows == 256) MV_WIDE(TT, 2, 4, 8, 64); else MV_WIDE(TT, 2, 2, 8, 64); } \
          else if (p.cchunk == 128) MV_WIDE_R(TT, 128); else if (p.cchunk == 64) MV_WIDE_R(TT, 64); else MV_WIDE_R(TT, 32); } while (0)
        if (dtype == MV_BF16) MV_WIDE_T(bf16); else MV_WIDE_T(f16);
#undef MV_WIDE_T
#undef MV_WIDE_R
#undef MV_WIDE
        MV_LAUNCH_CHECK();
        return MV_OK;
      }
    }
    p.cchunk = Cin;
  }
  MV_DISPATCH(dtype, {
    hipStream_t s_ = (hipStream_t)stream;
    static int m4 = -1;
    if (m4 < 0) { const char* e = getenv("MV_DCONV_M4"); m4 = e ? atoi(e) : 1; }
    if (Cout >= 256 && m4) {  // 4 waves x 64 rows: one B fragment feeds 4 MFMAs (LDS reads per MFMA halved), two workgroups per CU
      rc = dconv_launch<T, 4, 4, 8>(x, packed, bias, act_save, y, p, s_);
    } else if (Cout >= 256) {        // 8 waves cover all 256 rows: the x tile is staged once per column block
      rc = dconv_launch<T, 8, 2, 8>(x, packed, bias, act_save, y, p, s_);
      if (rc == MV_ERR_UNSUPPORTED) rc = dconv_launch<T, 8, 2, 4>(x, packed, bias, act_save, y, p, s_);
    } else if (Cout >= 128) {
      rc = dconv_launch<T, 4, 2, 8>(x, packed, bias, act_save, y, p, s_);
      if (rc == MV_ERR_UNSUPPORTED) rc = dconv_launch<T, 4, 2, 4>(x, packed, bias, act_save, y, p, s_);
    } else {
      rc = dconv_launch<T, 4, 1, 8>(x, packed, bias, act_save, y, p, s_);
      if (rc == MV_ERR_UNSUPPORTED) rc = dconv_launch<T, 4, 1, 4>(x, packed, bias, act_save, y, p, s_);
    }
  });
  if (rc != MV_OK) return rc;
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_dconv_cl_fwd(const void* x, const void* packed, const void* bias, const void* act_save, void* y,
                               int B, int H, int W, int Cin, int Cout, int kh, int kw, int dil_w, int act, float slope, int dtype,
                               void* stream) {
  return dconv_cl_fwd_impl(x, packed, bias, act_save, y, B, H, W, Cin, Cout, kh, kw, dil_w, act, slope, dtype, stream, nullptr, nullptr, 0);
}

extern "C" int mv_dconv_cl_fwd_head(const void* x, const void* packed, const void* bias, void* y, const void* head_packed, float* head_ws,
                                    int head_kh, int head_kw, int B, int H, int W, int Cin, int Cout, int kh, int kw, int act, float slope,
                                    int dtype, void* stream) {
  MV_CHECK_ARG(head_packed && head_ws && head_kh > 0 && head_kw > 0 && head_kh * head_kw <= 16 && ((uintptr_t)head_packed & 15) == 0);
  return dconv_cl_fwd_impl(x, packed, bias, nullptr, y, B, H, W, Cin, Cout, kh, kw, 1, act, slope, dtype, stream, head_packed, head_ws,
                           head_kh * head_kw);
}

extern "C" size_t mv_dhead_packed_bytes(int C, int dtype) { return (size_t)(C / 32 + C / 16) * 512 * (dtype == MV_F32 ? 4 : 2); }
extern "C" size_t mv_dhead_workspace_bytes(int B, int H, int W) { return sizeof(float) * 16 * (size_t)B * H * W; }

extern "C" int mv_dhead_pack(const void* w, int param_dtype, void* packed, int C, int kh, int kw, int dtype, void* stream) {
  MV_CHECK_ARG(w && packed && C == DH_C && kh * kw <= 16);
  if (dtype == MV_F32) return MV_ERR_UNSUPPORTED;
  MV_DISPATCH(dtype, {
    T* pf = (T*)packed; T* pd = pf + (C / 32) * 512;
    switch (param_dtype) {
      case MV_F32: hipLaunchKernelGGL((dhead_pack_kernel<T, float>), dim3(24), dim3(256), 0, (hipStream_t)stream, (const float*)w, pf, pd, C, kh * kw); break;
      case MV_BF16: hipLaunchKernelGGL((dhead_pack_kernel<T, bf16>), dim3(24), dim3(256), 0, (hipStream_t)stream, (const bf16*)w, pf, pd, C, kh * kw); break;
      case MV_F16: hipLaunchKernelGGL((dhead_pack_kernel<T, f16>), dim3(24), dim3(256), 0, (hipStream_t)stream, (const f16*)w, pf, pd, C, kh * kw); break;
      default: return MV_ERR_DTYPE;
    }
  });
  MV_LAUNCH_CHECK();
  return MV_OK;
}

template <typename T>
static void dhead_fwd_launch(const void* x, const void* packed, const void* bias, float* ws, void* y, int B, int H, int W, int kh,
                             int kw, hipStream_t s) {
  const long npos = (long)B * H * W;
  const long groups = (npos + 15) / 16;
  const int grid = (int)((groups + 3) / 4 > 4096 ? 4096 : (groups + 3) / 4);
  hipLaunchKernelGGL((dhead_z_kernel<T, DH_C>), dim3(grid), dim3(256), 0, s, (const T*)x, (const T*)packed, ws, npos, kh * kw);
  hipLaunchKernelGGL(dhead_sum_kernel<T>, dim3((unsigned)((npos + 255) / 256 > 4096 ? 4096 : (npos + 255) / 256)), dim3(256), 0, s,
                     ws, (const T*)bias, (T*)y, B, H, W, kh, kw, 0);
}

extern "C" int mv_dhead_fwd(const void* x, const void* packed, const void* bias, float* workspace, void* y, int B, int H, int W,
                            int C, int kh, int kw, int dtype, void* stream) {
  MV_CHECK_ARG(x && packed && workspace && y && B > 0 && H > 0 && W > 0 && C == DH_C && kh * kw <= 16 && ((uintptr_t)x & 15) == 0);
  if (dtype == MV_BF16) dhead_fwd_launch<bf16>(x, packed, bias, workspace, y, B, H, W, kh, kw, (hipStream_t)stream);
  else if (dtype == MV_F16) dhead_fwd_launch<f16>(x, packed, bias, workspace, y, B, H, W, kh, kw, (hipStream_t)stream);
  else return MV_ERR_UNSUPPORTED;
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_dhead_sum(const float* workspace, const void* bias, void* y, int B, int H, int W, int kh, int kw, int dtype, void* stream) {
  MV_CHECK_ARG(workspace && y && B > 0 && H > 0 && W > 0 && kh * kw <= 16);
  const long npos = (long)B * H * W;
  const unsigned grid = (unsigned)((npos + 255) / 256 > 4096 ? 4096 : (npos + 255) / 256);
  if (dtype == MV_BF16) hipLaunchKernelGGL(dhead_sum_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, workspace, (const bf16*)bias, (bf16*)y, B, H, W, kh, kw, 0);
  else if (dtype == MV_F16) hipLaunchKernelGGL(dhead_sum_kernel<f16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, workspace, (const f16*)bias, (f16*)y, B, H, W, kh, kw, 0);
  else return MV_ERR_UNSUPPORTED;
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_dhead_dgrad(const void* g, const void* packed, const void* xsave, void* gx, int B, int H, int W, int C, int kh,
                              int kw, float slope, int dtype, void* stream) {
  MV_CHECK_ARG(g && packed && xsave && gx && B > 0 && H > 0 && W > 0 && C == DH_C && kh * kw <= 16);
  MV_CHECK_ARG(((uintptr_t)xsave & 15) == 0 && ((uintptr_t)gx & 15) == 0);
  const long groups = ((long)B * H * W + 15) / 16;
  const int grid = (int)((groups + 3) / 4 > 4096 ? 4096 : (groups + 3) / 4);
  if (dtype == MV_BF16)
    hipLaunchKernelGGL(dhead_dgrad_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)g,
                       (const bf16*)packed + (C / 32) * 512, (const bf16*)xsave, (bf16*)gx, B, H, W, kh, kw, slope);
  else if (dtype == MV_F16)
    hipLaunchKernelGGL(dhead_dgrad_kernel<f16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16*)g,
                       (const f16*)packed + (C / 32) * 512, (const f16*)xsave, (f16*)gx, B, H, W, kh, kw, slope);
  else return MV_ERR_UNSUPPORTED;
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_dhead_wgrad(const void* g, const void* x, float* gw, float* gb, int B, int H, int W, int C, int kh, int kw,
                              int dtype, void* stream) {
  MV_CHECK_ARG(g && x && gw && gb && B > 0 && B <= 65535 && H > 0 && W > 0 && C > 0 && C <= 1024 && C % 64 == 0);
  MV_HIP(mvi_zero_async(gw, sizeof(float) * (size_t)kh * kw * C, (hipStream_t)stream));
  MV_HIP(mvi_zero_async(gb, sizeof(float), (hipStream_t)stream));
  MV_CHECK_ARG(kw <= DH_MAXTAPS && H <= 65535);
  if (dtype != MV_F32 && ((uintptr_t)x & 15) == 0) {   // 16-byte channel vectors: register-tiled reduction
    int rc = MV_ERR_UNSUPPORTED;
    if (dtype == MV_BF16) rc = dtap_launch<bf16>(x, g, gw, gb, B, H, W, C, kh, kw, 1, 1, C, 2, (hipStream_t)stream);
    else if (dtype == MV_F16) rc = dtap_launch<f16>(x, g, gw, gb, B, H, W, C, kh, kw, 1, 1, C, 2, (hipStream_t)stream);
    if (rc == MV_OK) { MV_LAUNCH_CHECK(); return MV_OK; }
  }
  const int chunk = 256;
  dim3 grid(cdiv(W, chunk), H, B);
  const size_t lds = sizeof(float) * (size_t)kh * (chunk + kw - 1);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(dhead_wgrad_kernel<T>, grid, dim3(256), lds, (hipStream_t)stream, (const T*)g,
                                        (const T*)x, gw, gb, H, W, C, kh, kw, chunk));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" size_t mv_dconv_wgrad_workspace_bytes(int Cin, int Cout, int kh, int kw) {
  return sizeof(float) * (size_t)Cin * Cout * kh * kw;
}

template <typename T, int TH, int TW_, int WT>
static int dwgrad_launch(const void* x, const void* g, float* gws, float* gb, int B, int H, int W, int Cin, int Cout, int dil, hipStream_t s,
                         int xW = 0, long sample_stride = 0) {
  const int wsplit = cdiv(W, WT);
  const int halo = (TW_ - 1) * dil;
  if (halo > (TH == 1 ? 64 : TW_ - 1)) return MV_ERR_UNSUPPORTED;
  const size_t lds = 2 * (size_t)(WT + TH * (WT + halo)) * 160;
  if (lds > 160 * 1024) return MV_ERR_UNSUPPORTED;
  auto kern = dconv_wgrad_kernel<T, TH, TW_, WT>;
  static size_t lds_set = 0;
  if (lds > lds_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); lds_set = lds; }
  const long nchunks = (long)B * H * wsplit;
  if (nchunks > (1L << 30)) return MV_ERR_UNSUPPORTED;
  const int tiles = cdiv(Cin, 64) * cdiv(Cout, 64);
  // one workgroup per CU is resident (LDS); one round of workgroups keeps the final fp32 atomics to a few tens of MB
  static int target = 0;
  if (!target) { const char* e = getenv("MV_WGRAD_WGS"); target = e ? atoi(e) : 256; if (target < 1) target = 256; }
  int groups = target / tiles; if (groups < 1) groups = 1; if (groups > nchunks) groups = (int)nchunks;
  int cpw = (int)((nchunks + groups - 1) / groups);
  if (sample_stride > 0) cpw = H * wsplit;            // one workgroup per (tile, sample)
  groups = (int)((nchunks + cpw - 1) / cpw);
  if (groups > 65535) return MV_ERR_UNSUPPORTED;
  dim3 grid(cdiv(Cin, 64), cdiv(Cout, 64), groups);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, (const T*)x, (const T*)g, gws, gb, B, H, W, Cin, Cout, wsplit, (int)nchunks, cpw, dil,
                     xW > 0 ? xW : W, sample_stride);
  return MV_OK;
}

#ifdef MV_DC_TIMING
static void w3_dump(int mo, dim3 grid, hipStream_t st) {
  static long long* dbg = nullptr;
  static int calls = 0;
  if (!dbg) { (void)hipMalloc(&dbg, 4096 * 8 * 8); (void)hipMemcpyToSymbol(HIP_SYMBOL(dc_dbg), &dbg, sizeof(dbg)); return; }
  if (++calls % 13 != 5) return;
  (void)hipStreamSynchronize(st);
  static long long hbuf[4096 * 8];
  (void)hipMemcpy(hbuf, dbg, sizeof(hbuf), hipMemcpyDeviceToHost);
  const long nwg = (long)grid.x * grid.y * grid.z < 4096 ? (long)grid.x * grid.y * grid.z : 4096;
  double avg[8] = {0}; int cnt[8] = {0};
  for (long w = 0; w < nwg; ++w) for (int i = 0; i < 8; ++i) { long long v = hbuf[w * 8 + i]; if (v >= 0) { avg[i] += (double)v; cnt[i]++; } }
  fprintf(stderr, "[wgrad3 timing] MO %d grid %u x %u x %u per chunk (issue+toff, bias, mfma, commit, barrier, chunks):", mo, grid.x, grid.y, grid.z);
  for (int i = 0; i < 8; ++i) if (cnt[i]) fprintf(stderr, " %.0f", avg[i] / cnt[i]);
  fprintf(stderr, "\n");
}
#endif
template <typename T, int MO>
static int dwgrad3_launch(const void* x, const void* g, float* gws, float* gb, int B, int H, int W, int Cin, int Cout, hipStream_t s) {
  constexpr int WT = 64, OT = 32 * MO;
  const int wsplit = cdiv(W, WT);
  const size_t lds = 2 * (size_t)WT * (OT * 2 + 32) + 9 * (size_t)(WT + 2) * 160;
  auto kern = dconv_wgrad3_kernel<T, MO>;
  static bool lds_set = false;
  if (!lds_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); lds_set = true; }
  const long nchunks = (long)B * wsplit * H;               // chunk = (sample, strip, row), rows fastest
  if (nchunks > (1L << 30) || (long)B * wsplit * (H + 2) > (1L << 30)) return MV_ERR_UNSUPPORTED;
  const int tiles = cdiv(Cin, 64) * cdiv(Cout, OT);
  static int target = 0;
  if (!target) { const char* e = getenv("MV_WGRAD_WGS"); target = e ? atoi(e) : 256; if (target < 1) target = 256; }
  int groups = target / tiles; if (groups < 1) groups = 1; if (groups > nchunks) groups = (int)nchunks;
  const int cpw = (int)((nchunks + groups - 1) / groups);
  groups = (int)((nchunks + cpw - 1) / cpw);
  if (groups > 65535) return MV_ERR_UNSUPPORTED;
  dim3 grid(cdiv(Cin, 64), cdiv(Cout, OT), groups);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, (const T*)x, (const T*)g, gws, gb, B, H, W, Cin, Cout, wsplit, (int)nchunks, cpw);
#ifdef MV_DC_TIMING
  w3_dump(MO, grid, s);
#endif
  return MV_OK;
}

template <typename T>
static int dwgrad_dispatch(const void* x, const void* g, float* ws, float* gb, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                           int dil, hipStream_t s) {
  if (kh == 3 && kw == 3 && dil == 1) {
    static int v3 = -1;
    if (v3 < 0) { const char* e = getenv("MV_WGRAD3"); v3 = e ? atoi(e) : 1; }
    if (v3) return Cout >= 128 ? dwgrad3_launch<T, 4>(x, g, ws, gb, B, H, W, Cin, Cout, s) : dwgrad3_launch<T, 2>(x, g, ws, gb, B, H, W, Cin, Cout, s);
  }
  if (kh == 3 && kw == 3) return dil == 1 ? dwgrad_launch<T, 3, 3, 64>(x, g, ws, gb, B, H, W, Cin, Cout, 1, s) : MV_ERR_UNSUPPORTED;
  if (kh != 1) return MV_ERR_UNSUPPORTED;
  switch (kw) {
    case 1: return dwgrad_launch<T, 1, 1, 128>(x, g, ws, gb, B, H, W, Cin, Cout, dil, s);
    case 3: return dwgrad_launch<T, 1, 3, 128>(x, g, ws, gb, B, H, W, Cin, Cout, dil, s);
    case 5: return dwgrad_launch<T, 1, 5, 128>(x, g, ws, gb, B, H, W, Cin, Cout, dil, s);
    case 7: return dwgrad_launch<T, 1, 7, 128>(x, g, ws, gb, B, H, W, Cin, Cout, dil, s);
    case 11: return dwgrad_launch<T, 1, 11, 128>(x, g, ws, gb, B, H, W, Cin, Cout, dil, s);
    case 15: return dwgrad_launch<T, 1, 15, 128>(x, g, ws, gb, B, H, W, Cin, Cout, dil, s);
    default: return MV_ERR_UNSUPPORTED;
  }
}

extern "C" int mv_dconv_wgrad_cl(const void* x, const void* g, float* gw, float* gb, float* workspace, int B, int H, int W, int Cin,
                                 int Cout, int kh, int kw, int dil_w, int dtype, void* stream) {
  MV_CHECK_ARG(x && g && gw && workspace && B > 0 && H > 0 && W > 0 && Cin % 8 == 0 && Cout % 8 == 0 && dil_w >= 1);
  MV_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)g & 15) == 0);
  if (dtype == MV_F32) return MV_ERR_UNSUPPORTED;   // 16-bit storage only (transposed LDS reads); callers fall back to the generic kernel
  hipStream_t s = (hipStream_t)stream;
  const size_t wn = (size_t)Cout * Cin * kh * kw;
  if (gb && gb == workspace + wn) {                  // bias sums placed right behind the workspace: one fill instead of two
    MV_HIP(mvi_zero_async(workspace, sizeof(float) * (wn + Cout), s));
  } else {
    MV_HIP(mvi_zero_async(workspace, sizeof(float) * wn, s));
    if (gb) MV_HIP(mvi_zero_async(gb, sizeof(float) * (size_t)Cout, s));
  }
  int rc = MV_ERR_UNSUPPORTED;
  if (dtype == MV_BF16) rc = dwgrad_dispatch<bf16>(x, g, workspace, gb, B, H, W, Cin, Cout, kh, kw, dil_w, s);
  else if (dtype == MV_F16) rc = dwgrad_dispatch<f16>(x, g, workspace, gb, B, H, W, Cin, Cout, kh, kw, dil_w, s);
  if (rc != MV_OK) return rc;
  const long total = (long)Cout * Cin * kh * kw;
  hipLaunchKernelGGL(dconv_wgrad_reorder_kernel, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256),
                     0, s, workspace, gw, Cout * Cin, kh * kw);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_dconv_wgrad_cl_pz(const void* x, const void* g, float* gw, float* gb, float* workspace, int B, int H, int W, int Cin,
                                    int Cout, int kh, int kw, int dil_w, int dtype, void* stream) {
  MV_CHECK_ARG(x && g && gw && workspace && B > 0 && H > 0 && W > 0 && Cin % 8 == 0 && Cout % 8 == 0 && dil_w >= 1);
  MV_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)g & 15) == 0);
  if (dtype == MV_F32) return MV_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)Cout * Cin * kh * kw;
  int rc = MV_ERR_UNSUPPORTED;
  if (dtype == MV_BF16) rc = dwgrad_dispatch<bf16>(x, g, workspace, workspace + total, B, H, W, Cin, Cout, kh, kw, dil_w, s);
  else if (dtype == MV_F16) rc = dwgrad_dispatch<f16>(x, g, workspace, workspace + total, B, H, W, Cin, Cout, kh, kw, dil_w, s);
  if (rc != MV_OK) return rc;
  hipLaunchKernelGGL(dconv_wgrad_finalize_kernel, dim3((unsigned)((total + Cout + 255) / 256 > 2048 ? 2048 : (total + Cout + 255) / 256)),
                     dim3(256), 0, s, workspace, gw, gb, Cout * Cin, kh * kw, Cout);
  MV_LAUNCH_CHECK();
  return MV_OK;
}


// ------------------------------------------------------------------------------------------------ tap matrix (im2col of a one-channel map)
// out[pos][tap] (16 taps per position, zero beyond kh*kw) = sc[pos + off(tap)]  (flip = 0)  or  sc[pos - off(tap)]  (flip = 1),
// off(tap) = (ih - kh/2, iw - kw/2), zero outside the image.  With it the one-channel weight gradients become plain 1x1
// weight-gradient GEMMs on the MFMA kernel:  head  gw[tap][c] = sum_pos out[pos][tap] x[pos][c]  (flip = 1, sc = output
// gradient; column kh*kw/2 sums to the bias gradient),  first layer  gw[o][tap] = sum_pos g1[pos][o] out[pos][tap]  (flip = 0).
template <typename T>
__global__ __launch_bounds__(256) void tap_matrix_kernel(const T* __restrict__ sc, T* __restrict__ out, int B, int H, int W, int kh,
                                                         int kw, int flip) {
  const long npos = (long)B * H * W;
  const int ph = kh / 2, pw = kw / 2, taps = kh * kw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npos * 2; i += (long)gridDim.x * blockDim.x) {
    const long pos = i >> 1;
    const int half = (int)(i & 1);
    const int w = (int)(pos % W), h = (int)((pos / W) % H);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tap = 8 * half + j;
      const int ih = tap / kw, iw = tap - ih * kw;
      const int dh = flip ? -(ih - ph) : ih - ph, dw = flip ? -(iw - pw) : iw - pw;
      const bool valid = tap < taps && h + dh >= 0 && h + dh < H && w + dw >= 0 && w + dw < W;
      const float raw = ld<T>(sc + (valid ? pos + (long)dh * W + dw : pos));
      v[j] = valid ? raw : 0.f;
    }
    alignas(16) T o8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) st<T>(o8 + j, v[j]);
    *reinterpret_cast<u32x4*>(out + pos * 16 + half * 8) = *reinterpret_cast<u32x4*>(o8);
  }
}

extern "C" int mv_tap_matrix(const void* sc, void* out, int B, int H, int W, int kh, int kw, int flip, int dtype, void* stream) {
  MV_CHECK_ARG(sc && out && B > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && kh * kw <= 16 && ((uintptr_t)out & 15) == 0);
  const long n = (long)B * H * W * 2;
  const unsigned grid = (unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(tap_matrix_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)sc, (T*)out, B, H, W,
                                        kh, kw, flip));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

// ------------------------------------------------------------------------------------------------ ODConvTranspose1d bank gradients
// gW[k][c][o][j] = sum_b alpha[b,k] sum_t x[b,t,c] g[b, t*s + j - p, o]   and   d alpha[b,k] = <per-sample gradient, W[k]>
// (odconv.py:172-205, ks = 2*stride).  With the time-padded gradient gp [B][Tin+1][s*Cout] (row q, channel r*Cout+o holds
// g[q*s + r - p][o]) the per-sample gradient is the two-tap weight gradient  tile[b][q][c][(r,o)] = sum_t x[t][c] gp[t+q][(r,o)]:
// one MFMA GEMM per sample (dconv_wgrad_kernel with per-sample output), then the existing alpha-chain reduction against the
// banks in the same tap-major layout, and the inverse layout map back to [K][Cin][Cout][ks].
template <typename T>
__global__ __launch_bounds__(256) void odconvT_tapmajor_kernel(const T* __restrict__ w, T* __restrict__ out, long n, int Cin, int Cout, int ks,
                                                               int stride) {
  // out[k][q][c][r*Cout + o] = w[k][c][o][q*stride + r]
  const long sc = (long)stride * Cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ro = (int)(i % sc);
    const long t1 = i / sc;
    const int c = (int)(t1 % Cin);
    const long kq = t1 / Cin;
    const int q = (int)(kq % 2);
    const long k = kq / 2;
    const int r = ro / Cout, o = ro - r * Cout;
    out[i] = w[((k * Cin + c) * Cout + o) * ks + q * stride + r];
  }
}
__global__ __launch_bounds__(256) void odconvT_tapmajor_inv_kernel(const float* __restrict__ gtm, float* __restrict__ gw, long n, int Cin, int Cout,
                                                                   int ks, int stride) {
  // gw[k][c][o][j] = gtm[k][j / stride][c][(j % stride)*Cout + o]
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % ks);
    const long t1 = i / ks;
    const int o = (int)(t1 % Cout);
    const long kc = t1 / Cout;
    const int c = (int)(kc % Cin);
    const long k = kc / Cin;
    const int q = j / stride, r = j - q * stride;
    gw[i] = gtm[((k * 2 + q) * Cin + c) * ((long)stride * Cout) + (long)r * Cout + o];
  }
}

extern "C" int mv_odconv_wgrad_reduce(const float* gws, const void* w, const float* alpha, float* gw, float* galpha, int B, int K,
                                      long nelem, int dtype, void* stream);

extern "C" size_t mv_odconvT_wgrad_workspace_bytes(int B, int Cin, int Cout, int ks, int K, int dtype) {
  const size_t n = (size_t)Cin * Cout * ks;                    // elements of one bank == one per-sample tile
  return sizeof(float) * n * B + (dtype == MV_F32 ? 4 : 2) * n * K + sizeof(float) * n * K + 1024;
}

extern "C" int mv_odconvT_wgrad_mfma(const void* x_cl, const void* gp, const void* w, const float* alpha, float* gw, float* galpha,
                                     void* workspace, int B, int Tin, int Cin, int Cout, int ks, int stride, int K, int dtype,
                                     void* stream) {
  MV_CHECK_ARG(x_cl && gp && w && alpha && gw && galpha && workspace && B > 0 && Tin > 0 && K >= 1 && K <= 8 && ks == 2 * stride);
  MV_CHECK_ARG(Cin % 8 == 0 && (stride * Cout) % 8 == 0 && ((uintptr_t)x_cl & 15) == 0 && ((uintptr_t)gp & 15) == 0);
  if (dtype == MV_F32) return MV_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const long n = (long)Cin * Cout * ks;
  float* tiles = (float*)workspace;                                           // [B][2][Cin][s*Cout]
  char* wtm = (char*)(tiles + n * B);                                         // [K][2][Cin][s*Cout] in `dtype`
  float* gtm = (float*)(wtm + ((size_t)2 * n * K + 255) / 256 * 256);         // [K][2][Cin][s*Cout] fp32
  MV_HIP(mvi_zero_async(tiles, sizeof(float) * (size_t)n * B, s));
  // "g" operand = x (rows = input channels), "x" operand = padded gradient rows (columns = (r, o)), taps q = 0, 1
  int rc;
  if (dtype == MV_BF16) rc = dwgrad_launch<bf16, 1, 2, 128>(gp, x_cl, tiles, nullptr, B, 1, Tin, stride * Cout, Cin, 1, s, Tin + 1, n);
  else rc = dwgrad_launch<f16, 1, 2, 128>(gp, x_cl, tiles, nullptr, B, 1, Tin, stride * Cout, Cin, 1, s, Tin + 1, n);
  if (rc != MV_OK) return rc;
  const int g1 = (int)((n * K + 255) / 256 > 4096 ? 4096 : (n * K + 255) / 256);
  if (dtype == MV_BF16) hipLaunchKernelGGL(odconvT_tapmajor_kernel<bf16>, dim3(g1), dim3(256), 0, s, (const bf16*)w, (bf16*)wtm, n * K, Cin, Cout, ks, stride);
  else hipLaunchKernelGGL(odconvT_tapmajor_kernel<f16>, dim3(g1), dim3(256), 0, s, (const f16*)w, (f16*)wtm, n * K, Cin, Cout, ks, stride);
  rc = mv_odconv_wgrad_reduce(tiles, wtm, alpha, gtm, galpha, B, K, n, dtype, stream);
  if (rc != MV_OK) return rc;
  hipLaunchKernelGGL(odconvT_tapmajor_inv_kernel, dim3(g1), dim3(256), 0, s, gtm, gw, n * K, Cin, Cout, ks, stride);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_dfirst_fwd_cl(const void* x0, const void* w, const void* bias, void* y, int B, int H, int W, int C1, int kh,
                                int kw, float slope, int dtype, void* stream) {
  MV_CHECK_ARG(x0 && w && y && B > 0 && B <= 65535 && H > 0 && W > 0 && C1 > 0);
  const bool tiled = C1 % 8 == 0 && C1 <= 2048 && 256 % (C1 / 8) == 0 && ((uintptr_t)y & 15) == 0 &&
                     ((kh == 3 && kw == 3) || (kh == 1 && kw == 15)) && dtype != MV_F32;
  if (tiled) {
    const int WT = 1024, wsplit = cdiv(W, WT);
    const long nchunks = (long)B * H * wsplit;
    if (nchunks <= (1L << 30)) {
      const size_t lds = sizeof(float) * (size_t)kh * (WT + kw - 1);
      hipStream_t s = (hipStream_t)stream;
#define MV_DF(TT) do { \
        if (kh == 3) hipLaunchKernelGGL((dfirst_fwd_tiled_kernel<TT, 3, 3>), dim3((unsigned)nchunks), dim3(256), lds, s, (const TT*)x0, (const TT*)w, (const TT*)bias, (TT*)y, H, W, C1, slope, WT, wsplit); \
        else hipLaunchKernelGGL((dfirst_fwd_tiled_kernel<TT, 1, 15>), dim3((unsigned)nchunks), dim3(256), lds, s, (const TT*)x0, (const TT*)w, (const TT*)bias, (TT*)y, H, W, C1, slope, WT, wsplit); } while (0)
      if (dtype == MV_BF16) MV_DF(bf16); else MV_DF(f16);
#undef MV_DF
      MV_LAUNCH_CHECK();
      return MV_OK;
    }
  }
  const long n = (long)H * W * C1;
  dim3 grid((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256), B);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(dfirst_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x0, (const T*)w,
                                        (const T*)bias, (T*)y, H, W, C1, kh, kw, slope));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" size_t mv_dfirst_dgrad_workspace_bytes(int B, int H, int W) { return sizeof(float) * 16 * (size_t)B * H * W + 4096; }

template <typename T>
static void dfirst_dgrad_mfma(const void* g1, const void* w, void* gx0, float* ws, int B, int H, int W, int kh, int kw, hipStream_t s) {
  const long npos = (long)B * H * W;
  T* pk = reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + sizeof(float) * 16 * (size_t)npos);   // 1 KB packed operator behind z
  hipLaunchKernelGGL(dfirst_dpack_kernel<T>, dim3(2), dim3(256), 0, s, (const T*)w, pk, 32, kh * kw);
  const long groups = (npos + 15) / 16;
  const int grid = (int)((groups + 3) / 4 > 4096 ? 4096 : (groups + 3) / 4);
  hipLaunchKernelGGL((dhead_z_kernel<T, 32>), dim3(grid), dim3(256), 0, s, (const T*)g1, (const T*)pk, ws, npos, kh * kw);
  hipLaunchKernelGGL(dhead_sum_kernel<T>, dim3((unsigned)((npos + 255) / 256 > 4096 ? 4096 : (npos + 255) / 256)), dim3(256), 0, s,
                     ws, (const T*)nullptr, (T*)gx0, B, H, W, kh, kw, 1);
}

extern "C" int mv_dfirst_dgrad_cl(const void* g1, const void* w, void* gx0, float* workspace, int B, int H, int W, int C1, int kh,
                                  int kw, int dtype, void* stream) {
  MV_CHECK_ARG(g1 && w && gx0 && B > 0 && B <= 65535 && H > 0 && W > 0 && C1 > 0);
  if (workspace && C1 == 32 && kh * kw <= 16 && dtype != MV_F32 && ((uintptr_t)g1 & 15) == 0) {
    // taps on the MFMA rows: z[tap][pos] = sum_o w[o][tap] g1[pos][o], then gx0[pos] = sum_tap z[tap][pos - off(tap)]
    if (dtype == MV_BF16) dfirst_dgrad_mfma<bf16>(g1, w, gx0, workspace, B, H, W, kh, kw, (hipStream_t)stream);
    else dfirst_dgrad_mfma<f16>(g1, w, gx0, workspace, B, H, W, kh, kw, (hipStream_t)stream);
    MV_LAUNCH_CHECK();
    return MV_OK;
  }
  dim3 grid((unsigned)(((long)H * W + 3) / 4), B);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(dfirst_dgrad_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)g1, (const T*)w,
                                        (T*)gx0, H, W, C1, kh, kw));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_dfirst_wgrad_cl(const void* g1, const void* x0, float* gw, float* gb, int B, int H, int W, int C1, int kh,
                                  int kw, int dtype, void* stream) {
  MV_CHECK_ARG(g1 && x0 && gw && gb && B > 0 && B <= 65535 && H > 0 && W > 0 && C1 > 0 && 256 % C1 == 0);
  MV_HIP(mvi_zero_async(gw, sizeof(float) * (size_t)C1 * kh * kw, (hipStream_t)stream));
  MV_HIP(mvi_zero_async(gb, sizeof(float) * (size_t)C1, (hipStream_t)stream));
  if (dtype != MV_F32 && ((uintptr_t)g1 & 15) == 0) {
    int rc = MV_ERR_UNSUPPORTED;
    if (dtype == MV_BF16) rc = dtap_launch<bf16>(g1, x0, gw, gb, B, H, W, C1, kh, kw, 0, kh * kw, 1, 1, (hipStream_t)stream);
    else if (dtype == MV_F16) rc = dtap_launch<f16>(g1, x0, gw, gb, B, H, W, C1, kh, kw, 0, kh * kw, 1, 1, (hipStream_t)stream);
    if (rc == MV_OK) { MV_LAUNCH_CHECK(); return MV_OK; }
  }
  const int chunk = 1024;
  dim3 grid((unsigned)(((long)H * W + chunk - 1) / chunk), B);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(dfirst_wgrad_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)g1, (const T*)x0,
                                        gw, gb, H, W, C1, kh, kw, chunk));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_colsum_cl(const void* x, float* out, long rows, int C, int dtype, void* stream) {
  MV_CHECK_ARG(x && out && rows > 0 && C >= 8 && C <= 2048 && C % 8 == 0 && (256 % (C / 8)) == 0 && ((uintptr_t)x & 15) == 0);
  MV_HIP(mvi_zero_async(out, sizeof(float) * (size_t)C, (hipStream_t)stream));
  if (dtype == MV_F32 && (256 % (C / 4)) != 0) return MV_ERR_UNSUPPORTED;
  const int chunk = 2048;
  MV_DISPATCH(dtype, hipLaunchKernelGGL(colsum_cl_kernel<T>, dim3((unsigned)((rows + chunk - 1) / chunk)), dim3(256), 0,
                                        (hipStream_t)stream, (const T*)x, out, rows, C, chunk));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_take_channel(const void* x, void* out, long rows, int C, int c, int dtype, void* stream) {
  MV_CHECK_ARG(x && out && rows > 0 && C > 0 && c >= 0 && c < C);
  const int grid = (int)((rows + 255) / 256 > 4096 ? 4096 : (rows + 255) / 256);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(take_channel_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)out, rows, C, c));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_put_channel(const void* g, void* y, long rows, int C, int c, int dtype, void* stream) {
  MV_CHECK_ARG(g && y && rows > 0 && C > 0 && c >= 0 && c < C);
  const long n = rows * C;
  const int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  MV_DISPATCH(dtype, hipLaunchKernelGGL(put_channel_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)g, (T*)y, rows, C, c));
  MV_LAUNCH_CHECK();
  return MV_OK;
}
