// Streaming form of the MultiReceptiveFieldBlock chain (the generator's `for blk in mrf_blocks: x = blk(x)`), fp32 storage with
// two-product operands (MV_F32_W16: hi + lo f16 activations x single f16 weights), the reference's dilations (1, 3, 5).
//   reference arithmetic: hifigan_modified/grc_lora.py:32-68 (GRC_LoRA_Block) and :157-163 (MultiReceptiveFieldBlock.forward)
//
// Same pass structure as the tile form's chain (mrf_fused.hip: V = statistics of v, F = stages 1-3 -> f + statistics of f,
// A = x' = a f + b + x written once + statistics of v'), different machinery, built from what the in-kernel marks of the tile form
// showed: its passes move ~10 B/clk per CU while a tile is in flight and nothing while a workgroup
// fills its first 128-256-step tile, stages its weights or sits in one of the two barriers per tile, and its eight waves hit the LDS
// with their stage-1 operand reads at the same moment.  Here
//   * every WAVE owns a contiguous span of L time steps and walks it in 16-step column tiles through a PRIVATE ring of three 16-row
//     slots in LDS - no workgroup barrier between the prologue and the final statistics reduction, so the eight waves drift apart
//     and their LDS / matrix / memory phases interleave by themselves;
//   * pass F reads its input by LDS-DMA (`buffer_load ... lds`, 16 B per lane): the rows never visit a register, two batches are in
//     flight per wave under a counted `s_waitcnt vmcnt`, and rows outside the sample arrive as zeros from the buffer range check
//     (the convolution's zero padding).  For that the residual stream between the blocks is stored PRE-SPLIT ("pair rows"): per time
//     step 8 groups of [8 x f16 hi | 8 x f16 lo] = 256 bytes, the same size as the fp32 row and exactly the LDS operand image, so
//     the split is paid once where the row is produced (pass A) instead of once per consumer;
//   * the pipeline fill of a wave is one 16-row batch (4 KB), not a 35-70 KB tile.
// The ring rows are padded to 272 bytes (bank-conflict-free ds_read_b128 operand reads up to one 2-way pair); a DMA instruction writes
// 1 KB linearly, so lane n of a batch carries chunk n % 17 of row n / 17 (chunk 16 = the pad, a re-read of chunk 15).
#include "mrf_common.h"
#include <cstdlib>
#include <cstdio>

namespace mv {

constexpr int SR_NW = 8;                    // waves per workgroup
constexpr int SR_RS = 272;                  // LDS row stride: 256-byte pair row + 16
constexpr int SR_SLOT = 16 * SR_RS;         // one batch of 16 rows
constexpr int SR_RING = 3 * SR_SLOT;        // per wave
constexpr int SR_H = 5;                     // halo of the dilations (1, 3, 5)
constexpr int SR_WB_ALL = (MRF_CONV_FRAGS + MRF_RES_FRAGS + MRF_FUS_FRAGS) * FRAG_BYTES;   // f16 image (mv_mrf_pack, MV_F16)
constexpr int SR_WB_CONV = MRF_CONV_FRAGS * FRAG_BYTES;
constexpr int SR_ROWB = 256;                // bytes of one row in HBM (fp32 [64] or pair)

// f as hl8 rows as well: measured at C2 it costs more than it saves - the encode (cvt, residual, clamp, cvt: ~5 VALU per value) lands
// in the matrix-bound F passes (+1.3 us each, F0 +3.1) while the A passes and the output conv, which read f, are not bound by bytes any
// more once x' is 192 bytes (A 37 -> 33 us either way).  The code paths stay (final kernel, output conv, tests of the format).
constexpr bool SR_FHL = false;
enum { SM_V0 = 0, SM_F0 = 1, SM_A = 2, SM_F = 3 };
// V0: fp32 rows -> statistics of v.   F0: fp32 rows -> f (fp32) + statistics of f.   F: pair rows (LDS-DMA) -> the same.
// A : f_prev (fp32), x_prev (fp32 or pair rows) -> x' = a f + b + x as pair rows (written once) + statistics of v'.

struct SrArgs {
  const void* x;             // V0 / F0: fp32 rows; F: pair rows; A: x_prev
  const float* fprev;        // A
  void* out;                 // F0 / F: f; A: x' pair rows
  const char* packed;        // this block's weights (f16 image + tables)
  const char* packed_prev;   // A: the previous block's (its GroupNorm(8,64) affine tables)
  const float* part5_in; float* part5_out;
  const float* part8_in; float* part8_out;
  int Tn, L, nwg;
  float eps;
  int dbg;                   // ablation switches of the MV_SR_TIMING build (bit 0: no f stores, bit 1: no stages 2-3, bit 2: no compute)
};

typedef __attribute__((address_space(3))) void lds_void;
#ifdef MV_SR_TIMING
__device__ long long* sr_dbg = nullptr;     // [mode][wave][8]: wait, stage 1, stages 2-3 + stores, fill / dma issue, total, t_start, t_end (100 MHz)
#define SR_TM(slot) do { const long long t_ = clock64(); tacc[slot] += t_ - tlast; tlast = t_; } while (0)
#else
#define SR_TM(slot) do {} while (0)
#endif

// HL: the streams between the blocks (x' and f) are hl8 rows (mfma.h) instead of pair / fp32 rows.  Then F0 writes f as hl8, A reads
// f as hl8 and x as fp32 rows (XPAIR false: block 1) or hl8 rows (XPAIR true) and writes x' as hl8, F reads x' through registers
// (hi plane = the operand image's hi halves as they are, lo bytes widened to f16) and writes f as hl8.  V0 is not affected.
template <int MODE, bool XPAIR, bool HL>
__global__ __launch_bounds__(SR_NW * 64) void mrf_stream_kernel(SrArgs a) {
  using M = Mma<f32w16>;
  using VA = M::VA;
  using VB = M::VB;
  constexpr bool FULL = (MODE == SM_F0 || MODE == SM_F);
  constexpr bool DMA = ((MODE == SM_F) && !HL) || (MODE == SM_V0 && XPAIR);     // pair rows in HBM ARE the LDS operand image
  constexpr bool XHL = HL && (MODE == SM_F || (MODE == SM_A && XPAIR));          // x arrives as hl8 rows
  constexpr bool FHL = HL && SR_FHL;                                             // f as hl8 rows too (off: see SR_FHL)
  constexpr int XRB = XHL ? SR_HLB : SR_ROWB;                                    // row bytes of x (in), of f (in or out), of x' (out)
  constexpr int FRB = FHL ? SR_HLB : SR_ROWB;
  constexpr int ORB = HL ? SR_HLB : SR_ROWB;
  constexpr bool NEED5 = FULL, NEED8 = (MODE == SM_A);
  constexpr int WLB = FULL ? SR_WB_ALL : SR_WB_CONV;

  extern __shared__ __align__(16) char lds[];
  char* wl = lds;
  float* tab = reinterpret_cast<float*>(lds + WLB);
  float* st5 = tab + MRF_TAB_FLOATS;            // [16][2] mean, rstd
  float* st8 = st5 + 32;                        // [8][2]
  float* red = st8 + 16;                        // [SR_NW][16][2]
  float2* sc = reinterpret_cast<float2*>(red + SR_NW * 32);   // 512 partial sums in flight
  char* rings = reinterpret_cast<char*>(sc + 512);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, wg = blockIdx.x;
  const int Tn = a.Tn, nwg = a.nwg;
  const int s0 = (wg * SR_NW + wid) * a.L;                      // first time step of this wave's span
  const int span = min(Tn, s0 + a.L) - s0;
  const int nt = span > 0 ? (span + 15) >> 4 : 0;               // column tiles of this wave
  const int nb = nt > 0 ? nt + 1 : 0;                           // 16-row batches: batch k = rows s0 - 5 + 16 k ..
  const int ring = (int)(rings - lds) + wid * SR_RING;          // LDS byte offset of this wave's ring
#ifdef MV_SR_TIMING
  long long tacc[5] = {0, 0, 0, 0, 0};
  const long long t_abs0 = __builtin_amdgcn_s_memrealtime();
  long long tlast = clock64();
  const long long tfirst = tlast;
#endif

  const size_t sample = (size_t)b * Tn * XRB;                 // of x
  const size_t sample_f = (size_t)b * Tn * FRB;               // of f (in or out)
  const size_t sample_o = (size_t)b * Tn * ORB;               // of x' (out)
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.x)) + sample, 0, Tn * XRB, 0x00020000);

  // ---- LDS-DMA of one batch (pass F): 4 x 1 KB + 16 lanes; out-of-range rows (t < 0, t >= Tn) fail the range check and land as zeros
  int doff[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int n = 64 * q + lane, row = n / 17, ch = n % 17;
    doff[q] = row * SR_ROWB + (ch < 16 ? ch : 15) * 16;
  }
  auto dma = [&](int k, int slot) {
    const int base = (s0 - SR_H + 16 * k) * SR_ROWB;
    const int dst = ring + slot * SR_SLOT;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(lds + dst + q * 1024), 16, base + doff[q], 0, 0, 0);
    if (lane < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(lds + dst + 4096), 16, base + doff[4], 0, 0, 0);
  };
  if constexpr (DMA) {
    if (nb > 0) { dma(0, 0); dma(1, 1); }
    if (nb > 2) dma(2, 2);
  }

  // ---- packed weights + tables into LDS, GroupNorm statistics from the producer's partial sums (fixed-order sums: deterministic)
  {
    constexpr int N16 = (WLB + MRF_TAB_FLOATS * 4) / 16;
    constexpr int PER = (N16 + SR_NW * 64 - 1) / (SR_NW * 64);
    const u32x4* src = reinterpret_cast<const u32x4*>(a.packed);
    u32x4* dst = reinterpret_cast<u32x4*>(lds);
    u32x4 wv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int idx = tid + i * SR_NW * 64;
      idx = idx < N16 ? idx : N16 - 1;
      wv[i] = src[idx < WLB / 16 ? idx : idx + (SR_WB_ALL - WLB) / 16];
    }
    float2 pp = {0.f, 0.f};
    if constexpr (NEED5) {
      const int q = tid & 15;
      for (int i = tid >> 4; i < nwg; i += 32) {
        const float2 v = *reinterpret_cast<const float2*>(a.part5_in + ((size_t)(b * nwg + i) * 16 + q) * 2);
        pp.x += v.x; pp.y += v.y;
      }
    }
    if constexpr (NEED8) {
      const int q = tid & 7;
      for (int i = tid >> 3; i < nwg; i += 64) {
        const float2 v = *reinterpret_cast<const float2*>(a.part8_in + ((size_t)(b * nwg + i) * 8 + q) * 2);
        pp.x += v.x; pp.y += v.y;
      }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + i * SR_NW * 64;
      if (idx < N16) dst[idx] = wv[i];
    }
    if constexpr (NEED5 || NEED8) sc[tid] = pp;
  }
  if constexpr (NEED5 || NEED8) {
    __syncthreads();
    if (NEED5 && tid < 16) {
      float s1 = 0.f, s2 = 0.f;
      for (int i = 0; i < 32; ++i) { s1 += sc[i * 16 + tid].x; s2 += sc[i * 16 + tid].y; }
      const float n = 4.f * (float)Tn, mu = s1 / n;
      st5[tid * 2] = mu;
      st5[tid * 2 + 1] = rsqrtf(fmaxf(s2 / n - mu * mu, 0.f) + a.eps);
    }
    if (NEED8 && tid < 8) {
      float s1 = 0.f, s2 = 0.f;
      for (int i = 0; i < 64; ++i) { s1 += sc[i * 8 + tid].x; s2 += sc[i * 8 + tid].y; }
      const float n = 8.f * (float)Tn, mu = s1 / n;
      st8[tid * 2] = mu;
      st8[tid * 2 + 1] = rsqrtf(fmaxf(s2 / n - mu * mu, 0.f) + a.eps);
    }
  }
  __syncthreads();

  const float* b_conv = tab, *b_res = tab + 64, *b_fus = tab + 128;
  const float* g5 = tab + 192, *be5 = tab + 256;

  // ---- register-staged fill (V0 / F0: fp32 rows are split on the way in; A: x' is formed, split, stored once and kept for stage 1).
  //      item = (row, 8-channel group): 32 contiguous bytes of an fp32 row, the same 32 bytes of a pair row
  const int irow = lane >> 3, cg = lane & 7;
  float ra[8], rb[8];
  __amdgpu_buffer_rsrc_t rf = rx;
  if constexpr (MODE == SM_A) {
    rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.fprev)) + sample_f, 0, Tn * FRB, 0x00020000);
    // deferred GroupNorm(8,64) of the previous block on this lane's 8 channels (one group): a = rstd * gamma, b = beta - mean * a
    const float* tprev = reinterpret_cast<const float*>(a.packed_prev + SR_WB_ALL);
    const float mu = st8[cg * 2], rs = st8[cg * 2 + 1];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ra[e] = rs * tprev[320 + cg * 8 + e];
      rb[e] = tprev[384 + cg * 8 + e] - mu * ra[e];
    }
  }
  u32x4 sx0[2], sx1[2], sf0[2], sf1[2];
  auto fill_issue = [&](int k) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int t = s0 - SR_H + 16 * k + irow + 8 * it;
      if constexpr (XHL) {          // hi plane: 16 bytes of this lane's 8 channels; lo plane: their 8 bytes
        sx0[it] = __builtin_amdgcn_raw_buffer_load_b128(rx, t * SR_HLB + cg * 16, 0, 0);
        const u32x2 l = __builtin_amdgcn_raw_buffer_load_b64(rx, t * SR_HLB + 128 + cg * 8, 0, 0);
        sx1[it] = u32x4{l[0], l[1], 0u, 0u};
      } else {
        const int off = t * SR_ROWB + cg * 32;
        sx0[it] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
        sx1[it] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + 16, 0, 0);
      }
      if constexpr (MODE == SM_A) {
        if constexpr (FHL) {
          sf0[it] = __builtin_amdgcn_raw_buffer_load_b128(rf, t * SR_HLB + cg * 16, 0, 0);
          const u32x2 l = __builtin_amdgcn_raw_buffer_load_b64(rf, t * SR_HLB + 128 + cg * 8, 0, 0);
          sf1[it] = u32x4{l[0], l[1], 0u, 0u};
        } else {
          const int off = t * SR_ROWB + cg * 32;
          sf0[it] = __builtin_amdgcn_raw_buffer_load_b128(rf, off, 0, 0);
          sf1[it] = __builtin_amdgcn_raw_buffer_load_b128(rf, off + 16, 0, 0);
        }
      }
    }
  };
  auto fill_commit = [&](int k, int slot) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int r = irow + 8 * it;
      const int t = s0 - SR_H + 16 * k + r;
      float v[8];
      u32x4 uh, ul;
      if constexpr (MODE == SM_F && HL) {
        // x' as hl8: the hi plane is the operand image's hi half as it is, the lo bytes widen to f16 exactly (no fp32 round trip)
        uh = sx0[it];
        ul = hl8_lo_f16(u32x2{sx1[it][0], sx1[it][1]});
      } else {
        if constexpr (MODE == SM_A) {
          float xv[8], fv[8];
          if constexpr (XHL) {
            hl8_decode(sx0[it], u32x2{sx1[it][0], sx1[it][1]}, xv);
          } else if constexpr (XPAIR) {
            const f16x8_t h = __builtin_bit_cast(f16x8_t, sx0[it]), l = __builtin_bit_cast(f16x8_t, sx1[it]);
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = (float)h[e] + (float)l[e];
          } else {
            const f32x4 p = __builtin_bit_cast(f32x4, sx0[it]), q = __builtin_bit_cast(f32x4, sx1[it]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[e] = p[e]; xv[4 + e] = q[e]; }
          }
          if constexpr (FHL) {
            hl8_decode(sf0[it], u32x2{sf1[it][0], sf1[it][1]}, fv);
          } else {
            const f32x4 fp = __builtin_bit_cast(f32x4, sf0[it]), fq = __builtin_bit_cast(f32x4, sf1[it]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { fv[e] = fp[e]; fv[4 + e] = fq[e]; }
          }
          const bool inside = t >= 0 && t < Tn;        // rows outside the sample are the conv's zero padding, not b
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = inside ? ra[e] * fv[e] + rb[e] + xv[e] : 0.f;
        } else {
          const f32x4 p = __builtin_bit_cast(f32x4, sx0[it]), q = __builtin_bit_cast(f32x4, sx1[it]);   // (zeros outside the sample)
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = p[e]; v[4 + e] = q[e]; }
        }
        if constexpr (MODE == SM_A && HL) {
          // the stored row IS what every later pass sees: stage 1 below (the statistics of v') runs on the hl8 value as well
          const Hl8 e8 = hl8_encode(v);
          uh = e8.hi;
          ul = hl8_lo_f16(e8.lo);
          if (t >= s0 && t < s0 + a.L && t < Tn) {      // the span's own rows leave for HBM here, once
            char* o = reinterpret_cast<char*>(a.out) + sample_o + (size_t)t * SR_HLB;
            *reinterpret_cast<u32x4*>(o + cg * 16) = e8.hi;
            *reinterpret_cast<u32x2*>(o + 128 + cg * 8) = e8.lo;
          }
        } else {
          uint32_t h[4], l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) M::split2(v[2 * e], v[2 * e + 1], h[e], l[e]);
          uh = u32x4{h[0], h[1], h[2], h[3]};
          ul = u32x4{l[0], l[1], l[2], l[3]};
        }
      }
      char* p = lds + ring + slot * SR_SLOT + r * SR_RS + cg * 32;
      *reinterpret_cast<u32x4*>(p) = uh;
      *reinterpret_cast<u32x4*>(p + 16) = ul;
      if constexpr (MODE == SM_A && !HL) {
        if (t >= s0 && t < s0 + a.L && t < Tn) {      // the span's own rows leave for HBM here, once
          char* o = reinterpret_cast<char*>(a.out) + sample_o + (size_t)t * SR_ROWB + cg * 32;
          *reinterpret_cast<u32x4*>(o) = uh;
          *reinterpret_cast<u32x4*>(o + 16) = ul;
        }
      }
    }
  };

  // ---- this lane's B-operand bases: tap tau (offset o), column col -> row q = col + o + 5 of the (tile, tile + 1) batch pair
  int baddr[7][3];
#pragma unroll
  for (int tap = 0; tap < 7; ++tap) {
    const int q = col + mrf_std_off(tap) + SR_H, carry = q >> 4, i = q & 15;
    const int bt = ring + i * SR_RS + g * 32;
#pragma unroll
    for (int u = 0; u < 3; ++u) baddr[tap][u] = bt + (carry ? (u + 1) % 3 : u) * SR_SLOT;
  }

  const __amdgpu_buffer_rsrc_t ro =
      __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out) + sample_f, 0, FULL ? Tn * FRB : 0, 0x00020000);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};

  // ---- stage 1 of one column tile into v (U = slot of the tile's first batch); xb0 keeps the tap-0 operands for the residual 1x1.
  //      Software-pipelined (the operands of k-step s + 1 are read while the MFMAs of k-step s run); `between(s)` is called once per
  //      k-step, behind its MFMAs: pass F hangs one slice of the PREVIOUS tile's stages 2-3 there, so the VALU work of one tile
  //      (GroupNorm + SiLU, operand split, statistics: ~200 instructions) issues in the shadow of the next tile's 64 stage-1 MFMAs
  // Pass V0 keeps the 32 stage-1 weight fragments in REGISTERS (it has the room: 120 VGPRs without them).  Read from LDS per k-step
  // they are 32 KB per 16-step tile and wave against 28 KB of activation operands - with eight waves per CU the LDS array, not the
  // matrix pipe, set the pace of this pass (21.8 us for 6.8 us of MFMA time, whether its rows came through registers or by LDS-DMA).
  constexpr bool AREG = (MODE == SM_V0);
  VA areg[AREG ? 16 : 1][2];
  if constexpr (AREG) {
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) areg[f][ks] = M::load_a(wl + (size_t)(f * 2 + ks) * FRAG_BYTES + lane * 16, 0);
  }
  auto stage1 = [&](auto uc, f32x4 (&v)[4], VB (&xb0)[2], auto&& between) {
    constexpr int U = decltype(uc)::value;
#pragma unroll
    for (int m = 0; m < 4; ++m) v[m] = *reinterpret_cast<const f32x4*>(b_conv + 16 * m + 4 * g);
    VB bfr[2];
    VA afr[2][4];
    auto ld_step = [&](int sidx, int set) {
      const int tap = sidx >> 1, ks = sidx & 1;
      bfr[set] = M::load_bp(lds + baddr[tap][U] + ks * 128, 16);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (mrf_std_frag(m, tap) >= 0) {
          if constexpr (AREG) afr[set][m] = areg[mrf_std_frag(m, tap)][ks];
          else afr[set][m] = M::load_a(wl + (size_t)(mrf_std_frag(m, tap) * 2 + ks) * FRAG_BYTES + lane * 16, 0);
        }
    };
    ld_step(0, 0);
#pragma unroll
    for (int sidx = 0; sidx < 14; ++sidx) {
      if (sidx + 1 < 14) ld_step(sidx + 1, (sidx + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      const int tap = sidx >> 1, set = sidx & 1;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (mrf_std_frag(m, tap) >= 0) v[m] = M::mma(afr[set][m], bfr[set], v[m]);
      __builtin_amdgcn_s_setprio(0);
      if (tap == 3) xb0[sidx & 1] = bfr[set];
      between(sidx);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto nothing = [](int) {};

  if constexpr (!FULL) {
    // ---- V0 / A: stage 1 + partial sums of v per GN(5,20) group (concat rows 16m + 4g .. +3 are exactly group 4m + g)
    auto tile = [&](auto uc, int j) {
      f32x4 v[4];
      VB xb0[2];
      stage1(uc, v, xb0, nothing);
      SR_TM(1);
      const int t0 = s0 + 16 * j;
      if (t0 + 16 <= Tn) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float q = v[m][r]; s1[m] += q; s2[m] += q * q; }
      } else {
        const bool ok = (t0 + col) < Tn;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float q = ok ? v[m][r] : 0.f; s1[m] += q; s2[m] += q * q; }
      }
    };
    if constexpr (!DMA) {
      if (nb > 0) {
        fill_issue(0); fill_commit(0, 0);
        fill_issue(1); fill_commit(1, 1);
      }
    }
    auto step = [&](auto uc, int j) {
      constexpr int U = decltype(uc)::value;
      if constexpr (DMA) {
        // V0 on pair rows: batches 0-2 were requested in the prologue; tile j needs batches j and j + 1, and the only thing requested
        // after them is DMA(j + 2) (5 instructions; nothing is stored in this pass).  Slot j % 3 is free behind the tile: batch j + 3.
        if (j + 2 < nb) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SR_TM(0);
        tile(uc, j);
        SR_TM(2);
        if (j + 3 < nb) dma(j + 3, U);
        SR_TM(3);
      } else {
        if (j + 2 < nb) fill_issue(j + 2);              // in flight under this tile's MFMAs
        SR_TM(0);
        tile(uc, j);
        SR_TM(2);
        if (j + 2 < nb) fill_commit(j + 2, (U + 2) % 3);
        SR_TM(3);
      }
    };
    for (int j = 0; j < nt; j += 3) {
      step(std::integral_constant<int, 0>{}, j);
      if (j + 1 < nt) step(std::integral_constant<int, 1>{}, j + 1);
      if (j + 2 < nt) step(std::integral_constant<int, 2>{}, j + 2);
    }
  } else {
    // ---- F0 / F: stages 2-3 of tile j ride on stage 1 of tile j + 1.  Per (m, r): w = v * scl + sh (GroupNorm(5,20)), c = SiLU(w) + b_res
    float scl[4][4], sh[4][4], brs[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float mu = st5[(4 * m + g) * 2], rs = st5[(4 * m + g) * 2 + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        scl[m][r] = rs * g5[16 * m + 4 * g + r];
        sh[m][r] = be5[16 * m + 4 * g + r] - mu * scl[m][r];
        brs[m][r] = b_res[16 * m + 4 * g + r];
      }
    }
    f32x4 vc[4], vn[4], f[4];
    VB xb0c[2], xb0n[2], cb;
    VA af[4];
    int jcur = 0;
    auto piece = [&](int S) {        // slice S (0..13) of stages 2-3 of tile jcur; S is a compile-time constant after unrolling
      if (S < 8) {
        const int m = S >> 1, r0 = (S & 1) * 2;
#pragma unroll
        for (int r = r0; r < r0 + 2; ++r) {
          const float w = vc[m][r] * scl[m][r] + sh[m][r];
          const float e = __builtin_amdgcn_exp2f(w * -1.4426950408889634f);
          vc[m][r] = w * __builtin_amdgcn_rcpf(1.f + e) + brs[m][r];          // SiLU (v_exp + v_rcp, 1 ulp) + residual bias
        }
        if (S == 7) {
#pragma unroll
          for (int m2 = 0; m2 < 4; ++m2) af[m2] = M::load_a(wl + (size_t)(MRF_CONV_FRAGS + m2 * 2 + 0) * FRAG_BYTES + lane * 16, 0);
        }
      } else if (S == 8 || S == 9) {  // residual 1x1 accumulates onto the activated values
        const int ks = S - 8;
#pragma unroll
        for (int m = 0; m < 4; ++m) vc[m] = M::mma(af[m], xb0c[ks], vc[m]);
        if (ks == 0) {
#pragma unroll
          for (int m = 0; m < 4; ++m) af[m] = M::load_a(wl + (size_t)(MRF_CONV_FRAGS + m * 2 + 1) * FRAG_BYTES + lane * 16, 0);
        } else {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            af[m] = M::load_a(wl + (size_t)(MRF_CONV_FRAGS + MRF_RES_FRAGS + m * 2 + 0) * FRAG_BYTES + lane * 16, 0);
            f[m] = *reinterpret_cast<const f32x4*>(b_fus + 16 * m + 4 * g);
          }
        }
      } else if (S == 10) {
        cb = M::from_acc(vc[0], vc[1]);
      } else if (S == 11) {          // fusion 1x1, k-step 0 (c straight from the accumulators)
#pragma unroll
        for (int m = 0; m < 4; ++m) f[m] = M::mma(af[m], cb, f[m]);
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = M::load_a(wl + (size_t)(MRF_CONV_FRAGS + MRF_RES_FRAGS + m * 2 + 1) * FRAG_BYTES + lane * 16, 0);
        cb = M::from_acc(vc[2], vc[3]);
      } else if (S == 12) {
#pragma unroll
        for (int m = 0; m < 4; ++m) f[m] = M::mma(af[m], cb, f[m]);
      } else {
        // partial sums of f per GN(8,64) group (rows 16m + 4g + r -> group 2m + (g >> 1)), then f leaves straight from the accumulators:
        // EXACTLY four store instructions per tile (rows past the sample are dropped by the range check, not by a branch) - the
        // counted vmcnt below relies on that
        const int t0 = s0 + 16 * jcur;
        if (t0 + 16 <= Tn) {
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float q = f[m][r]; s1[m] += q; s2[m] += q * q; }
        } else {
          const bool ok = (t0 + col) < Tn;
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float q = ok ? f[m][r] : 0.f; s1[m] += q; s2[m] += q * q; }
        }
        if constexpr (FHL) {          // this lane's 4 channels of M-tile m: 8 bytes of the hi plane, 4 of the lo plane
          const int vo = (t0 + col) * SR_HLB;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const uint32_t h0 = pack_f16(f[m][0], f[m][1]), h1 = pack_f16(f[m][2], f[m][3]);
            const f16x2_t a0 = __builtin_bit_cast(f16x2_t, h0), a1 = __builtin_bit_cast(f16x2_t, h1);
            const uint32_t l4 = hl8_pack4((f[m][0] - (float)a0[0]) * SR_LOS, (f[m][1] - (float)a0[1]) * SR_LOS,
                                          (f[m][2] - (float)a1[0]) * SR_LOS, (f[m][3] - (float)a1[1]) * SR_LOS);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{h0, h1}, ro, vo + 32 * m + 8 * g, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(l4, ro, vo + 128 + 16 * m + 4 * g, 0, 0);
          }
        } else {
          const int vo = (t0 + col) * SR_ROWB + 16 * g;
#pragma unroll
          for (int m = 0; m < 4; ++m) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f[m]), ro, vo + 64 * m, 0, 0);
        }
      }
    };
    auto finish_only = [&]() {
#pragma unroll
      for (int S = 0; S < 14; ++S) piece(S);
    };
    auto rotate = [&]() {
#pragma unroll
      for (int m = 0; m < 4; ++m) vc[m] = vn[m];
      xb0c[0] = xb0n[0]; xb0c[1] = xb0n[1];
    };
    if (nt > 0) {
      if constexpr (DMA) {
        // Batches j and j + 1 feed stage 1 of tile j, which runs in iteration j - 1; slot j % 3 is free once it is done, and batch j + 3
        // is requested there.  Vector-memory operations retire in issue order: before stage 1 of tile j + 1 everything up to DMA(j + 2)
        // must have landed, and what was issued after it is the 4 stores of tile j - 1 and, if it exists, DMA(j + 3) (5 instructions).
        if (nb > 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stage1(std::integral_constant<int, 0>{}, vc, xb0c, nothing);
        if (3 < nb) dma(3, 0);
        auto step = [&](auto un, int j) {          // UN = (j + 1) % 3
          constexpr int UN = decltype(un)::value;
          jcur = j;
          if (j + 1 < nt) {
            if (j == 0) {
              if (3 < nb) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
              if (j + 3 < nb) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            SR_TM(0);
            stage1(un, vn, xb0n, piece);
            SR_TM(1);
            if (j + 4 < nb) dma(j + 4, UN);
            rotate();
            SR_TM(3);
          } else {
            finish_only();
            SR_TM(2);
          }
        };
        for (int j = 0; j < nt; j += 3) {
          step(std::integral_constant<int, 1>{}, j);
          if (j + 1 < nt) step(std::integral_constant<int, 2>{}, j + 1);
          if (j + 2 < nt) step(std::integral_constant<int, 0>{}, j + 2);
        }
      } else {
        fill_issue(0); fill_commit(0, 0);
        fill_issue(1); fill_commit(1, 1);
        if (nb > 2) { fill_issue(2); fill_commit(2, 2); }
        stage1(std::integral_constant<int, 0>{}, vc, xb0c, nothing);
        auto step = [&](auto un, int j) {
          constexpr int UN = decltype(un)::value;
          jcur = j;
          if (j + 1 < nt) {
            if (j + 3 < nb) fill_issue(j + 3);        // in flight under this iteration's MFMAs
            SR_TM(0);
            stage1(un, vn, xb0n, piece);
            SR_TM(1);
            if (j + 3 < nb) fill_commit(j + 3, (UN + 2) % 3);     // slot j % 3: batch j is dead (stage 1 of tile j ran an iteration ago)
            rotate();
            SR_TM(3);
          } else {
            finish_only();
            SR_TM(2);
          }
        };
        for (int j = 0; j < nt; j += 3) {
          step(std::integral_constant<int, 1>{}, j);
          if (j + 1 < nt) step(std::integral_constant<int, 2>{}, j + 1);
          if (j + 2 < nt) step(std::integral_constant<int, 0>{}, j + 2);
        }
      }
    }
  }

#ifdef MV_SR_TIMING
  if (lane == 0 && sr_dbg) {
    long long* d = sr_dbg + ((size_t)MODE * 65536 + ((size_t)(b * nwg + wg) * SR_NW + wid)) * 8;
    for (int i = 0; i < 4; ++i) d[i] = tacc[i];
    d[4] = clock64() - tfirst;
    d[5] = t_abs0;
    d[6] = __builtin_amdgcn_s_memrealtime();
    d[7] = nt;
  }
#endif
  // ---- statistics: lanes -> wave -> workgroup (fixed order), one partial per workgroup
  if constexpr (!FULL) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s1[m] += __shfl_xor(s1[m], o, 64); s2[m] += __shfl_xor(s2[m], o, 64); }
    }
    if (col == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) { red[(wid * 16 + 4 * m + g) * 2] = s1[m]; red[(wid * 16 + 4 * m + g) * 2 + 1] = s2[m]; }
    }
    __syncthreads();
    if (tid < 32) {
      float acc = 0.f;
      for (int w = 0; w < SR_NW; ++w) acc += red[w * 32 + tid];
      a.part5_out[(size_t)(b * nwg + wg) * 32 + tid] = acc;
    }
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int o = 1; o <= 16; o <<= 1) { s1[m] += __shfl_xor(s1[m], o, 64); s2[m] += __shfl_xor(s2[m], o, 64); }
    }
    if (col == 0 && (g & 1) == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) { red[(wid * 8 + 2 * m + (g >> 1)) * 2] = s1[m]; red[(wid * 8 + 2 * m + (g >> 1)) * 2 + 1] = s2[m]; }
    }
    __syncthreads();
    if (tid < 16) {
      float acc = 0.f;
      for (int w = 0; w < SR_NW; ++w) acc += red[w * 16 + tid];
      a.part8_out[(size_t)(b * nwg + wg) * 16 + tid] = acc;
    }
  }
}

// the chain's materialised output (mv_mrf_chain_fwd_cl): out = a f + b + x as fp32 rows; x fp32 or pair rows
template <bool XPAIR, bool HL>      // HL: f is hl8 rows, and so is x when XPAIR
__global__ __launch_bounds__(256) void mrf_stream_final_kernel(const float* __restrict__ f, const void* __restrict__ x,
                                                               const float* __restrict__ part8, const float* __restrict__ tab,
                                                               float* __restrict__ out, int Tn, int nwg, float eps) {
  __shared__ float ab[128];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid < 64) {
    const int q = tid >> 3;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < nwg; ++i) {
      s1 += part8[((size_t)(b * nwg + i) * 8 + q) * 2];
      s2 += part8[((size_t)(b * nwg + i) * 8 + q) * 2 + 1];
    }
    const float n = 8.f * (float)Tn, mu = s1 / n;
    const float rs = rsqrtf(fmaxf(s2 / n - mu * mu, 0.f) + eps);
    const float aa = rs * tab[320 + tid];
    ab[tid] = aa;
    ab[64 + tid] = tab[384 + tid] - mu * aa;
  }
  __syncthreads();
  const size_t base = (size_t)b * Tn * 64;
  const long total = (long)Tn * 8;                   // items = (row, 8-channel group)
  for (long i = (long)blockIdx.x * 256 + tid; i < total; i += (long)gridDim.x * 256) {
    const long t = i >> 3;
    const int cg = (int)(i & 7);
    float xv[8];
    if constexpr (XPAIR && HL) {
      const char* p = reinterpret_cast<const char*>(x) + ((size_t)b * Tn + t) * SR_HLB;
      hl8_decode(*reinterpret_cast<const u32x4*>(p + cg * 16), *reinterpret_cast<const u32x2*>(p + 128 + cg * 8), xv);
    } else if constexpr (XPAIR) {
      const char* p = reinterpret_cast<const char*>(x) + (base + (size_t)t * 64) * 4 + cg * 32;
      const f16x8_t h = *reinterpret_cast<const f16x8_t*>(p), l = *reinterpret_cast<const f16x8_t*>(p + 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = (float)h[e] + (float)l[e];
    } else {
      const float* p = reinterpret_cast<const float*>(x) + base + (size_t)t * 64 + cg * 8;
      const f32x4 u = *reinterpret_cast<const f32x4*>(p), w = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { xv[e] = u[e]; xv[4 + e] = w[e]; }
    }
    f32x4 fu, fw;
    if constexpr (HL && SR_FHL) {
      const char* p = reinterpret_cast<const char*>(f) + ((size_t)b * Tn + t) * SR_HLB;
      float fv[8];
      hl8_decode(*reinterpret_cast<const u32x4*>(p + cg * 16), *reinterpret_cast<const u32x2*>(p + 128 + cg * 8), fv);
      fu = f32x4{fv[0], fv[1], fv[2], fv[3]};
      fw = f32x4{fv[4], fv[5], fv[6], fv[7]};
    } else {
      const float* fp = f + base + (size_t)t * 64 + cg * 8;
      fu = *reinterpret_cast<const f32x4*>(fp);
      fw = *reinterpret_cast<const f32x4*>(fp + 4);
    }
    f32x4 o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o0[e] = ab[cg * 8 + e] * fu[e] + ab[64 + cg * 8 + e] + xv[e];
      o1[e] = ab[cg * 8 + 4 + e] * fw[e] + ab[64 + cg * 8 + 4 + e] + xv[4 + e];
    }
    float* op = out + base + (size_t)t * 64 + cg * 8;
    *reinterpret_cast<f32x4*>(op) = o0;
    *reinterpret_cast<f32x4*>(op + 4) = o1;
  }
}

static inline void sr_geometry(int B, int Tn, int* L, int* nwg) {
  const int tiles = cdiv(Tn, 16);
  const int wgs = 256 / B > 1 ? 256 / B : 1;          // workgroups per sample that fill the chip once
  int tpw = cdiv(tiles, wgs * SR_NW);
  if (tpw < 1) tpw = 1;
  *L = tpw * 16;
  *nwg = cdiv(tiles, tpw * SR_NW);
}

template <int MODE, bool XPAIR, bool HL = false>
static void sr_launch(const SrArgs& a, int B, hipStream_t stream) {
  constexpr bool FULL = (MODE == SM_F0 || MODE == SM_F);
  const size_t lds = (size_t)(FULL ? SR_WB_ALL : SR_WB_CONV) + MRF_TAB_FLOATS * 4 + (32 + 16 + SR_NW * 32) * 4 + 512 * 8 + (size_t)SR_NW * SR_RING;
  auto k = mrf_stream_kernel<MODE, XPAIR, HL>;
  static bool set = false;
  if (!set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set = true; }
  hipLaunchKernelGGL(k, dim3(a.nwg, B), dim3(SR_NW * 64), lds, stream, a);
}

}  // namespace mv

using namespace mv;

// internal entry (mrf_fused.hip): the whole chain in the streaming form.  packed[i] = MV_F16 image of block i, every block with the
// dilations (1, 3, 5).  ws: 3 activation buffers + partial sums (mv_mrf_chain_workspace_bytes).  out != nullptr: fp32 rows of the last
// block's output are written; otherwise the caller gets (f_last, x_last, x_last_pair, part8, tab, nwg) for the fused output conv.
int mvi_mrf_chain_stream(const void* x, void* out, const void* const* packed, int nblocks, char* ws, size_t act_bytes, int B, int Tn,
                         float eps, hipStream_t stream, const void** f_last, const void** x_last, int* x_last_pair,
                         const float** part8_last, const float** tab_last, int* nwg_last, int x_pair_in) {
  if ((size_t)Tn * SR_ROWB >= (1ull << 31)) return MV_ERR_UNSUPPORTED;     // 32-bit buffer offsets
  int L, nwg;
  sr_geometry(B, Tn, &L, &nwg);
#ifdef MV_SR_TIMING
  static long long* dbg = nullptr;
  static int calls = 0;
  if (!dbg) { (void)hipMalloc(&dbg, 4 * 65536 * 8 * 8); (void)hipMemset(dbg, 0xff, 4 * 65536 * 8 * 8); (void)hipMemcpyToSymbol(HIP_SYMBOL(sr_dbg), &dbg, sizeof(dbg)); }
#endif
  float* fbuf = reinterpret_cast<float*>(ws);
  char* xbuf[2] = {ws + act_bytes, ws + 2 * act_bytes};
  float* part5 = reinterpret_cast<float*>(ws + 3 * act_bytes);
  float* part8 = part5 + (size_t)B * nwg * 32;
  SrArgs a{};
  a.Tn = Tn; a.L = L; a.nwg = nwg; a.eps = eps;
  { const char* e = getenv("MV_SR_DBG"); a.dbg = e ? atoi(e) : 0; }
  const void* xi = x;
  bool xi_pair = x_pair_in != 0;               // the producer already wrote pair rows (the last upsampler's streaming kernel)
  // hl8 rows between the blocks (MV_MRF_HL8=0: the 256-byte pair / fp32 rows of the first streaming form; pair-row input keeps them)
  static int hl_env = -1;
  if (hl_env < 0) { const char* e = getenv("MV_MRF_HL8"); hl_env = e ? atoi(e) : 1; }
  const bool hl = hl_env != 0 && !xi_pair;
  for (int i = 0; i < nblocks; ++i) {
    a.packed = (const char*)packed[i];
    if (i == 0) {
      a.x = xi; a.fprev = nullptr; a.out = nullptr; a.packed_prev = nullptr;
      a.part5_in = nullptr; a.part5_out = part5; a.part8_in = nullptr; a.part8_out = nullptr;
      if (xi_pair) sr_launch<SM_V0, true>(a, B, stream); else sr_launch<SM_V0, false>(a, B, stream);
      a.out = fbuf; a.part5_in = part5; a.part5_out = nullptr; a.part8_out = part8;
      if (xi_pair) sr_launch<SM_F, true>(a, B, stream);
      else if (hl) sr_launch<SM_F0, false, true>(a, B, stream);
      else sr_launch<SM_F0, false>(a, B, stream);
    } else {
      char* xn = xbuf[i & 1];
      a.x = xi; a.fprev = fbuf; a.out = xn; a.packed_prev = (const char*)packed[i - 1];
      a.part5_in = nullptr; a.part5_out = part5; a.part8_in = part8; a.part8_out = nullptr;
      if (hl) { if (xi_pair) sr_launch<SM_A, true, true>(a, B, stream); else sr_launch<SM_A, false, true>(a, B, stream); }
      else if (xi_pair) sr_launch<SM_A, true>(a, B, stream);
      else sr_launch<SM_A, false>(a, B, stream);
      xi = xn; xi_pair = true;
      a.x = xi; a.fprev = nullptr; a.out = fbuf; a.packed_prev = nullptr;
      a.part5_in = part5; a.part5_out = nullptr; a.part8_in = nullptr; a.part8_out = part8;
      if (hl) sr_launch<SM_F, true, true>(a, B, stream); else sr_launch<SM_F, true>(a, B, stream);
    }
  }
#ifdef MV_SR_TIMING
  {
    const char* e = getenv("MV_MRF_TIMING_CALL");
    if (++calls == (e ? atoi(e) : 30)) {
      (void)hipStreamSynchronize(stream);
      static long long hbuf[4 * 65536 * 8];
      (void)hipMemcpy(hbuf, dbg, sizeof(hbuf), hipMemcpyDeviceToHost);
      for (int md = 0; md < 4; ++md) {
        double av[5] = {0}; int cnt = 0; long long tmin = -1, tmax = -1, smax = -1;
        for (int w = 0; w < 65536; ++w) {
          const long long* d = hbuf + ((size_t)md * 65536 + w) * 8;
          if (d[4] < 0 || d[7] <= 0) continue;
          for (int k = 0; k < 5; ++k) av[k] += (double)d[k];
          ++cnt;
          if (tmin < 0 || d[5] < tmin) tmin = d[5];
          if (d[5] > smax) smax = d[5];
          if (d[6] > tmax) tmax = d[6];
        }
        if (!cnt) continue;
        fprintf(stderr, "[mrf stream timing] mode %d L %d nwg %d waves %d: span %.2f us (last start +%.2f us); per wave cycles: wait %.0f stage1 %.0f stage23+store %.0f fill/dma %.0f total %.0f\n",
                md, L, nwg, cnt, (tmax - tmin) / 100.0, (smax - tmin) / 100.0, av[0] / cnt, av[1] / cnt, av[2] / cnt, av[3] / cnt, av[4] / cnt);
      }
    }
  }
#endif
  const float* tabp = reinterpret_cast<const float*>((const char*)packed[nblocks - 1] + SR_WB_ALL);
  if (out) {
    const long items = (long)Tn * 8;
    const int gx = (int)((items + 255) / 256 > 1024 ? 1024 : (items + 255) / 256);
    if (hl && xi_pair)
      hipLaunchKernelGGL((mrf_stream_final_kernel<true, true>), dim3(gx, B), dim3(256), 0, stream, fbuf, xi, part8, tabp, (float*)out, Tn, nwg, eps);
    else if (hl)
      hipLaunchKernelGGL((mrf_stream_final_kernel<false, true>), dim3(gx, B), dim3(256), 0, stream, fbuf, xi, part8, tabp, (float*)out, Tn, nwg, eps);
    else if (xi_pair)
      hipLaunchKernelGGL((mrf_stream_final_kernel<true, false>), dim3(gx, B), dim3(256), 0, stream, fbuf, xi, part8, tabp, (float*)out, Tn, nwg, eps);
    else
      hipLaunchKernelGGL((mrf_stream_final_kernel<false, false>), dim3(gx, B), dim3(256), 0, stream, fbuf, xi, part8, tabp, (float*)out, Tn, nwg, eps);
    return MV_OK;
  }
  // x_last_pair: 0 = x and f fp32 rows, 1 = x pair rows, 2 = x and f hl8 rows, 3 = x fp32 with f hl8 (one-block chain), 4 = x hl8, f fp32
  *f_last = fbuf; *x_last = xi;
  *x_last_pair = !hl ? (xi_pair ? 1 : 0) : SR_FHL ? (xi_pair ? 2 : 3) : (xi_pair ? 4 : 0);
  *part8_last = part8; *tab_last = tabp; *nwg_last = nwg;
  return MV_OK;
}
