// Shared definitions of the fused MultiReceptiveFieldBlock kernels (mrf_fused.hip: tile form; mrf_stream.hip: streaming form):
// packed-weight layout (mv_mrf_pack), tap tables, and the compile-time tables of the reference's dilations (1, 3, 5).
#pragma once
#include "mfma.h"

namespace mv {

constexpr int MRF_C = 64;        // residual-stream channels
constexpr int MRF_CPD = 20;      // channels per dilation branch
constexpr int MRF_NBR = 3;
constexpr int MRF_MAXTAPS = 8;
constexpr int MRF_CONV_FRAGS = 32, MRF_RES_FRAGS = 8, MRF_FUS_FRAGS = 8;
constexpr int MRF_TAB_FLOATS = 7 * 64;

struct MrfMeta {
  int ntaps, halo;
  int tap_off[MRF_MAXTAPS];
  int frag_of[4][MRF_MAXTAPS];   // fragment-pair index of (M-tile, tap) or -1
  int dil[MRF_NBR];
};

static inline bool mrf_make_meta(const int* dil, MrfMeta* m) {
  int offs[MRF_MAXTAPS], n = 0;
  auto add = [&](int o) {
    for (int i = 0; i < n; ++i) if (offs[i] == o) return;
    if (n < MRF_MAXTAPS) offs[n++] = o; else n = MRF_MAXTAPS + 1;
  };
  add(0);
  for (int i = 0; i < MRF_NBR; ++i) { if (dil[i] < 1 || dil[i] > 8) return false; add(-dil[i]); add(dil[i]); }
  if (n > 7) return false;
  for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) if (offs[j] < offs[i]) { int t = offs[i]; offs[i] = offs[j]; offs[j] = t; }
  m->ntaps = n; m->halo = 0;
  for (int i = 0; i < MRF_MAXTAPS; ++i) m->tap_off[i] = i < n ? offs[i] : 0;
  for (int i = 0; i < MRF_NBR; ++i) { m->dil[i] = dil[i]; if (dil[i] > m->halo) m->halo = dil[i]; }
  int next = 0;
  for (int mt = 0; mt < 4; ++mt) {
    for (int t = 0; t < MRF_MAXTAPS; ++t) m->frag_of[mt][t] = -1;
    for (int row = 16 * mt; row < 16 * mt + 16 && row < MRF_NBR * MRF_CPD; ++row) {
      const int br = row / MRF_CPD;
      for (int t = 0; t < n; ++t)
        if ((offs[t] == 0 || offs[t] == dil[br] || offs[t] == -dil[br]) && m->frag_of[mt][t] < 0) m->frag_of[mt][t] = -2;
    }
    for (int t = 0; t < n; ++t) if (m->frag_of[mt][t] == -2) m->frag_of[mt][t] = next++;
  }
  return next * 2 <= MRF_CONV_FRAGS;
}

template <typename T> constexpr size_t mrf_packed_bytes() {
  return (size_t)(MRF_CONV_FRAGS + MRF_RES_FRAGS + MRF_FUS_FRAGS) * Mma<T>::NSETS * FRAG_BYTES + MRF_TAB_FLOATS * 4;
}

__host__ __device__ constexpr int mrf_std_off(int tap) { return tap == 0 ? -5 : tap == 1 ? -3 : tap == 2 ? -1 : tap == 3 ? 0 : tap == 4 ? 1 : tap == 5 ? 3 : 5; }
__host__ __device__ constexpr int mrf_std_frag(int m, int tap) {
  // mrf_make_meta's numbering for dil = {1,3,5}: M-tile 0 uses taps 2,3,4; 1: 1..5; 2: 0,1,3,5,6; 3: 0,3,6
  return m == 0 ? (tap >= 2 && tap <= 4 ? tap - 2 : -1)
       : m == 1 ? (tap >= 1 && tap <= 5 ? 3 + tap - 1 : -1)
       : m == 2 ? (tap == 0 ? 8 : tap == 1 ? 9 : tap == 3 ? 10 : tap == 5 ? 11 : tap == 6 ? 12 : -1)
                : (tap == 0 ? 13 : tap == 3 ? 14 : tap == 6 ? 15 : -1);
}
static inline bool mrf_meta_is_std(const MrfMeta& m) {
  if (m.ntaps != 7 || m.halo != 5) return false;
  for (int t = 0; t < 7; ++t) {
    if (m.tap_off[t] != mrf_std_off(t)) return false;
    for (int a = 0; a < 4; ++a) if (m.frag_of[a][t] != mrf_std_frag(a, t)) return false;
  }
  return true;
}


}  // namespace mv
