// Host-side table construction for an afx_plan: what the reference gets from
// scipy.signal.get_window('hamming', n, fftbins=True), librosa.filters.mel
// (Slaney scale + Slaney area norm, float32 storage) and scipy.fft.dct(type=2,
// norm='ortho') underneath librosa.feature.mfcc
// (audio_feature_extraction_toolkit/core/feature_extractor.py:127-134).
// Everything is evaluated in double and rounded where librosa rounds.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>

#include "afx_internal.h"
#include "afx_frames3.h"

namespace afx {

static const double kPi = 3.14159265358979323846;

int validate_params(const afx_params& p, std::string& msg) {
  auto bad = [&](const char* m) { msg = m; return (int)AFX_ERR_INVALID; };
  if (p.sr <= 0) return bad("sr must be positive");
  if (p.hop <= 0) return bad("hop_length must be positive");
  if (p.n_mels <= 0 || p.n_mels > 512) return bad("n_mels must be in [1, 512]");
  if (p.n_mels % 4 != 0) return bad("n_mels must be a multiple of 4");
  if (p.n_mfcc <= 0 || p.n_mfcc > p.n_mels || p.n_mfcc > 128)
    return bad("n_mfcc must be in [1, min(n_mels, 128)]");
  if (p.window != AFX_WINDOW_HAMMING && p.window != AFX_WINDOW_HANN) return bad("unknown window");
  if (p.delta_width != 9) return bad("delta_width must be 9 (librosa default used by the reference)");
  if (p.trim_hop <= 0 || p.trim_frame <= 0 || p.trim_frame % p.trim_hop != 0 ||
      (p.trim_frame / p.trim_hop) % 2 != 0)
    return bad("trim_frame must be an even multiple of trim_hop");
  if (!(p.amin > 0.f)) return bad("amin must be positive");
  if (!(p.fmin >= 0.f) || (p.fmax != 0.f && !(p.fmax > p.fmin)) || p.fmax > 0.5f * (float)p.sr + 1e-3f)
    return bad("need 0 <= fmin < fmax <= sr / 2 (fmax = 0 means sr / 2)");
  if (!(p.lifter >= 0.f)) return bad("lifter must be >= 0");
  bool pow2 = p.n_fft > 0 && (p.n_fft & (p.n_fft - 1)) == 0;
  if (!pow2 || p.n_fft < 256 || p.n_fft > 4096) {
    msg = "frame_length must be a power of two in [256, 4096]";
    return (int)AFX_ERR_UNSUPPORTED;
  }
  return AFX_OK;
}

// librosa.core.convert.hz_to_mel / mel_to_hz, htk=False
static double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0;
  const double min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0;
  const double min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

static void build_mel_dense(const afx_params& p, std::vector<float>& W, std::vector<double>& mel_f) {
  const int M = p.n_mels, NB = p.n_fft / 2 + 1;
  const double fmin = (double)p.fmin, fmax = p.fmax > 0.f ? (double)p.fmax : (double)p.sr / 2.0;
  // mel_frequencies(n_mels + 2, fmin, fmax, htk): np.linspace in the mel domain (Slaney's scale, or HTK's 2595 log10(1 + f / 700))
  auto to_mel = [&](double f) { return p.htk ? 2595.0 * std::log10(1.0 + f / 700.0) : hz_to_mel(f); };
  auto to_hz = [&](double m) { return p.htk ? 700.0 * (std::pow(10.0, m / 2595.0) - 1.0) : mel_to_hz(m); };
  mel_f.assign(M + 2, 0.0);
  const double m0 = to_mel(fmin), m1 = to_mel(fmax);
  const double step = (m1 - m0) / (double)(M + 1);
  for (int i = 0; i < M + 2; ++i) mel_f[i] = to_hz(i == M + 1 ? m1 : (double)i * step + m0);
  W.assign((size_t)M * NB, 0.f);
  for (int i = 0; i < M; ++i) {
    const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
    const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
    for (int k = 0; k < NB; ++k) {
      // np.fft.rfftfreq(n, d=1/sr): arange(N) * (1.0 / (n * d))
      const double fk = (double)k * (1.0 / ((double)p.n_fft * (1.0 / (double)p.sr)));
      const double lower = -(mel_f[i] - fk) / fd0;
      const double upper = (mel_f[i + 2] - fk) / fd1;
      const double tri = std::fmax(0.0, std::fmin(lower, upper));
      const float w32 = (float)tri;                 // weights[i] = ... (float32 array)
      W[(size_t)i * NB + k] = (float)((double)w32 * enorm);   // weights *= enorm[:, None]
    }
  }
}

static void build_mel_blocks(const afx_params& p, const std::vector<float>& W,
                             const std::vector<double>& mel_f, MelBlocks& s) {
  const int M = p.n_mels, NB = p.n_fft / 2 + 1;
  const int G = (M + 15) / 16;
  s.n_groups = G; s.grp.clear(); s.coef.clear(); s.koff.clear();
  const double delta = 1.0 / ((double)p.n_fft * (1.0 / (double)p.sr));
  int first_blk = 0;
  std::vector<int> nblk(G, 0);
  for (int g = 0; g < G; ++g) {
    int kmin = NB, kmax = -1;
    for (int m = 16 * g; m < std::min(M, 16 * g + 16); ++m)
      for (int k = 0; k < NB; ++k)
        if (W[(size_t)m * NB + k] != 0.f) { kmin = std::min(kmin, k); kmax = std::max(kmax, k); }
    int nb = 0;
    if (kmax >= 0) nb = (kmax - kmin + 1 + 3) / 4; else kmin = 0;
    nblk[g] = nb;
    s.grp.push_back(kmin); s.grp.push_back(nb); s.grp.push_back(first_blk); s.grp.push_back(g);
    for (int i = 0; i < 16; ++i) {
      const int m = 16 * g + i;
      if (m >= M) { for (int c = 0; c < 4; ++c) s.coef.push_back(0.f); s.koff.push_back(0.f); continue; }
      const double fd0 = mel_f[m + 1] - mel_f[m], fd1 = mel_f[m + 2] - mel_f[m + 1];
      const double en = 2.0 / (mel_f[m + 2] - mel_f[m]);
      const long kc = std::lround(mel_f[m + 1] / delta);
      s.coef.push_back((float)(((double)kc * delta - mel_f[m]) * en / fd0));
      s.coef.push_back((float)(delta * en / fd0));
      s.coef.push_back((float)((mel_f[m + 2] - (double)kc * delta) * en / fd1));
      s.coef.push_back((float)(-delta * en / fd1));
      s.koff.push_back((float)((long)kmin - kc));
    }
    first_blk += nb;
  }
  // ---- work items: split oversized groups, then longest-processing-time-first over 4 waves
  struct Part { int g, b0, nb, role, slot, nslots; };
  std::vector<Part> parts;
  int total = 0;
  for (int g = 0; g < G; ++g) total += nblk[g];
  const double fair = std::max(1.0, total / 4.0);
  int slots = 0;
  for (int g = 0; g < G; ++g) {
    if (nblk[g] == 0) { parts.push_back({g, 0, 0, 0, 0, 0}); continue; }
    int np = 1;
    if (nblk[g] > 0.8 * fair) np = std::min(kMelRegItems, (int)std::ceil(nblk[g] / (0.7 * fair)));
    if (slots + (np - 1) > kMelMaxSlots) np = 1 + std::max(0, kMelMaxSlots - slots);
    const int base = nblk[g] / np, rem = nblk[g] % np;
    int b0 = 0;
    for (int i = 0; i < np; ++i) {
      const int nb = base + (i < rem ? 1 : 0);
      Part pt{g, b0, nb, 0, 0, 0};
      if (np > 1) {
        if (i < np - 1) { pt.role = 1; pt.slot = slots + i; }
        else { pt.role = 2; pt.slot = slots; pt.nslots = np - 1; }
      }
      parts.push_back(pt);
      b0 += nb;
    }
    slots += np - 1;
  }
  s.n_slots = slots;
  std::stable_sort(parts.begin(), parts.end(), [](const Part& a, const Part& b) { return a.nb > b.nb; });
  int load[4] = {0, 0, 0, 0};
  std::vector<Part> per[4];
  for (const Part& pt : parts) {
    int best = -1;
    for (int w = 0; w < 4; ++w) {
      if ((int)per[w].size() >= kMelMaxItems) continue;
      bool clash = false;      // parts of one group go to different waves
      for (const Part& q : per[w]) if (q.g == pt.g) clash = true;
      if (clash) continue;
      if (best < 0 || load[w] < load[best]) best = w;
    }
    if (best < 0) best = (int)(std::min_element(load, load + 4) - load);
    per[best].push_back(pt);
    load[best] += pt.nb + 1;
  }
  for (int w = 0; w < 4; ++w)          // split parts first (they are the largest anyway)
    std::stable_sort(per[w].begin(), per[w].end(), [](const Part& a, const Part& b) { return (a.role != 0) > (b.role != 0); });
  s.items.assign((size_t)4 * kMelMaxItems * 8, 0);
  for (int w = 0; w < 4; ++w) {
    s.item_cnt[w] = (int32_t)std::min<size_t>(per[w].size(), kMelMaxItems);
    for (int i = 0; i < s.item_cnt[w]; ++i) {
      int32_t* it = &s.items[((size_t)w * kMelMaxItems + i) * 8];
      const Part& pt = per[w][i];
      it[0] = pt.g; it[1] = pt.b0; it[2] = pt.nb; it[3] = pt.role; it[4] = pt.slot; it[5] = pt.nslots;
    }
  }
}

static void build_mel_taps(const afx_params& p, const std::vector<float>& W, MelTaps& t) {
  // Filters are taken eight at a time (an "oct"): in k_frames2's mel walk lane (pair, j) accumulates filter
  // 8 * oct + j for the two frames of a frame pair, so the eight filters of an oct are padded with zero weights
  // to the oct's longest tap count (a multiple of 4).
  const int M = p.n_mels, NB = p.n_fft / 2 + 1, NO = (M + 7) / 8;
  t.usable = NO <= kMelMaxOcts;
  t.taps.clear(); t.meta.assign((size_t)NO * 8, 0);
  for (int o = 0; o < NO; ++o) {
    int first[8], nnz[8], n4 = 1;                      // the kernel always accumulates the first batch
    for (int j = 0; j < 8; ++j) {
      const int m = 8 * o + j;
      first[j] = 0; nnz[j] = 0;
      if (m >= M) continue;
      int f = -1, l = -1;
      for (int k = 0; k < NB; ++k) if (W[(size_t)m * NB + k] != 0.f) { if (f < 0) f = k; l = k; }
      if (f >= 0) { first[j] = f; nnz[j] = l - f + 1; }
      n4 = std::max(n4, (nnz[j] + 3) / 4);
    }
    for (int j = 0; j < 8; ++j) {
      // the padded taps of every filter must stay inside the spectrum buffer's pad rows
      if (first[j] + 4 * n4 - 1 > NB + kPbPadRows - 1) t.usable = false;
      const int m = 8 * o + j;
      const int woff = (int)t.taps.size();
      if (n4 > 31 || woff >= (1 << 16)) t.usable = false;
      t.meta[m] = first[j] | (n4 << 10) | (woff << 15);
      for (int i = 0; i < 4 * n4; ++i)
        t.taps.push_back((m < M && i < nnz[j]) ? W[(size_t)m * NB + first[j] + i] : 0.f);
    }
  }
  if (t.taps.empty()) t.taps.push_back(0.f);
}


// k_frames3's mel schedule (afx_frames3.h).  Filters sorted by tap count are cut into rounds of 64 / width filters;
// a dynamic programme picks the widths: a round costs its batches (4 reads + 1 weight read + 4 packed FMAs each)
// plus a fixed finish (reduction, two logs, two stores).
void build_f3_mel(const std::vector<float>& W, int M, int NB, int max_slot, int lanes, int align, HostF3Mel& out) {
  out = HostF3Mel();
  std::vector<int> first(M, 0), nnz(M, 0), order(M);
  for (int m = 0; m < M; ++m) {
    int f = -1, l = -1;
    for (int k = 0; k < NB; ++k) if (W[(size_t)m * NB + k] != 0.f) { if (f < 0) f = k; l = k; }
    if (f >= 0) { first[m] = f; nnz[m] = l - f + 1; }
    order[m] = m;
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nnz[a] < nnz[b]; });
  const int INF = 1 << 28;
  // dp[r][i]: cheapest way to place the first i filters (sorted) in r rounds
  std::vector<std::vector<int>> dp(kF3KernelRounds + 1, std::vector<int>(M + 1, INF)), from_w = dp, from_i = dp;
  dp[0][0] = 0;
  // a lane's first bin is a multiple of `align` (16-byte reads: two (A, B) bins of a frame pair, or four bins of one
  // frame): a filter starting in between takes the taps before it as zeros
  for (int m = 0; m < M; ++m) if (nnz[m] > 0 && (first[m] % align)) { const int d = first[m] % align; first[m] -= d; nnz[m] += d; }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nnz[a] < nnz[b]; });
  auto batches = [&](int i, int j, int w) {            // filters [i, j) at width w
    const int maxt = nnz[order[j - 1]];
    const int per = (maxt + w - 1) / w;
    return std::max(1, (per + 3) / 4);
  };
  for (int r = 0; r < kF3KernelRounds; ++r)
    for (int i = 0; i < M; ++i) {
      if (dp[r][i] >= INF) continue;
      for (int w = 1; w <= 8; w *= 2) {
        if (w > lanes) continue;
        const int j = std::min(M, i + lanes / w);
        const int nb = batches(i, j, w);
        if (nb > kF3MaxBatches) continue;
        const int lg = w == 1 ? 0 : (w == 2 ? 1 : (w == 4 ? 2 : 3));
        const int c = dp[r][i] + 9 * nb + 16 + 4 * lg;
        if (c < dp[r + 1][j]) { dp[r + 1][j] = c; from_w[r + 1][j] = w; from_i[r + 1][j] = i; }
      }
    }
  int best_r = -1;
  for (int r = 1; r <= kF3KernelRounds; ++r) if (dp[r][M] < INF && (best_r < 0 || dp[r][M] < dp[best_r][M])) best_r = r;
  if (best_r < 0) return;
  std::vector<int> cut_i(best_r), cut_w(best_r);
  for (int r = best_r, j = M; r >= 1; --r) { cut_i[r - 1] = from_i[r][j]; cut_w[r - 1] = from_w[r][j]; j = from_i[r][j]; }
  out.rounds = best_r;
  out.meta.assign((size_t)best_r * 64, 0);
  for (int r = 0; r < best_r; ++r) {
    const int i0 = cut_i[r], i1 = r + 1 < best_r ? cut_i[r + 1] : M, w = cut_w[r];
    const int nb = batches(i0, i1, w), S = 4 * nb;
    out.nb[r] = nb; out.width[r] = w; out.woff[r] = (int32_t)out.w.size();
    out.w.resize(out.w.size() + (size_t)nb * 64 * 4, 0.f);
    float* wr = out.w.data() + out.woff[r];
    // Width-1 rounds: which lane takes which filter is free, and so is a downward shift of a lane's first bin while its
    // taps still fit -- used to keep the 16-byte spectrum reads free of LDS bank conflicts.  A ds_read_b128 is served
    // in four groups of 16 lanes (MI355X_MICROARCH.md, LDS table); a lane's read covers one of 16 four-bank slots,
    // slot = (first bin / 2 + 2 i + h) mod 16 at step (i, h), so a group is conflict-free for the whole walk iff its
    // lanes start on 16 different slots: a perfect matching of filters to (group, slot) cells (Kuhn's algorithm).
    std::vector<int> lane_of(i1 - i0), shift_of(i1 - i0, 0);
    for (int i = i0; i < i1; ++i) lane_of[i - i0] = (i - i0) * w;
    if (w == 1) {
      static const int glanes[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                        {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                        {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                        {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
      const int n = i1 - i0, ngrp = lanes / 16;
      std::vector<int> cell_item(64, -1), item_cell(n, -1);
      auto max_shift = [&](int it) {                       // slots (`align` bins) the first bin may move down
        const int m = order[i0 + it];
        return std::max(0, std::min(first[m] / align, (S - nnz[m]) / align));
      };
      std::vector<char> seen;
      std::function<bool(int)> place = [&](int it) -> bool {
        const int m = order[i0 + it], s0 = first[m] / align;
        for (int d = 0; d <= max_shift(it); ++d)
          for (int g = 0; g < ngrp; ++g) {
            const int c = g * 16 + ((s0 - d) & 15);
            if (seen[c]) continue;
            seen[c] = 1;
            if (cell_item[c] < 0 || place(cell_item[c])) { cell_item[c] = it; item_cell[it] = c; return true; }
          }
        return false;
      };
      std::vector<int> by_slack(n);
      for (int it = 0; it < n; ++it) by_slack[it] = it;
      std::stable_sort(by_slack.begin(), by_slack.end(), [&](int a, int b) { return max_shift(a) < max_shift(b); });
      for (int it : by_slack) { seen.assign(64, 0); place(it); }
      // lanes: the k-th filled cell of group g sits on that group's k-th lane; unmatched filters take what is left
      std::vector<char> lane_used(64, 0);
      int fill[4] = {0, 0, 0, 0};
      for (int c = 0; c < 64; ++c) {
        const int it = cell_item[c];
        if (it < 0) continue;
        const int g = c / 16, m = order[i0 + it];
        lane_of[it] = glanes[g][fill[g]++];
        lane_used[lane_of[it]] = 1;
        int d = 0;
        while ((((first[m] / align) - d) & 15) != (c & 15)) ++d;
        shift_of[it] = d;
      }
      for (int it = 0; it < n; ++it)
        if (item_cell[it] < 0) {
          int l = 0;
          while (lane_used[l]) ++l;
          lane_of[it] = l; lane_used[l] = 1; shift_of[it] = 0;
        }
    }
    for (int i = i0; i < i1; ++i) {
      const int m = order[i];
      for (int q = 0; q < w; ++q) {
        const int lane = lane_of[i - i0] + q;
        int bin0 = first[m] - align * shift_of[i - i0] + q * S;
        const bool live = nnz[m] > 0 && bin0 < first[m] + nnz[m];
        if (!live) bin0 = 0;
        if (bin0 + S - 1 > max_slot) return;          // a padded tap would leave the image: unusable
        out.meta[(size_t)r * 64 + lane] = bin0 | (m << 11) | ((q == 0) ? (1 << 20) : 0);
        for (int t = 0; t < S; ++t) {
          const int k = bin0 + t;
          const float v = (live && k >= first[m] && k < first[m] + nnz[m] && k < NB) ? W[(size_t)m * NB + k] : 0.f;
          wr[((size_t)(t >> 2) * 64 + lane) * 4 + (t & 3)] = v;
        }
      }
    }
  }
  if (lanes == 32)            // two frame pairs per wave: the upper half-wave walks the same schedule on its own image
    for (int r = 0; r < best_r; ++r)
      for (int l = 0; l < 32; ++l) {
        out.meta[(size_t)r * 64 + 32 + l] = out.meta[(size_t)r * 64 + l];
        for (int bt = 0; bt < out.nb[r]; ++bt)
          for (int c = 0; c < 4; ++c)
            out.w[out.woff[r] + ((size_t)bt * 64 + 32 + l) * 4 + c] = out.w[out.woff[r] + ((size_t)bt * 64 + l) * 4 + c];
      }
  out.usable = true;
}

static void build_dct_blocks(const afx_params& p, const std::vector<float>& D, DctBlocks& d) {
  const int M = p.n_mels, K = p.n_mfcc;
  const int C = (K + 15) / 16, NI = M / 4;
  d.n_cgroups = C;
  d.A.assign((size_t)C * NI * 64, 0.f);
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < NI; ++i)
      for (int l = 0; l < 64; ++l) {
        const int k = 16 * c + (l & 15), m = 4 * i + (l >> 4);
        if (k < K) d.A[((size_t)c * NI + i) * 64 + l] = D[(size_t)k * M + m];
      }
  // k_dct16: lane (f, q) fetches filters 16 s + 4 q + {0..3} of frame f with one 16-byte load and feeds component c
  // to MFMA (s, c), whose k index q therefore stands for filter 16 s + 4 q + c.  One image per group of 16 coefficients.
  d.P.clear();
  if (M % 16 == 0 && C <= 3) {
    const int S = M / 16;
    d.P.assign((size_t)C * S * 4 * 64, 0.f);
    for (int g = 0; g < C; ++g)
      for (int s = 0; s < S; ++s)
        for (int c = 0; c < 4; ++c)
          for (int l = 0; l < 64; ++l) {
            const int k = 16 * g + (l & 15), m = 16 * s + 4 * (l >> 4) + c;
            if (k < K) d.P[(((size_t)g * S + s) * 4 + c) * 64 + l] = D[(size_t)k * M + m];
          }
  }
  if (d.P.empty()) d.P.push_back(0.f);
}

void build_host_tables(const afx_params& p, HostTables& t) {
  const int N = p.n_fft, N2 = N / 2, M = p.n_mels, K = p.n_mfcc;
  t.window.resize(N);
  for (int n = 0; n < N; ++n) {
    const double c = std::cos(2.0 * kPi * (double)n / (double)N);
    t.window[n] = (float)(p.window == AFX_WINDOW_HANN ? 0.5 - 0.5 * c : 0.54 - 0.46 * c);
  }
  std::vector<double> mel_f;
  build_mel_dense(p, t.mel_dense, mel_f);
  build_mel_blocks(p, t.mel_dense, mel_f, t.mel);
  build_mel_taps(p, t.mel_dense, t.taps);
  // wave-level frame kernels: 1024 -> one frame pair per wave (bins as (A, B) pairs), 2048 -> one frame per wave (bins
  // as floats), 512 -> two frame pairs per wave, one per half-wave with half the image each
  if (N == 1024) build_f3_mel(t.mel_dense, M, N / 2 + 1, kF3ExFloats / 2 - 1, 64, 2, t.f3mel);
  else if (N == 2048) build_f3_mel(t.mel_dense, M, N / 2 + 1, kF3ExFloats - 1, 64, 4, t.f3mel);
  else if (N == 512) build_f3_mel(t.mel_dense, M, N / 2 + 1, kF3ExFloats / 4 - 1, 32, 2, t.f3mel);
  t.dct.resize((size_t)K * M);
  for (int k = 0; k < K; ++k) {
    const double s = k == 0 ? std::sqrt(1.0 / M) : std::sqrt(2.0 / M);
    for (int m = 0; m < M; ++m)
      t.dct[(size_t)k * M + m] = (float)(s * std::cos(kPi * k * (2.0 * m + 1.0) / (2.0 * M)));
    if (p.lifter > 0.f) {         // librosa.feature.mfcc(lifter=L): row n = k + 1 scaled by 1 + (L / 2) sin(pi n / L), folded into the DCT row
      const float li = (float)std::sin(kPi * (double)(k + 1) / (double)p.lifter);
      const float f = 1.0f + (p.lifter * 0.5f) * li;
      for (int m = 0; m < M; ++m) t.dct[(size_t)k * M + m] *= f;
    }
  }
  build_dct_blocks(p, t.dct, t.dctb);
  t.tw.resize((size_t)2 * N2); t.post.resize((size_t)2 * N2);
  for (int n = 0; n < N2; ++n) {
    const double a = -2.0 * kPi * (double)n / (double)N2;
    t.tw[2 * n] = (float)std::cos(a); t.tw[2 * n + 1] = (float)std::sin(a);
    const double b = -2.0 * kPi * (double)n / (double)N;
    t.post[2 * n] = (float)std::cos(b); t.post[2 * n + 1] = (float)std::sin(b);
  }
}

}  // namespace afx

extern "C" void afx_default_params(afx_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->sr = 22050; p->n_fft = 1024; p->hop = 256; p->n_mfcc = 13; p->n_mels = 128;
  p->window = AFX_WINDOW_HAMMING; p->preemph = 0.97f; p->trim_top_db = 30.f;
  p->trim_frame = 2048; p->trim_hop = 512; p->top_db = 80.f; p->amin = 1e-10f;
  p->delta_width = 9;
  p->fmin = 0.f; p->fmax = 0.f; p->htk = 0; p->lifter = 0.f;
}

extern "C" int afx_build_mel_schedule(const afx_params* p, int32_t* info, float* weights, int32_t* meta) {
  if (!p) { afx::set_error("afx_build_mel_schedule: null params"); return AFX_ERR_INVALID; }
  std::string msg;
  int st = afx::validate_params(*p, msg);
  if (st != AFX_OK) { afx::set_error(msg); return st; }
  afx::HostTables t;
  afx::build_host_tables(*p, t);
  const afx::HostF3Mel& f = t.f3mel;
  if (!f.usable) { afx::set_error("no wave-level mel schedule for this configuration"); return AFX_ERR_UNSUPPORTED; }
  if (info) {
    info[0] = f.rounds; info[1] = (int32_t)f.w.size();
    for (int r = 0; r < afx::kF3MaxRounds; ++r) { info[2 + 3 * r] = f.nb[r]; info[3 + 3 * r] = f.width[r]; info[4 + 3 * r] = f.woff[r]; }
  }
  if (weights) std::memcpy(weights, f.w.data(), f.w.size() * sizeof(float));
  if (meta) std::memcpy(meta, f.meta.data(), f.meta.size() * sizeof(int32_t));
  return AFX_OK;
}

extern "C" int afx_build_tables(const afx_params* p, float* window, float* mel_dense, float* dct) {
  if (!p) { afx::set_error("afx_build_tables: null params"); return AFX_ERR_INVALID; }
  std::string msg;
  int st = afx::validate_params(*p, msg);
  if (st != AFX_OK) { afx::set_error(msg); return st; }
  afx::HostTables t;
  afx::build_host_tables(*p, t);
  if (window) std::memcpy(window, t.window.data(), t.window.size() * sizeof(float));
  if (mel_dense) std::memcpy(mel_dense, t.mel_dense.data(), t.mel_dense.size() * sizeof(float));
  if (dct) std::memcpy(dct, t.dct.data(), t.dct.size() * sizeof(float));
  return AFX_OK;
}
