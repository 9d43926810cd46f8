// Host-side tables of the pYIN stage (librosa.pyin defaults as the reference uses them,
// audio_feature_extraction_toolkit/core/feature_extractor.py:87-94).
#include <algorithm>
#include <cmath>
#include <string>

#include "afx_f0.h"

namespace afx {

bool build_f0_tables(int sr, int n_fft, int hop, double fmin, double fmax, HostF0Tables& t, std::string& why) {
  F0Params& p = t.p;
  if (!(fmin > 0.0) || !(fmax > fmin)) { why = "f0_min/f0_max must satisfy 0 < f0_min < f0_max"; return false; }
  p.n_fft = n_fft; p.hop = hop; p.W = n_fft / 2;
  p.sr = (double)sr; p.fmin = fmin;
  p.min_period = (int)std::floor((double)sr / fmax);
  p.max_period = std::min((int)std::ceil((double)sr / fmin), n_fft - p.W - 1);
  if (p.min_period < 1 || p.max_period <= p.min_period + 1) { why = "f0 range leaves fewer than three lags"; return false; }
  p.n_tau = p.max_period + 1;
  p.n_tau_pad = (p.n_tau + 63) / 64 * 64;
  p.n_lag = p.max_period - p.min_period + 1;
  p.R = (p.n_tau + 63) / 64;
  p.slots = (p.n_lag + 63) / 64;
  if (p.R > 16 || p.slots > 16) { why = "frame_length too large for the f0 kernels"; return false; }
  p.cap = ((p.n_lag + 1) / 2 + 1 + 7) / 8 * 8;
  const int bins_per_semitone = (int)std::ceil(1.0 / 0.1);                 // resolution = 0.1
  p.bins_per_octave = 12.0 * bins_per_semitone;
  p.n_bins = (int)std::floor(12 * bins_per_semitone * std::log2(fmax / fmin)) + 1;
  if (p.n_bins < 2 || p.n_bins > 4096) { why = "unsupported number of pitch bins"; return false; }
  const double rate = 35.92 * 12.0 * (double)hop / (double)sr;             // max_transition_rate
  const int max_semitones = (int)std::nearbyint(rate);                     // Python round(): half to even
  const int width = max_semitones * bins_per_semitone + 1;
  p.band = width / 2;
  if (p.band < 1 || 2 * p.band + 1 > p.n_bins) { why = "transition band wider than the pitch range"; return false; }
  p.tiny = 2.2250738585072014e-308;
  p.c0 = std::log(p.tiny);
  p.no_trough_prob = 0.01;
  p.debug = 0;
  // frames per block of k_f0_energy: the span and the history must fit 150 KB of LDS
  p.epb = 0;
  for (int e : {64, 32, 16}) {
    const size_t span = (size_t)(e - 1) * hop + p.W + p.n_tau;
    const size_t bytes = (span + span / hop + 8) * 4 + (size_t)p.n_tau * (e + 1) * 4;
    if (bytes <= 150 * 1024) { p.epb = e; break; }
  }
  if (!p.epb) { why = "hop_length too large for the f0 energy kernel"; return false; }

  t.thr.resize(kF0Thresholds + 1);
  const double step = 1.0 / kF0Thresholds;
  for (int k = 0; k <= kF0Thresholds; ++k) t.thr[k] = (double)k * step;    // np.linspace(0, 1, 101)
  t.thr[kF0Thresholds] = 1.0;
  // Beta(2, 18) CDF in closed form: I_x(2, 18) = 1 - (1 - x)^18 (1 + 18 x)
  std::vector<double> cdf(kF0Thresholds + 1);
  for (int k = 0; k <= kF0Thresholds; ++k) {
    const double x = t.thr[k];
    cdf[k] = 1.0 - std::pow(1.0 - x, 18.0) * (1.0 + 18.0 * x);
  }
  t.beta.resize(kF0Thresholds);
  t.cumbeta.assign(kF0Thresholds + 1, 0.0);
  for (int k = 0; k < kF0Thresholds; ++k) {
    t.beta[k] = cdf[k + 1] - cdf[k];
    t.cumbeta[k + 1] = t.cumbeta[k] + t.beta[k];
  }
  // scipy.stats.boltzmann.pmf(k, 2, N) = (1 - e^-2) / (1 - e^-2N) * e^-2k
  t.bfact.assign(p.cap + 1, 0.0); t.bexp.assign(p.cap + 1, 0.0);
  for (int n = 0; n <= p.cap; ++n) {
    t.bexp[n] = std::exp(-2.0 * n);
    if (n > 0) t.bfact[n] = (1.0 - std::exp(-2.0)) / (1.0 - std::exp(-2.0 * n));
  }
  // librosa.sequence.transition_local(n_bins, width, 'triangle', wrap=False) x transition_loop(2, 0.99):
  // a source row is the triangle centred on it, cut at the ends of the pitch range, normalised to 1.
  std::vector<double> win(width);
  for (int k = 0; k < width; ++k) {
    const int n = k < (width + 1) / 2 ? k + 1 : width - k;                 // scipy.signal.windows.triang, odd M
    win[k] = 2.0 * n / (width + 1.0);
  }
  const int NC = 2 * p.band + 1;                                           // row classes: 0 interior, 1..band low edge, band+1..2band high edge
  t.lt.assign((size_t)2 * NC * width, p.c0);
  const double sw[2] = {1.0 - 0.01, 0.01};                                 // stay, switch
  for (int rc = 0; rc < NC; ++rc) {
    // class -> a representative source bin b: d = j - b + band is cut when j < 0 or j >= n_bins
    int dlo = 0, dhi = width - 1;
    if (rc >= 1 && rc <= p.band) dlo = p.band - (rc - 1);                  // b = rc - 1 < band
    if (rc > p.band) dhi = p.band + (rc - p.band - 1);                     // b = n_bins - 1 - (rc - band - 1)
    double rowsum = 0.0;
    for (int d = dlo; d <= dhi; ++d) rowsum += win[d];
    for (int v = 0; v < 2; ++v)
      for (int d = dlo; d <= dhi; ++d)
        t.lt[((size_t)v * NC + rc) * width + d] = std::log(sw[v] * (win[d] / rowsum) + p.tiny);
  }
  t.ltw.resize((size_t)2 * width);
  for (int e = 0; e < width; ++e)
    for (int v = 0; v < 2; ++v) t.ltw[(size_t)v * width + e] = t.lt[(size_t)v * NC * width + (2 * p.band - e)];
  t.freqs.resize(p.n_bins);
  for (int b = 0; b < p.n_bins; ++b) t.freqs[b] = fmin * std::pow(2.0, (double)b / p.bins_per_octave);
  return true;
}

}  // namespace afx
