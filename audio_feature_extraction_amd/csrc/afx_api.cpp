// C-ABI layer of libafx.so: contexts, plans, workspace management and the
// batch entry point that strings the five kernels together on one HIP stream.
// See include/afx.h for the contract and the reference call sites replaced.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "afx_device.h"
#include "afx_devenv.h"
#include "afx_f0.h"
#include "afx_frames3.h"
#include "afx_internal.h"

namespace afx {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace afx

using namespace afx;

struct afx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
};

struct afx_plan {
  afx_ctx* ctx = nullptr;
  int device = 0;             // copy: the plan may be destroyed after its context by a garbage-collected binding
  afx_params p{};
  KParams kp{};
  HostTables ht;
  DevTables dt{};
  F3Tables f3{};              // k_frames3 (n_fft 1024 / hop 256): mel schedule + twiddle source
  bool use_f3 = false;
  std::vector<void*> table_allocs;
  DevBuf samples, clips, info, blocks, bsum, logmel, rms, mfcc, frames, frame_offs, stamps;     // statistics go straight to h_pin
  DevBuf blocks_spec, blockmax, items, n_items;      // speculative pipeline (k_frames3 before the trim decision)
  std::vector<BlockDesc> h_blocks;     // AFX_HOST_BLOCKS (A/B) only: the block list is built by k_build_blocks3
  // pinned staging of the clip records: the upload is a true asynchronous copy, and the event tells when the
  // staging may be rewritten (no stream synchronisation on a batch whose clip lengths are new)
  ClipDesc* h_clips_pin = nullptr;
  size_t h_clips_pin_cap = 0;
  hipEvent_t clips_ev = nullptr;
  bool clips_ev_pending = false;
  // extract_f0 (pYIN): tables for the last (fmin, fmax) used and the stage's workspace
  bool f0_ready = false;
  double f0_fmin = 0.0, f0_fmax = 0.0;
  HostF0Tables f0_ht;
  F0Tables f0_dt{};
  std::vector<void*> f0_allocs;
  DevBuf f0_in, f0_ysig, f0_energy, f0_cnt, f0_vp, f0_bin, f0_prob, f0_lprob, f0_lu, f0_ptr, f0_best, f0_states, f0_stats, f0_out, f0_offs;
  // cached per-batch descriptors
  std::vector<int64_t> c_off, c_len;
  std::vector<ClipDesc> h_clips;
  std::vector<int64_t> h_rebased;   // per-frame output offsets of the pending chunk (source of an asynchronous upload: lives until collect)
  int nblocks = 0, max_tblocks = 0, max_tmax = 0;
  int64_t total_tpad = 0, total_tblk = 0;
  // pinned staging for the small per-call results (a device-to-pageable copy is staged and synchronous)
  void* h_pin = nullptr;
  void* h_pin_dev = nullptr;       // the same block as the device addresses it
  size_t h_pin_cap = 0;
  int info_clean_n = 0;            // leading clip records (and the counters) known to be zero on the stream
  // a submitted, not yet collected chunk (afx_extract_submit / afx_extract_collect; afx_extract_batch = both, per chunk)
  struct Pending {
    bool active = false;
    int n = 0;
    size_t stats_bytes = 0;
    float* out_stats = nullptr; int32_t* out_status = nullptr; int64_t* out_trim = nullptr; int32_t* out_nframes = nullptr;
  } pend;
  hipEvent_t done = nullptr;       // recorded behind the chunk's last copy: collect waits for this chunk, not for the stream
  volatile unsigned* flag = nullptr;   // host word the device stores the chunk's sequence number to (behind the same copy)
  unsigned* flag_dev = nullptr;
  unsigned seq = 0;
  // timing
  bool timing = false;
  bool timing_frames_only = false;   // afx_plan_set_timing(plan, 2): events around the frame kernel only
  hipEvent_t ev[AFX_K_COUNT][2] = {};
  bool ev_ready = false;
  double ms_sum[AFX_K_COUNT] = {};
  int32_t launches[AFX_K_COUNT] = {};
  std::vector<std::pair<double, double>> spans[AFX_K_COUNT];   // (start, end) ms on the device's common clock, newest kMaxSpans
  int n_cu = 256;
};

// Developer switches (afx_devenv.h): read once per process, never on the per-call path.
namespace afx {
DevEnv::DevEnv() {
  if (const char* v = getenv("AFX_DEBUG_SKIP")) debug_skip = atoi(v) & 127;
  if (const char* v = getenv("AFX_F0_DEBUG")) f0_debug = atoi(v);
  stamps = getenv("AFX_DEBUG_STAMPS") != nullptr;
  f3_debug = getenv("AFX_F3_DEBUG") != nullptr;
  no_spec = getenv("AFX_NO_SPEC") != nullptr;
  no_tickets = getenv("AFX_NO_TICKETS") != nullptr;      // A/B: equal static shares in the speculative frame launch
  no_frames3 = getenv("AFX_NO_FRAMES3") != nullptr;
  no_frames3s = getenv("AFX_NO_FRAMES3S") != nullptr;
  no_frames3d = getenv("AFX_NO_FRAMES3D") != nullptr;
  f3_generic_mel = getenv("AFX_F3_GENERIC_MEL") != nullptr;
  generic_1024 = getenv("AFX_GENERIC_1024") != nullptr;
  no_dct16l = getenv("AFX_NO_DCT16L") != nullptr;
  host_blocks = getenv("AFX_HOST_BLOCKS") != nullptr;
  no_fused_tail = getenv("AFX_NO_FUSED_TAIL") != nullptr;
  if (const char* v = getenv("AFX_F3_WAVES")) f3_waves = atoi(v);
  if (const char* v = getenv("AFX_TAIL_MODE")) tail_mode = atoi(v);
  if (const char* v = getenv("AFX_TEST_CHUNK_CLIPS")) chunk_clips = std::max(1, std::min(32768, atoi(v)));
  if (const char* v = getenv("AFX_TEST_F0_CHUNK_FRAMES")) f0_chunk_frames = std::max<int64_t>(64, atoll(v));
  f0_dump = getenv("AFX_F0_DUMP");
}
const DevEnv& dev_env() { static const DevEnv e; return e; }
}  // namespace afx

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess) {                                                           \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e__));                   \
      return AFX_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

static int ensure(DevBuf& b, size_t bytes) {
  if (bytes <= b.cap && b.p) return AFX_OK;
  if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
  size_t want = std::max<size_t>(bytes + bytes / 8, 256);
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    set_error(std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    b.p = nullptr;
    return e == hipErrorOutOfMemory ? AFX_ERR_NOMEM : AFX_ERR_HIP;
  }
  b.cap = want;
  return AFX_OK;
}

static void release(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr; b.cap = 0;
}

extern "C" int afx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int afx_init(int device, afx_ctx** out) {
  if (!out) { set_error("afx_init: null out"); return AFX_ERR_INVALID; }
  *out = nullptr;
  int n = afx_device_count();
  if (n <= 0) { set_error("no HIP device visible"); return AFX_ERR_NO_DEVICE; }
  if (device < 0 || device >= n) { set_error("device index out of range"); return AFX_ERR_INVALID; }
  HIP_TRY(hipSetDevice(device));
  afx_ctx* c = new afx_ctx();
  c->device = device;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; set_error(std::string("hipStreamCreate: ") + hipGetErrorString(e)); return AFX_ERR_HIP; }
  *out = c;
  return AFX_OK;
}

extern "C" void afx_destroy(afx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int afx_malloc(afx_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) { set_error("afx_malloc: null argument"); return AFX_ERR_INVALID; }
  HIP_TRY(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(out, bytes ? bytes : 1);
  if (e != hipSuccess) { set_error(std::string("hipMalloc: ") + hipGetErrorString(e)); return e == hipErrorOutOfMemory ? AFX_ERR_NOMEM : AFX_ERR_HIP; }
  return AFX_OK;
}

extern "C" int afx_free(afx_ctx* ctx, void* d) {
  if (!ctx) { set_error("afx_free: null ctx"); return AFX_ERR_INVALID; }
  HIP_TRY(hipSetDevice(ctx->device));
  if (d) HIP_TRY(hipFree(d));
  return AFX_OK;
}

extern "C" int afx_host_alloc(afx_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) { set_error("afx_host_alloc: null argument"); return AFX_ERR_INVALID; }
  *out = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable);
  if (e != hipSuccess) { set_error(std::string("hipHostMalloc: ") + hipGetErrorString(e)); *out = nullptr; return e == hipErrorOutOfMemory ? AFX_ERR_NOMEM : AFX_ERR_HIP; }
  return AFX_OK;
}

extern "C" int afx_host_free(afx_ctx* ctx, void* h) {
  if (!ctx) { set_error("afx_host_free: null ctx"); return AFX_ERR_INVALID; }
  HIP_TRY(hipSetDevice(ctx->device));
  if (h) HIP_TRY(hipHostFree(h));
  return AFX_OK;
}

extern "C" int afx_memcpy_h2d(afx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (!dst && bytes) || (!src && bytes)) { set_error("afx_memcpy_h2d: null argument"); return AFX_ERR_INVALID; }
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return AFX_OK;
}

extern "C" int afx_memcpy_d2h(afx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (!dst && bytes) || (!src && bytes)) { set_error("afx_memcpy_d2h: null argument"); return AFX_ERR_INVALID; }
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return AFX_OK;
}

extern "C" int afx_synchronize(afx_ctx* ctx) {
  if (!ctx) { set_error("afx_synchronize: null ctx"); return AFX_ERR_INVALID; }
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return AFX_OK;
}

template <typename T>
static int upload(afx_plan* pl, const T* host, size_t count, const T** dev) {
  void* d = nullptr;
  size_t bytes = std::max<size_t>(count * sizeof(T), 16);
  hipError_t e = hipMalloc(&d, bytes);
  if (e != hipSuccess) { set_error(std::string("hipMalloc(table): ") + hipGetErrorString(e)); return AFX_ERR_HIP; }
  pl->table_allocs.push_back(d);
  HIP_TRY(hipMemset(d, 0, bytes));
  if (count) HIP_TRY(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
  *dev = (const T*)d;
  return AFX_OK;
}

extern "C" int afx_plan_create(afx_ctx* ctx, const afx_params* p, afx_plan** out) {
  if (!ctx || !p || !out) { set_error("afx_plan_create: null argument"); return AFX_ERR_INVALID; }
  *out = nullptr;
  std::string msg;
  int st = validate_params(*p, msg);
  if (st != AFX_OK) { set_error(msg); return st; }
  if (p->n_fft > 2048) { set_error("frame_length > 2048 is not supported by the LDS-resident FFT"); return AFX_ERR_UNSUPPORTED; }
  HIP_TRY(hipSetDevice(ctx->device));
  afx_plan* pl = new afx_plan();
  pl->ctx = ctx; pl->device = ctx->device; pl->p = *p;
  build_host_tables(*p, pl->ht);
  const size_t lds = frames_lds_bytes(p->n_fft, p->hop);
  if (lds == 0 || lds > 160 * 1024) {
    delete pl;
    set_error("frame_length/hop_length combination needs more than 160 KiB of LDS per workgroup (hop too large)");
    return AFX_ERR_UNSUPPORTED;
  }
  KParams& kp = pl->kp;
  kp.n_fft = p->n_fft; kp.hop = p->hop; kp.n_mels = p->n_mels; kp.n_mfcc = p->n_mfcc;
  kp.trim_frame = p->trim_frame; kp.trim_hop = p->trim_hop;
  kp.preemph_b1 = (float)(-(double)p->preemph);     // np.asarray([1.0, -coef], dtype=float32)
  kp.trim_top_db = p->trim_top_db; kp.top_db = p->top_db; kp.amin = p->amin;
  kp.flags = 0; kp.fmt = AFX_FMT_F32; kp.rms_sub = 0;
  int rc = AFX_OK;
  const HostTables& t = pl->ht;
  const int4* grp4 = nullptr;
  const float4* coef4 = nullptr;
  const int4* items4 = nullptr;
  if ((rc = upload(pl, t.window.data(), t.window.size(), &pl->dt.window)) != AFX_OK ||
      (rc = upload(pl, t.tw.data(), t.tw.size(), &pl->dt.tw)) != AFX_OK ||
      (rc = upload(pl, t.post.data(), t.post.size(), &pl->dt.post)) != AFX_OK ||
      (rc = upload(pl, reinterpret_cast<const float4*>(t.mel.coef.data()), t.mel.coef.size() / 4, &coef4)) != AFX_OK ||
      (rc = upload(pl, t.mel.koff.data(), t.mel.koff.size(), &pl->dt.mel_koff)) != AFX_OK ||
      (rc = upload(pl, reinterpret_cast<const int4*>(t.mel.grp.data()), t.mel.grp.size() / 4, &grp4)) != AFX_OK ||
      (rc = upload(pl, reinterpret_cast<const int4*>(t.mel.items.data()), t.mel.items.size() / 4, &items4)) != AFX_OK ||
      (rc = upload(pl, t.taps.taps.data(), t.taps.taps.size(), &pl->dt.mel_taps)) != AFX_OK ||
      (rc = upload(pl, t.taps.meta.data(), t.taps.meta.size(), &pl->dt.mel_meta)) != AFX_OK ||
      (rc = upload(pl, t.dctb.A.data(), t.dctb.A.size(), &pl->dt.dctA)) != AFX_OK ||
      (rc = upload(pl, t.dctb.P.data(), t.dctb.P.size(), &pl->dt.dctP)) != AFX_OK) {
    afx_plan_destroy(pl);
    return rc;
  }
  pl->dt.mel_grp = grp4;
  pl->dt.mel_coef = coef4;
  pl->dt.mel_items = items4;
  for (int w = 0; w < 4; ++w) pl->dt.mel_item_cnt[w] = t.mel.item_cnt[w];
  pl->dt.mel_n_slots = t.mel.n_slots;
  pl->dt.mel_ntaps = t.taps.usable ? (int32_t)t.taps.taps.size() : 0;
  pl->dt.n_groups = t.mel.n_groups;
  if (t.f3mel.usable) {
    if ((rc = upload(pl, t.f3mel.w.data(), t.f3mel.w.size(), &pl->f3.mel_w)) != AFX_OK ||
        (rc = upload(pl, t.f3mel.meta.data(), t.f3mel.meta.size(), &pl->f3.mel_meta)) != AFX_OK) {
      afx_plan_destroy(pl);
      return rc;
    }
    // twiddle sources: post = exp(-2 pi i k / n_fft), k < n_fft / 2;  tw = exp(-2 pi i n / (n_fft / 2)), n < n_fft / 2
    pl->f3.window = pl->dt.window;
    pl->f3.w1024 = p->n_fft == 2048 ? pl->dt.tw : pl->dt.post;
    pl->f3.w2048 = pl->dt.post;
    pl->f3.mel_rounds = t.f3mel.rounds; pl->f3.mel_wfloats = (int32_t)t.f3mel.w.size();
    pl->f3.mel_all_own = 1;                        // every lane of every round is the owner of a filter (width-1 rounds, all lanes used)
    for (size_t i = 0; i < t.f3mel.meta.size(); ++i) if (!(t.f3mel.meta[i] & (1 << 20))) pl->f3.mel_all_own = 0;
    pl->f3.mel_own_w1 = 1;
    for (int r = 0; r < t.f3mel.rounds; ++r)
      if (t.f3mel.width[r] == 1)
        for (int l = 0; l < 64; ++l) if (!(t.f3mel.meta[(size_t)r * 64 + l] & (1 << 20))) pl->f3.mel_own_w1 = 0;
    for (int r = 0; r < kF3MaxRounds; ++r)
      pl->f3.mel_rp[r] = (uint32_t)t.f3mel.nb[r] | ((uint32_t)t.f3mel.width[r] << 4) | ((uint32_t)t.f3mel.woff[r] << 8);
  }
  pl->use_f3 = frames3_eligible(kp, pl->f3);
  kp.rms_sub = (pl->use_f3 || frames2_eligible(kp, pl->dt)) ? kp.trim_hop / kp.hop : 0;
  pl->dt.n_cgroups = t.dctb.n_cgroups;
  if (t.dctb.P.size() < 64) pl->dt.dctP = nullptr;
  pl->dt.dctS = nullptr;
  if (p->n_mels % 32 == 0 && p->n_mels <= 128 && p->n_mfcc > 16 && p->n_mfcc <= 48) {
    // A images of the folded DCT (k_tail<., true>): [(g KS + s) 4 + c][lane (i, q)] = dct[coef(g, i)][16 s + 4 q + c], the even
    // coefficients in the first GH groups, the odd ones in the rest; rows beyond n_mfcc are zero
    const int M = p->n_mels, K = p->n_mfcc, KS = M / 32, GH = ((K + 1) / 2 + 15) / 16, NG = 2 * GH;
    std::vector<float> S((size_t)NG * KS * 4 * 64, 0.f);
    for (int g = 0; g < NG; ++g)
      for (int s = 0; s < KS; ++s)
        for (int c = 0; c < 4; ++c)
          for (int l = 0; l < 64; ++l) {
            const int i = l & 15, q = l >> 4, k = 2 * (16 * (g % GH) + i) + g / GH, m = 16 * s + 4 * q + c;
            if (k < K) S[(((size_t)g * KS + s) * 4 + c) * 64 + l] = t.dct[(size_t)k * M + m];
          }
    if ((rc = upload(pl, S.data(), S.size(), &pl->dt.dctS)) != AFX_OK) { afx_plan_destroy(pl); return rc; }
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
    pl->n_cu = prop.multiProcessorCount;
  *out = pl;
  return AFX_OK;
}

extern "C" void afx_plan_destroy(afx_plan* pl) {
  if (!pl) return;
  (void)hipSetDevice(pl->device);
  if (pl->pend.active && pl->ctx) (void)hipStreamSynchronize(pl->ctx->stream);   // a submitted batch still writes the plan's memory
  for (void* d : pl->table_allocs) (void)hipFree(d);
  for (void* q : pl->f0_allocs) (void)hipFree(q);
  release(pl->f0_in); release(pl->f0_ysig); release(pl->f0_energy); release(pl->f0_cnt); release(pl->f0_vp);
  release(pl->f0_bin); release(pl->f0_prob); release(pl->f0_ptr); release(pl->f0_best); release(pl->f0_lprob); release(pl->f0_lu); release(pl->f0_states); release(pl->f0_stats);
  release(pl->f0_out); release(pl->f0_offs);
  release(pl->samples); release(pl->clips); release(pl->info); release(pl->blocks); release(pl->bsum);
  release(pl->logmel); release(pl->rms); release(pl->mfcc); release(pl->frames);
  release(pl->frame_offs); release(pl->stamps);
  release(pl->blocks_spec); release(pl->blockmax); release(pl->items); release(pl->n_items);
  if (pl->h_pin) (void)hipHostFree(pl->h_pin);
  if (pl->h_clips_pin) (void)hipHostFree(pl->h_clips_pin);
  if (pl->clips_ev) (void)hipEventDestroy(pl->clips_ev);
  if (pl->done) (void)hipEventDestroy(pl->done);
  if (pl->flag) (void)hipHostFree((void*)pl->flag);
  if (pl->ev_ready)
    for (int k = 0; k < AFX_K_COUNT; ++k) { (void)hipEventDestroy(pl->ev[k][0]); (void)hipEventDestroy(pl->ev[k][1]); }
  delete pl;
}

// One reference event per device, recorded once: launch intervals of every plan (= every stream) on that device are
// reported against it, so that a caller can merge the intervals of streams that run side by side.
static std::mutex g_ref_mu;
static hipEvent_t g_ref[64] = {};
static bool g_ref_set[64] = {};
constexpr size_t kMaxSpans = 1 << 16;

static int device_ref(afx_plan* pl, hipEvent_t* out) {
  std::lock_guard<std::mutex> lk(g_ref_mu);
  const int d = pl->device;
  if (d < 0 || d >= 64) { set_error("device index out of range for timing"); return AFX_ERR_INVALID; }
  if (!g_ref_set[d]) {
    HIP_TRY(hipEventCreate(&g_ref[d]));
    HIP_TRY(hipEventRecord(g_ref[d], pl->ctx->stream));
    HIP_TRY(hipEventSynchronize(g_ref[d]));
    g_ref_set[d] = true;
  }
  *out = g_ref[d];
  return AFX_OK;
}

extern "C" int afx_plan_set_timing(afx_plan* pl, int enable) {
  if (!pl) { set_error("afx_plan_set_timing: null plan"); return AFX_ERR_INVALID; }
  HIP_TRY(hipSetDevice(pl->device));
  if (enable) { hipEvent_t r; int rc = device_ref(pl, &r); if (rc != AFX_OK) return rc; }
  if (enable && !pl->ev_ready) {
    for (int k = 0; k < AFX_K_COUNT; ++k) { HIP_TRY(hipEventCreate(&pl->ev[k][0])); HIP_TRY(hipEventCreate(&pl->ev[k][1])); }
    pl->ev_ready = true;
  }
  pl->timing = enable != 0;
  pl->timing_frames_only = enable == 2;
  return AFX_OK;
}

extern "C" int afx_plan_get_timings(afx_plan* pl, float* ms, int32_t* launches, int reset) {
  if (!pl) { set_error("afx_plan_get_timings: null plan"); return AFX_ERR_INVALID; }
  for (int k = 0; k < AFX_K_COUNT; ++k) {
    if (ms) ms[k] = (float)pl->ms_sum[k];
    if (launches) launches[k] = pl->launches[k];
    if (reset) { pl->ms_sum[k] = 0.0; pl->launches[k] = 0; pl->spans[k].clear(); }
  }
  return AFX_OK;
}

extern "C" int afx_plan_get_intervals(afx_plan* pl, int slot, double* start_ms, double* end_ms, int cap, int32_t* count) {
  if (!pl || slot < 0 || slot >= AFX_K_COUNT || cap < 0 || !count) { set_error("afx_plan_get_intervals: invalid argument"); return AFX_ERR_INVALID; }
  const auto& v = pl->spans[slot];
  const int n = (int)std::min<size_t>(v.size(), (size_t)cap);
  for (int i = 0; i < n; ++i) {
    if (start_ms) start_ms[i] = v[v.size() - n + i].first;
    if (end_ms) end_ms[i] = v[v.size() - n + i].second;
  }
  *count = (int32_t)v.size();
  return AFX_OK;
}

// Builds (or reuses) the per-clip descriptors and the k_frames block list.
static int prepare_descriptors(afx_plan* pl, const int64_t* offsets, const int64_t* lengths, int n) {
  const bool same = (int)pl->c_len.size() == n &&
                    std::memcmp(pl->c_len.data(), lengths, n * sizeof(int64_t)) == 0 &&
                    std::memcmp(pl->c_off.data(), offsets, n * sizeof(int64_t)) == 0;
  if (same) return AFX_OK;
  const int hop = pl->p.hop, th = pl->p.trim_hop;
  pl->h_clips.resize(n);
  BatchGeom g;
  {
    std::string why;
    if (!build_clip_descs(hop, th, offsets, lengths, n, pl->h_clips.data(), g, why)) { set_error(why); return AFX_ERR_INVALID; }
  }
  const int64_t nblk = g.nblocks;
  pl->total_tpad = g.total_tpad; pl->total_tblk = g.total_tblk; pl->max_tblocks = g.max_tblocks; pl->max_tmax = g.max_tmax;
  pl->nblocks = g.nblocks;
  int rc;
  if ((rc = ensure(pl->clips, n * sizeof(ClipDesc))) != AFX_OK) return rc;
  if ((rc = ensure(pl->blocks, std::max<size_t>((size_t)nblk, 1) * sizeof(BlockDesc))) != AFX_OK) return rc;
  hipStream_t s = pl->ctx->stream;
  // the clip records go through pinned staging of the plan's own: the previous upload from it has normally long
  // completed (a plan has one batch in flight), the event only makes that certain
  if (pl->clips_ev_pending) { HIP_TRY(hipEventSynchronize(pl->clips_ev)); pl->clips_ev_pending = false; }
  if (pl->h_clips_pin_cap < (size_t)n) {
    if (pl->h_clips_pin) (void)hipHostFree(pl->h_clips_pin);
    pl->h_clips_pin = nullptr; pl->h_clips_pin_cap = 0;
    const size_t want = (size_t)n + (size_t)n / 4 + 64;
    HIP_TRY(hipHostMalloc((void**)&pl->h_clips_pin, want * sizeof(ClipDesc), hipHostMallocDefault));
    pl->h_clips_pin_cap = want;
  }
  if (!pl->clips_ev) HIP_TRY(hipEventCreateWithFlags(&pl->clips_ev, hipEventDisableTiming));
  std::memcpy(pl->h_clips_pin, pl->h_clips.data(), (size_t)n * sizeof(ClipDesc));
  HIP_TRY(hipMemcpyAsync(pl->clips.p, pl->h_clips_pin, n * sizeof(ClipDesc), hipMemcpyHostToDevice, s));
  HIP_TRY(hipEventRecord(pl->clips_ev, s));
  pl->clips_ev_pending = true;
  if (pl->use_f3) {
    // the speculative pass's blocks: every absolute 16-frame block of every clip, nothing trimmed yet -- written on the
    // device from the clip records just uploaded (k_build_blocks3): a batch of new clip lengths costs the host one
    // 48-byte record per clip, not a 64-byte record per 16 frames
    const size_t nb_alloc = (size_t)std::max<int64_t>(nblk, 1);
    if ((rc = ensure(pl->blocks_spec, nb_alloc * sizeof(BlockDesc))) != AFX_OK) return rc;
    if ((rc = ensure(pl->blockmax, nb_alloc * sizeof(float))) != AFX_OK) return rc;
    if ((rc = ensure(pl->items, (size_t)n * kF3ItemsPerClip * sizeof(BlockDesc))) != AFX_OK) return rc;
    if ((rc = ensure(pl->n_items, 16)) != AFX_OK) return rc;
    if (!dev_env().host_blocks) {
      HIP_TRY(launch_build_blocks3(s, (const ClipDesc*)pl->clips.p, n, (int)nblk, (BlockDesc*)pl->blocks_spec.p, pl->kp));
    } else {                      // A/B: round 2's host-built list (3.5 MB for 1000 ten-second clips) and its synchronisation
      const int per = pl->kp.rms_sub, half = pl->p.n_fft / 2;
      const int64_t lim = (int64_t)1 << 30;
      pl->h_blocks.resize(nb_alloc);
      for (int i = 0; i < n; ++i) {
        const ClipDesc& c = pl->h_clips[i];
        const int64_t ntb = (c.len + th - 1) / th;
        for (int b = 0; b < c.tpad / kFramesPerBlock; ++b) {
          BlockDesc& d = pl->h_blocks[(size_t)c.blk_base + b];
          const int64_t gs = (int64_t)b * kFramesPerBlock * hop - half;       // clip sample of staged index 0
          auto rel = [&](int64_t x) { const int64_t q = x - gs; return (int32_t)(q < -lim ? -lim : (q > lim ? lim : q)); };
          d.sample_base = c.off + gs; d.frame_slot = c.frame_base + (int64_t)b * kFramesPerBlock; d.clip_off = c.off;
          d.keep_lo = rel(0); d.keep_hi = rel(c.len); d.have_lo = rel(0); d.have_hi = rel(c.len);
          d.clip = i; d.t0 = b * kFramesPerBlock; d.T = c.tmax; d.active = (c.len >= 2 && d.t0 < c.tmax) ? 1 : 0;
          d.pad_[0] = (int32_t)(c.tblk_base * per); d.pad_[1] = (int32_t)(ntb * per);
        }
      }
      HIP_TRY(hipMemcpyAsync(pl->blocks_spec.p, pl->h_blocks.data(), pl->h_blocks.size() * sizeof(BlockDesc), hipMemcpyHostToDevice, s));
      HIP_TRY(hipStreamSynchronize(s));   // h_blocks is pageable and rebuilt by the next call
    }
  }
  pl->c_off.assign(offsets, offsets + n);
  pl->c_len.assign(lengths, lengths + n);
  return AFX_OK;
}

#define TIMED(slot, call)                                                      \
  do {                                                                         \
    const bool timed_ = pl->timing && (!pl->timing_frames_only || (slot) == AFX_K_FRAMES);   \
    if (timed_) HIP_TRY(hipEventRecord(pl->ev[slot][0], s));                   \
    HIP_TRY(call);                                                             \
    if (timed_) { HIP_TRY(hipEventRecord(pl->ev[slot][1], s)); pl->launches[slot]++; } \
  } while (0)

// Everything a chunk needs on the stream, up to the copies of its results into pinned staging; chunk_finish waits and
// hands the results out.  by_event: wait for the chunk's own event (other plans may have queued work behind it on the
// same stream) instead of the whole stream.
static int chunk_enqueue(afx_plan* pl, const void* samples, int fmt, int mem_kind,
                         const int64_t* offsets, const int64_t* lengths, int n, int flags,
                         float* out_stats, int32_t* out_status, int64_t* out_trim, int32_t* out_nframes,
                         float* out_frames, const int64_t* frame_offsets, bool by_event) {
  hipStream_t s = pl->ctx->stream;
  if (pl->pend.active) { set_error("the plan has a submitted batch that has not been collected (afx_extract_collect)"); return AFX_ERR_INVALID; }
  const int K = pl->p.n_mfcc, M = pl->p.n_mels;
  const int nstat = 4 * K + 3;
  int rc;
  if ((rc = prepare_descriptors(pl, offsets, lengths, n)) != AFX_OK) return rc;

  const void* d_samples = samples;
  const size_t esz = fmt == AFX_FMT_S16 ? 2 : 4;
  if (mem_kind == AFX_MEM_HOST) {
    int64_t hi = 0;
    for (int i = 0; i < n; ++i) hi = std::max(hi, offsets[i] + lengths[i]);
    if ((rc = ensure(pl->samples, (size_t)hi * esz + 16)) != AFX_OK) return rc;
    if (hi > 0) HIP_TRY(hipMemcpyAsync(pl->samples.p, samples, (size_t)hi * esz, hipMemcpyHostToDevice, s));
    d_samples = pl->samples.p;
  }
  {
    const size_t before = pl->info.cap;
    if ((rc = ensure(pl->info, n * sizeof(ClipInfo))) != AFX_OK) return rc;
    if (pl->info.cap != before) pl->info_clean_n = 0;          // a new block: contents unknown
  }
  if ((rc = ensure(pl->bsum, std::max<int64_t>(pl->total_tblk, 1) * 4 * sizeof(float))) != AFX_OK) return rc;
  // + 16 frames: with a left cut (frame offset start / hop > 0) the DCT's last 16-frame tile of the batch's last clip reads
  // up to 15 rows past the clip's padded frame count (values discarded); they must be mapped memory
  if ((rc = ensure(pl->logmel, (size_t)(pl->total_tpad + kFramesPerBlock) * M * sizeof(float))) != AFX_OK) return rc;
  if ((rc = ensure(pl->rms, (size_t)pl->total_tpad * sizeof(float))) != AFX_OK) return rc;
  if ((rc = ensure(pl->mfcc, (size_t)pl->total_tpad * K * sizeof(float))) != AFX_OK) return rc;
  // Per-frame output: this chunk's clips occupy [f_lo, f_hi) of the caller's buffer.  The device copy holds exactly that
  // range (offsets rebased), so that a later chunk never touches -- or copies stale device memory over -- an earlier one's rows.
  float* d_frames = nullptr;
  int64_t f_lo = 0, f_hi = 0;
  std::vector<int64_t>& rebased = pl->h_rebased;
  if (out_frames) {
    if (!frame_offsets) { set_error("out_frames given without frame_offsets"); return AFX_ERR_INVALID; }
    f_lo = INT64_MAX;
    for (int i = 0; i < n; ++i) {
      if (frame_offsets[i] < 0) { set_error("negative frame offset"); return AFX_ERR_INVALID; }
      f_lo = std::min(f_lo, frame_offsets[i]);
      f_hi = std::max(f_hi, frame_offsets[i] + (int64_t)(3 * K + 1) * pl->h_clips[i].tmax);
    }
    rebased.resize(n);
    for (int i = 0; i < n; ++i) rebased[i] = frame_offsets[i] - f_lo;
    if ((rc = ensure(pl->frames, (size_t)(f_hi - f_lo) * sizeof(float))) != AFX_OK) return rc;
    if ((rc = ensure(pl->frame_offs, n * sizeof(int64_t))) != AFX_OK) return rc;
    HIP_TRY(hipMemcpyAsync(pl->frame_offs.p, rebased.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(pl->frames.p, 0, (size_t)(f_hi - f_lo) * sizeof(float), s));
    d_frames = (float*)pl->frames.p;
  }

  KParams kp = pl->kp;
  kp.flags = flags; kp.fmt = fmt;
  kp.flags |= dev_env().debug_skip << 8;   // timing ablation only (0 unless AFX_DEBUG_SKIP was set when the library loaded)
  const ClipDesc* d_clips = (const ClipDesc*)pl->clips.p;
  ClipInfo* d_info = (ClipInfo*)pl->info.p;

  // the clip records and the list / ticket counters start a batch cleared: k_finish of the previous batch left them so
  {
    const size_t before = pl->n_items.cap;
    if ((rc = ensure(pl->n_items, 16)) != AFX_OK) return rc;
    if (pl->n_items.cap != before) pl->info_clean_n = 0;
  }
  if (pl->info_clean_n < n) {
    HIP_TRY(hipMemsetAsync(d_info, 0, n * sizeof(ClipInfo), s));
    HIP_TRY(hipMemsetAsync(pl->n_items.p, 0, 16, s));
  }
  pl->info_clean_n = 0;
  const bool want_stamps = dev_env().stamps;     // diagnostic build of k_frames
  const bool f3 = pl->use_f3 && !want_stamps && (!(kp.flags & 0x7f00) || dev_env().f3_debug);
  bool no_spec = dev_env().no_spec;             // A/B: the two-pass pipeline with k_frames3
  // k_trim_blocks sums 256-sample runs: it has the hop-sized sub-block sums the wave-level kernels' RMS rows need only
  // for hops of 256 (or one sum per trim block); other shapes keep the speculative pipeline whatever the switch says
  if (f3 && no_spec && kp.rms_sub > 1 && kp.trim_hop / kp.rms_sub != 256) no_spec = false;
  if (!f3) kp.rms_sub = frames2_eligible(kp, pl->dt) ? kp.trim_hop / kp.hop : 0;      // round 1's kernels: their own rule
  // No per-frame output wanted: one kernel per clip does clamp + DCT + statistics and the MFCC rows stay on the chip
  // (k_tail).  A batch of very few, very long clips keeps the many-workgroups-per-clip kernels.
  const bool fused_tail = f3 && pl->nblocks > 0 && !out_frames && !dev_env().no_fused_tail && tail_eligible(kp, pl->dt) &&
                          (n >= 32 || pl->max_tmax <= 2048);
  if (f3 && !no_spec && pl->nblocks > 0) {
    // the samples are read once: frames before the trim decision (which the same pass feeds), then the few frames a cut touches
    const int max_items = n * kF3ItemsPerClip;
    TIMED(AFX_K_FRAMES, launch_frames3_any(s, d_samples, d_info, (const BlockDesc*)pl->blocks_spec.p, pl->nblocks, nullptr, pl->f3, kp,
                                       (float*)pl->logmel.p, (float*)pl->blockmax.p, (float*)pl->bsum.p, true,
                                       dev_env().no_tickets ? nullptr : (int*)pl->n_items.p + 1, pl->n_cu));
    TIMED(AFX_K_TRIM_DECIDE, launch_trim_decide3(s, d_clips, d_info, (const float*)pl->bsum.p, (const float*)pl->blockmax.p,
                                                 (BlockDesc*)pl->items.p, (int*)pl->n_items.p, max_items, (float*)pl->rms.p, n, kp));
    TIMED(AFX_K_TRIM_BLOCKS, launch_frames3_any(s, d_samples, d_info, (const BlockDesc*)pl->items.p, max_items, (const int*)pl->n_items.p,
                                            pl->f3, kp, (float*)pl->logmel.p, nullptr, nullptr, false, nullptr, pl->n_cu));
    if (!fused_tail)
      TIMED(AFX_K_DCT, launch_dct(s, d_clips, d_info, pl->dt, kp, (const float*)pl->logmel.p, (float*)pl->mfcc.p, n, pl->max_tmax, true, true));
  } else {
  TIMED(AFX_K_TRIM_BLOCKS, launch_trim_blocks(s, d_samples, d_clips, d_info, (float*)pl->bsum.p, n, pl->max_tblocks, kp));
  TIMED(AFX_K_TRIM_DECIDE, launch_trim_decide(s, d_clips, d_info, (const float*)pl->bsum.p, (BlockDesc*)pl->blocks.p, (float*)pl->rms.p, n, kp, d_samples));
  if (pl->nblocks > 0) {
    const int grid = std::min(pl->nblocks, pl->n_cu * 2);
    unsigned long long* d_stamps = nullptr;
    if (want_stamps) {
      if ((rc = ensure(pl->stamps, (size_t)grid * kWaves * kStampPhases * 8)) != AFX_OK) return rc;
      d_stamps = (unsigned long long*)pl->stamps.p;
      HIP_TRY(hipMemsetAsync(d_stamps, 0, (size_t)grid * kWaves * kStampPhases * 8, s));
    }
    if (f3)
      TIMED(AFX_K_FRAMES, launch_frames3_any(s, d_samples, d_info, (const BlockDesc*)pl->blocks.p, pl->nblocks, nullptr, pl->f3, kp,
                                         (float*)pl->logmel.p, nullptr, nullptr, false, nullptr, pl->n_cu));
    else
      TIMED(AFX_K_FRAMES, launch_frames(s, d_samples, d_info, (const BlockDesc*)pl->blocks.p, pl->nblocks,
                                        pl->dt, kp, (float*)pl->logmel.p, (float*)pl->rms.p, grid, d_stamps));
    if (want_stamps) {
      std::vector<unsigned long long> h((size_t)grid * kWaves * kStampPhases);
      HIP_TRY(hipMemcpyAsync(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      static const char* names[kStampPhases] = {"stage", "bar1", "fft", "prefetch", "bar2", "mel", "bar3", "melfin", "x0", "x1", "x2", "x3"};
      double tot[kStampPhases] = {}, wv[kWaves][kStampPhases] = {};
      for (int g = 0; g < grid; ++g)
        for (int w = 0; w < kWaves; ++w)
          for (int ph = 0; ph < kStampPhases; ++ph) {
            const double v = (double)h[((size_t)g * kWaves + w) * kStampPhases + ph];
            tot[ph] += v; wv[w][ph] += v;
          }
      const double per = (double)pl->nblocks * kWaves;      // cycles per block per wave
      fprintf(stderr, "[afx stamps] cycles per block (avg over waves):");
      double all = 0;
      for (int ph = 0; ph < kStampPhases; ++ph) { fprintf(stderr, " %s=%.0f", names[ph], tot[ph] / per); all += tot[ph] / per; }
      fprintf(stderr, " total=%.0f\n", all);
      for (int w = 0; w < kWaves; ++w) {
        fprintf(stderr, "[afx stamps]   wave %d:", w);
        for (int ph = 0; ph < kStampPhases; ++ph) fprintf(stderr, " %s=%.0f", names[ph], wv[w][ph] / pl->nblocks);
        fprintf(stderr, "\n");
      }
    }
    if (!fused_tail)
      TIMED(AFX_K_DCT, launch_dct(s, d_clips, d_info, pl->dt, kp, (const float*)pl->logmel.p, (float*)pl->mfcc.p, n, pl->max_tmax, f3));
  }
  }
  const int tail_spec = (f3 && !no_spec) ? 1 : 0;          // the spill holds absolute frames (speculative pipeline)
  // The batch's small results (statistics, clip records) are written by k_stats straight into pinned host memory the
  // device can address: no copy commands behind the last kernel (each cost ~15 us of stream time).
  const size_t stats_bytes = (size_t)n * nstat * sizeof(float), info_bytes = (size_t)n * sizeof(ClipInfo);
  if (pl->h_pin_cap < stats_bytes + info_bytes) {
    HIP_TRY(hipStreamSynchronize(s));                                       // nothing in flight may still write the old block
    if (pl->h_pin) (void)hipHostFree(pl->h_pin);
    pl->h_pin = nullptr; pl->h_pin_cap = 0; pl->h_pin_dev = nullptr;
    const size_t want = (stats_bytes + info_bytes) * 5 / 4 + 256;          // slack covers the 16-byte round-up
    HIP_TRY(hipHostMalloc(&pl->h_pin, want, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(&pl->h_pin_dev, pl->h_pin, 0));
    pl->h_pin_cap = want;
  }
  const size_t info_at = (stats_bytes + 15) & ~(size_t)15;
  if (fused_tail)
    TIMED(AFX_K_DCT, launch_tail(s, d_clips, d_info, pl->dt, kp, (const float*)pl->logmel.p, (const float*)pl->rms.p,
                                 (float*)pl->h_pin_dev, (ClipInfo*)((char*)pl->h_pin_dev + info_at), n, tail_spec, pl->n_cu));
  else
  TIMED(AFX_K_STATS, launch_stats(s, d_clips, d_info, kp, (const float*)pl->mfcc.p, (const float*)pl->rms.p,
                                  (float*)pl->h_pin_dev, d_frames, (const int64_t*)pl->frame_offs.p, n,
                                  (ClipInfo*)((char*)pl->h_pin_dev + info_at)));
  if (out_frames && f_hi > f_lo)
    HIP_TRY(hipMemcpyAsync(out_frames + f_lo, d_frames, (size_t)(f_hi - f_lo) * sizeof(float), hipMemcpyDeviceToHost, s));
  if (by_event) {
    if (!pl->done) HIP_TRY(hipEventCreateWithFlags(&pl->done, hipEventDisableTiming));
    if (!pl->flag) {
      void* h = nullptr;
      HIP_TRY(hipHostMalloc(&h, 64, hipHostMallocMapped));
      std::memset(h, 0, 64);
      pl->flag = (volatile unsigned*)h;
      HIP_TRY(hipHostGetDevicePointer((void**)&pl->flag_dev, h, 0));
    }
  }
  HIP_TRY(launch_finish(s, d_info, n, (int*)pl->n_items.p, by_event ? pl->flag_dev : nullptr, by_event ? ++pl->seq : 0u));
  pl->info_clean_n = n;
  if (by_event) HIP_TRY(hipEventRecord(pl->done, s));
  pl->pend.active = true; pl->pend.n = n; pl->pend.stats_bytes = stats_bytes;
  pl->pend.out_stats = out_stats; pl->pend.out_status = out_status; pl->pend.out_trim = out_trim; pl->pend.out_nframes = out_nframes;
  return AFX_OK;
}

static int chunk_finish(afx_plan* pl, bool by_event) {
  if (!pl->pend.active) { set_error("afx_extract_collect: nothing submitted"); return AFX_ERR_INVALID; }
  hipStream_t s = pl->ctx->stream;
  pl->pend.active = false;          // whatever happens below, the chunk is over
  if (by_event) {
    // spin on the flag; every so often ask the runtime, so that a failed launch ends the wait with its error
    for (unsigned it = 1; *pl->flag != pl->seq; ++it) {
      if ((it & 0xfff) == 0) {
        const hipError_t q = hipEventQuery(pl->done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { set_error(std::string("afx_extract_collect: ") + hipGetErrorString(q)); return AFX_ERR_HIP; }
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else {
    HIP_TRY(hipStreamSynchronize(s));
  }
  const int n = pl->pend.n;
  const size_t stats_bytes = pl->pend.stats_bytes;
  float* out_stats = pl->pend.out_stats; int32_t* out_status = pl->pend.out_status;
  int64_t* out_trim = pl->pend.out_trim; int32_t* out_nframes = pl->pend.out_nframes;
  const float* h_stats = (const float*)pl->h_pin;
  const ClipInfo* h_info = (const ClipInfo*)((char*)pl->h_pin + ((stats_bytes + 15) & ~(size_t)15));
  if (pl->timing) {
    for (int k = 0; k < AFX_K_COUNT; ++k) {
      if (pl->nblocks == 0 && (k == AFX_K_FRAMES || k == AFX_K_DCT)) continue;
      if (pl->launches[k] == 0) continue;
      float ms = 0.f, t0 = 0.f;
      if (hipEventElapsedTime(&ms, pl->ev[k][0], pl->ev[k][1]) == hipSuccess) {
        pl->ms_sum[k] += ms;
        hipEvent_t ref;
        if (device_ref(pl, &ref) == AFX_OK && hipEventElapsedTime(&t0, ref, pl->ev[k][0]) == hipSuccess) {
          if (pl->spans[k].size() >= kMaxSpans) pl->spans[k].erase(pl->spans[k].begin(), pl->spans[k].begin() + kMaxSpans / 2);
          pl->spans[k].emplace_back((double)t0, (double)t0 + (double)ms);
        }
      }
    }
  }
  std::memcpy(out_stats, h_stats, stats_bytes);
  for (int i = 0; i < n; ++i) {
    const ClipInfo& ci = h_info[i];
    out_status[i] = ci.status;
    if (out_trim) { out_trim[2 * i] = ci.start; out_trim[2 * i + 1] = ci.end; }
    if (out_nframes) out_nframes[i] = ci.T;
  }
  return AFX_OK;
}

static int extract_chunk(afx_plan* pl, const void* samples, int fmt, int mem_kind,
                         const int64_t* offsets, const int64_t* lengths, int n, int flags,
                         float* out_stats, int32_t* out_status, int64_t* out_trim, int32_t* out_nframes,
                         float* out_frames, const int64_t* frame_offsets) {
  int rc = chunk_enqueue(pl, samples, fmt, mem_kind, offsets, lengths, n, flags, out_stats, out_status, out_trim, out_nframes,
                         out_frames, frame_offsets, false);
  if (rc != AFX_OK) return rc;
  return chunk_finish(pl, false);
}

static int check_extract_args(const char* who, afx_plan* pl, const void* samples, int sample_fmt, int mem_kind,
                              const int64_t* offsets, const int64_t* lengths, int n_clips, float* out_stats, int32_t* out_status) {
  if (!pl || !offsets || !lengths || !out_stats || !out_status || n_clips < 0 || (!samples && n_clips > 0)) {
    set_error(std::string(who) + ": null/invalid argument");
    return AFX_ERR_INVALID;
  }
  if (sample_fmt != AFX_FMT_F32 && sample_fmt != AFX_FMT_S16) { set_error("unknown sample format"); return AFX_ERR_INVALID; }
  if (mem_kind != AFX_MEM_HOST && mem_kind != AFX_MEM_DEVICE) { set_error("unknown mem_kind"); return AFX_ERR_INVALID; }
  return AFX_OK;
}

extern "C" int afx_extract_submit(afx_plan* pl, const void* samples, int sample_fmt, int mem_kind,
                                  const int64_t* offsets, const int64_t* lengths, int n_clips, int flags,
                                  float* out_stats, int32_t* out_status, int64_t* out_trim,
                                  int32_t* out_nframes, float* out_frames, const int64_t* frame_offsets) {
  int rc = check_extract_args("afx_extract_submit", pl, samples, sample_fmt, mem_kind, offsets, lengths, n_clips, out_stats, out_status);
  if (rc != AFX_OK) return rc;
  if (n_clips == 0) { set_error("afx_extract_submit: empty batch"); return AFX_ERR_INVALID; }
  if (n_clips > dev_env().chunk_clips) { set_error("afx_extract_submit: more clips than one chunk holds; use afx_extract_batch"); return AFX_ERR_UNSUPPORTED; }
  if (pl->pend.active) { set_error("afx_extract_submit: the plan's previous batch has not been collected"); return AFX_ERR_INVALID; }
  (void)hipGetLastError();
  HIP_TRY(hipSetDevice(pl->device));
  rc = chunk_enqueue(pl, samples, sample_fmt, mem_kind, offsets, lengths, n_clips, flags, out_stats, out_status, out_trim,
                     out_nframes, out_frames, frame_offsets, true);
  return rc;
}

extern "C" int afx_extract_collect(afx_plan* pl) {
  if (!pl) { set_error("afx_extract_collect: null plan"); return AFX_ERR_INVALID; }
  HIP_TRY(hipSetDevice(pl->device));
  return chunk_finish(pl, true);
}

extern "C" int afx_extract_batch(afx_plan* pl, const void* samples, int sample_fmt, int mem_kind,
                                 const int64_t* offsets, const int64_t* lengths, int n_clips, int flags,
                                 float* out_stats, int32_t* out_status, int64_t* out_trim,
                                 int32_t* out_nframes, float* out_frames, const int64_t* frame_offsets) {
  if (!pl || !offsets || !lengths || !out_stats || !out_status || n_clips < 0 || (!samples && n_clips > 0)) {
    set_error("afx_extract_batch: null/invalid argument");
    return AFX_ERR_INVALID;
  }
  if (sample_fmt != AFX_FMT_F32 && sample_fmt != AFX_FMT_S16) { set_error("unknown sample format"); return AFX_ERR_INVALID; }
  if (mem_kind != AFX_MEM_HOST && mem_kind != AFX_MEM_DEVICE) { set_error("unknown mem_kind"); return AFX_ERR_INVALID; }
  if (n_clips == 0) return AFX_OK;
  (void)hipGetLastError();      // a stale error of an unrelated earlier call must not be blamed on this one
  HIP_TRY(hipSetDevice(pl->device));
  const int nstat = 4 * pl->p.n_mfcc + 3;
  const int kChunk = dev_env().chunk_clips;   // 32768: gridDim.y limit is 65535
  for (int c0 = 0; c0 < n_clips; c0 += kChunk) {
    const int n = std::min(kChunk, n_clips - c0);
    int rc = extract_chunk(pl, samples, sample_fmt, mem_kind, offsets + c0, lengths + c0, n, flags,
                           out_stats + (size_t)c0 * nstat, out_status + c0,
                           out_trim ? out_trim + 2 * (size_t)c0 : nullptr,
                           out_nframes ? out_nframes + c0 : nullptr, out_frames,
                           frame_offsets ? frame_offsets + c0 : nullptr);
    if (rc != AFX_OK) return rc;
  }
  return AFX_OK;
}

// ---- extract_f0 ---------------------------------------------------------------------------------
static int f0_setup(afx_plan* pl, double fmin, double fmax) {
  if (pl->f0_ready && pl->f0_fmin == fmin && pl->f0_fmax == fmax) return AFX_OK;
  std::string why;
  HostF0Tables ht;
  if (!build_f0_tables(pl->p.sr, pl->p.n_fft, pl->p.hop, fmin, fmax, ht, why)) {
    set_error("afx_f0_batch: " + why);
    return AFX_ERR_UNSUPPORTED;
  }
  if (f0_energy_lds_bytes(ht.p) > 160 * 1024 || f0_yin_lds_bytes(ht.p) > 160 * 1024 ||
      f0_viterbi_lds_bytes(ht.p) > 160 * 1024 || f0_backtrack_lds_bytes(ht.p) > 160 * 1024 || 2 * ht.p.band + 1 > 64) {
    set_error("afx_f0_batch: frame_length / f0 range needs more than 160 KiB of LDS");
    return AFX_ERR_UNSUPPORTED;
  }
  for (void* q : pl->f0_allocs) (void)hipFree(q);
  pl->f0_allocs.clear();
  pl->f0_ready = false;
  auto up = [&](const std::vector<double>& v, const double** dst) -> int {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, v.size() * sizeof(double)));
    pl->f0_allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    *dst = (const double*)d;
    return AFX_OK;
  };
  int rc;
  if ((rc = up(ht.thr, &pl->f0_dt.thr)) != AFX_OK || (rc = up(ht.beta, &pl->f0_dt.beta)) != AFX_OK ||
      (rc = up(ht.cumbeta, &pl->f0_dt.cumbeta)) != AFX_OK || (rc = up(ht.bfact, &pl->f0_dt.bfact)) != AFX_OK ||
      (rc = up(ht.bexp, &pl->f0_dt.bexp)) != AFX_OK || (rc = up(ht.lt, &pl->f0_dt.lt)) != AFX_OK || (rc = up(ht.ltw, &pl->f0_dt.ltw)) != AFX_OK ||
      (rc = up(ht.freqs, &pl->f0_dt.freqs)) != AFX_OK)
    return rc;
  if (dev_env().f0_debug) ht.p.debug = dev_env().f0_debug;
  pl->f0_ht = ht;
  pl->f0_fmin = fmin; pl->f0_fmax = fmax;
  pl->f0_ready = true;
  return AFX_OK;
}

static int f0_chunk(afx_plan* pl, const void* d_samples, int fmt, const int64_t* offsets, const int64_t* lengths,
                    int n, int flags, double* out_stats, int32_t* out_status, double* out_f0,
                    const int64_t* f0_offsets) {
  hipStream_t s = pl->ctx->stream;
  const F0Params& fp = pl->f0_ht.p;
  int rc;
  if ((rc = prepare_descriptors(pl, offsets, lengths, n)) != AFX_OK) return rc;
  int64_t hi = 0, max_len = 0;
  for (int i = 0; i < n; ++i) { hi = std::max(hi, offsets[i] + lengths[i]); max_len = std::max(max_len, lengths[i]); }
  const int64_t frames = std::max<int64_t>(pl->total_tpad, 1);
  if ((rc = ensure(pl->info, n * sizeof(ClipInfo))) != AFX_OK) return rc;
  if ((rc = ensure(pl->bsum, std::max<int64_t>(pl->total_tblk, 1) * 4 * sizeof(float))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_ysig, (size_t)std::max<int64_t>(hi, 1) * sizeof(float))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_energy, (size_t)frames * fp.n_tau_pad * sizeof(float))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_cnt, (size_t)frames * sizeof(int32_t))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_vp, (size_t)frames * sizeof(double))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_bin, f0_cand_bins_bytes(fp, frames))) != AFX_OK) return rc;
  const bool dump_obs = dev_env().f0_dump != nullptr;        // the linear probabilities are kept for the diagnostic dump only
  if (dump_obs && (rc = ensure(pl->f0_prob, f0_cand_prob_bytes(fp, frames))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_ptr, f0_vrows_bytes(fp, frames))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_best, (size_t)frames * sizeof(VitBest))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_lprob, f0_cand_prob_bytes(fp, frames))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_lu, (size_t)frames * sizeof(double))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_states, (size_t)frames * sizeof(uint16_t))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_stats, (size_t)n * 4 * sizeof(double))) != AFX_OK) return rc;
  // per-frame f0 of this chunk: [o_lo, o_hi) of the caller's buffer, device copy rebased to it (see extract_chunk)
  double* d_f0 = nullptr;
  int64_t o_lo = 0, o_hi = 0;
  std::vector<int64_t> rebased;
  if (out_f0) {
    if (!f0_offsets) { set_error("out_f0 given without f0_offsets"); return AFX_ERR_INVALID; }
    o_lo = INT64_MAX;
    for (int i = 0; i < n; ++i) {
      if (f0_offsets[i] < 0) { set_error("negative f0 offset"); return AFX_ERR_INVALID; }
      o_lo = std::min(o_lo, f0_offsets[i]);
      o_hi = std::max(o_hi, f0_offsets[i] + (int64_t)pl->h_clips[i].tmax);
    }
    rebased.resize(n);
    for (int i = 0; i < n; ++i) rebased[i] = f0_offsets[i] - o_lo;
    if ((rc = ensure(pl->f0_out, (size_t)(o_hi - o_lo) * sizeof(double))) != AFX_OK) return rc;
    if ((rc = ensure(pl->f0_offs, n * sizeof(int64_t))) != AFX_OK) return rc;
    HIP_TRY(hipMemcpyAsync(pl->f0_offs.p, rebased.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(pl->f0_out.p, 0xff, (size_t)(o_hi - o_lo) * sizeof(double), s));     // NaN wherever no clip writes
    d_f0 = (double*)pl->f0_out.p;
  }
  KParams kp = pl->kp;
  kp.flags = flags; kp.fmt = fmt;
  const ClipDesc* d_clips = (const ClipDesc*)pl->clips.p;
  ClipInfo* d_info = (ClipInfo*)pl->info.p;
  HIP_TRY(hipMemsetAsync(d_info, 0, n * sizeof(ClipInfo), s));
  pl->info_clean_n = 0;      // this path leaves the clip records used
  HIP_TRY(launch_trim_blocks(s, d_samples, d_clips, d_info, (float*)pl->bsum.p, n, pl->max_tblocks, kp));
  HIP_TRY(launch_trim_decide(s, d_clips, d_info, (const float*)pl->bsum.p, (BlockDesc*)pl->blocks.p, nullptr, n, kp));
  HIP_TRY(launch_f0_prep(s, d_samples, d_clips, d_info, (float*)pl->f0_ysig.p, n, max_len, kp));
  HIP_TRY(launch_f0_energy(s, (const float*)pl->f0_ysig.p, d_clips, d_info, (float*)pl->f0_energy.p, n, pl->max_tmax, fp));
  HIP_TRY(launch_f0_yin(s, (const float*)pl->f0_ysig.p, d_clips, d_info, (const float*)pl->f0_energy.p, pl->f0_dt, fp,
                        (int32_t*)pl->f0_cnt.p, (double*)pl->f0_vp.p, (int16_t*)pl->f0_bin.p,
                        dump_obs ? (double*)pl->f0_prob.p : nullptr, (double*)pl->f0_lprob.p, (double*)pl->f0_lu.p, n, pl->max_tmax));
  HIP_TRY(launch_f0_viterbi(s, d_clips, d_info, pl->f0_dt, fp, (const int32_t*)pl->f0_cnt.p,
                            (const int16_t*)pl->f0_bin.p, (const double*)pl->f0_lprob.p,
                            (const double*)pl->f0_lu.p, (double*)pl->f0_ptr.p, (VitBest*)pl->f0_best.p,
                            (uint16_t*)pl->f0_states.p, (double*)pl->f0_stats.p, d_f0, (const int64_t*)pl->f0_offs.p, n));
  std::vector<ClipInfo> h_info(n);
  HIP_TRY(hipMemcpyAsync(out_stats, pl->f0_stats.p, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_info.data(), d_info, n * sizeof(ClipInfo), hipMemcpyDeviceToHost, s));
  if (out_f0 && o_hi > o_lo) HIP_TRY(hipMemcpyAsync(out_f0 + o_lo, d_f0, (size_t)(o_hi - o_lo) * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (const char* dump = dev_env().f0_dump) {                       // diagnostics: the sparse observation columns
    std::vector<int32_t> cnt(frames); std::vector<double> vp(frames), pr((size_t)frames * fp.cap);
    std::vector<int16_t> bn((size_t)frames * fp.cap);
    HIP_TRY(hipMemcpy(cnt.data(), pl->f0_cnt.p, frames * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(vp.data(), pl->f0_vp.p, frames * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pr.data(), pl->f0_prob.p, pr.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(bn.data(), pl->f0_bin.p, bn.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(dump, "wb")) {
      const int64_t hdr[2] = {frames, fp.cap};
      fwrite(hdr, sizeof(hdr), 1, f);
      fwrite(cnt.data(), sizeof(int32_t), cnt.size(), f); fwrite(vp.data(), sizeof(double), vp.size(), f);
      fwrite(bn.data(), sizeof(int16_t), bn.size(), f); fwrite(pr.data(), sizeof(double), pr.size(), f);
      fclose(f);
    }
  }
  for (int i = 0; i < n; ++i) out_status[i] = h_info[i].nonfinite ? AFX_CLIP_NONFINITE : AFX_CLIP_OK;
  return AFX_OK;
}

extern "C" int afx_f0_batch(afx_plan* pl, const void* samples, int sample_fmt, int mem_kind,
                            const int64_t* offsets, const int64_t* lengths, int n_clips, int flags,
                            double fmin, double fmax, double* out_f0stats, int32_t* out_status,
                            double* out_f0, const int64_t* f0_offsets) {
  if (!pl || !offsets || !lengths || !out_f0stats || !out_status || n_clips < 0 || (!samples && n_clips > 0)) {
    set_error("afx_f0_batch: null/invalid argument");
    return AFX_ERR_INVALID;
  }
  if (sample_fmt != AFX_FMT_F32 && sample_fmt != AFX_FMT_S16) { set_error("unknown sample format"); return AFX_ERR_INVALID; }
  if (mem_kind != AFX_MEM_HOST && mem_kind != AFX_MEM_DEVICE) { set_error("unknown mem_kind"); return AFX_ERR_INVALID; }
  if (n_clips == 0) return AFX_OK;
  (void)hipGetLastError();
  HIP_TRY(hipSetDevice(pl->device));
  if (pl->pend.active) { set_error("afx_f0_batch: the plan has a submitted batch that has not been collected"); return AFX_ERR_INVALID; }
  int rc;
  if ((rc = f0_setup(pl, fmin, fmax)) != AFX_OK) return rc;
  const void* d_samples = samples;
  if (mem_kind == AFX_MEM_HOST) {
    const size_t esz = sample_fmt == AFX_FMT_S16 ? 2 : 4;
    int64_t hi = 0;
    for (int i = 0; i < n_clips; ++i) hi = std::max(hi, offsets[i] + lengths[i]);
    if ((rc = ensure(pl->f0_in, (size_t)hi * esz + 16)) != AFX_OK) return rc;
    if (hi > 0) HIP_TRY(hipMemcpyAsync(pl->f0_in.p, samples, (size_t)hi * esz, hipMemcpyHostToDevice, pl->ctx->stream));
    d_samples = pl->f0_in.p;
  }
  // the stage keeps ~14 KB of workspace per frame (Viterbi value columns 9.6 KB, candidates and their logs, energies): bound it
  // per chunk (18 GB; a chunk should still hold several clips per CU so that every CU runs two Viterbi workgroups)
  const int64_t kMaxFrames = dev_env().f0_chunk_frames;
  int c0 = 0;
  while (c0 < n_clips) {
    int n = 0;
    int64_t fr = 0;
    while (c0 + n < n_clips && n < dev_env().chunk_clips) {
      const int64_t t = 1 + lengths[c0 + n] / pl->p.hop + kFramesPerBlock;
      if (n > 0 && fr + t > kMaxFrames) break;
      fr += t; ++n;
    }
    rc = f0_chunk(pl, d_samples, sample_fmt, offsets + c0, lengths + c0, n, flags, out_f0stats + (size_t)c0 * 4,
                  out_status + c0, out_f0, f0_offsets ? f0_offsets + c0 : nullptr);
    if (rc != AFX_OK) return rc;
    c0 += n;
  }
  return AFX_OK;
}

// ---- zero-crossing rate per frame (the sibling feature the reference's experiment scripts store) ---------
extern "C" int afx_zcr_batch(afx_plan* pl, const void* samples, int sample_fmt, int mem_kind,
                             const int64_t* offsets, const int64_t* lengths, int n_clips, int flags,
                             double* out_zcr, const int64_t* zcr_offsets, int32_t* out_status) {
  if (!pl || !offsets || !lengths || !out_zcr || !zcr_offsets || !out_status || n_clips < 0 || (!samples && n_clips > 0)) {
    set_error("afx_zcr_batch: null/invalid argument");
    return AFX_ERR_INVALID;
  }
  if (sample_fmt != AFX_FMT_F32 && sample_fmt != AFX_FMT_S16) { set_error("unknown sample format"); return AFX_ERR_INVALID; }
  if (mem_kind != AFX_MEM_HOST && mem_kind != AFX_MEM_DEVICE) { set_error("unknown mem_kind"); return AFX_ERR_INVALID; }
  if (n_clips == 0) return AFX_OK;
  if (n_clips > 32768) { set_error("afx_zcr_batch: at most 32768 clips per call"); return AFX_ERR_INVALID; }
  (void)hipGetLastError();
  HIP_TRY(hipSetDevice(pl->device));
  if (pl->pend.active) { set_error("afx_zcr_batch: the plan has a submitted batch that has not been collected"); return AFX_ERR_INVALID; }
  hipStream_t s = pl->ctx->stream;
  const int n = n_clips;
  int rc;
  const void* d_samples = samples;
  int64_t hi = 0, max_len = 0;
  for (int i = 0; i < n; ++i) { hi = std::max(hi, offsets[i] + lengths[i]); max_len = std::max(max_len, lengths[i]); }
  if (mem_kind == AFX_MEM_HOST) {
    const size_t esz = sample_fmt == AFX_FMT_S16 ? 2 : 4;
    if ((rc = ensure(pl->f0_in, (size_t)hi * esz + 16)) != AFX_OK) return rc;
    if (hi > 0) HIP_TRY(hipMemcpyAsync(pl->f0_in.p, samples, (size_t)hi * esz, hipMemcpyHostToDevice, s));
    d_samples = pl->f0_in.p;
  }
  if ((rc = prepare_descriptors(pl, offsets, lengths, n)) != AFX_OK) return rc;
  size_t count = 0;
  for (int i = 0; i < n; ++i) count = std::max<size_t>(count, (size_t)zcr_offsets[i] + (size_t)pl->h_clips[i].tmax);
  if ((rc = ensure(pl->info, n * sizeof(ClipInfo))) != AFX_OK) return rc;
  if ((rc = ensure(pl->bsum, std::max<int64_t>(pl->total_tblk, 1) * 4 * sizeof(float))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_ysig, (size_t)std::max<int64_t>(hi, 1) * sizeof(float))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_out, std::max<size_t>(count, 1) * sizeof(double))) != AFX_OK) return rc;
  if ((rc = ensure(pl->f0_offs, n * sizeof(int64_t))) != AFX_OK) return rc;
  HIP_TRY(hipMemcpyAsync(pl->f0_offs.p, zcr_offsets, n * sizeof(int64_t), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(pl->f0_out.p, 0, std::max<size_t>(count, 1) * sizeof(double), s));
  KParams kp = pl->kp;
  kp.flags = flags; kp.fmt = sample_fmt;
  const ClipDesc* d_clips = (const ClipDesc*)pl->clips.p;
  ClipInfo* d_info = (ClipInfo*)pl->info.p;
  HIP_TRY(hipMemsetAsync(d_info, 0, n * sizeof(ClipInfo), s));
  pl->info_clean_n = 0;      // this path leaves the clip records used
  HIP_TRY(launch_trim_blocks(s, d_samples, d_clips, d_info, (float*)pl->bsum.p, n, pl->max_tblocks, kp));
  HIP_TRY(launch_trim_decide(s, d_clips, d_info, (const float*)pl->bsum.p, (BlockDesc*)pl->blocks.p, nullptr, n, kp));
  HIP_TRY(launch_f0_prep(s, d_samples, d_clips, d_info, (float*)pl->f0_ysig.p, n, max_len, kp));
  HIP_TRY(launch_zcr(s, (const float*)pl->f0_ysig.p, d_clips, d_info, pl->p.n_fft, pl->p.hop, (double*)pl->f0_out.p,
                     (const int64_t*)pl->f0_offs.p, n, pl->max_tmax));
  std::vector<ClipInfo> h_info(n);
  HIP_TRY(hipMemcpyAsync(out_zcr, pl->f0_out.p, count * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_info.data(), d_info, n * sizeof(ClipInfo), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i) out_status[i] = h_info[i].nonfinite ? AFX_CLIP_NONFINITE : AFX_CLIP_OK;
  return AFX_OK;
}

// ---- spectral descriptors (librosa.feature.spectral_centroid / _bandwidth / _rolloff / _contrast at their defaults) ------
extern "C" int afx_spectral_batch(afx_plan* pl, const void* samples, int sample_fmt, int mem_kind,
                                  const int64_t* offsets, const int64_t* lengths, int n_clips, int flags,
                                  float* out_desc, const int64_t* desc_offsets, int32_t* out_status) {
  if (!pl || !offsets || !lengths || !out_desc || !desc_offsets || !out_status || n_clips < 0 || (!samples && n_clips > 0)) {
    set_error("afx_spectral_batch: null/invalid argument");
    return AFX_ERR_INVALID;
  }
  if (sample_fmt != AFX_FMT_F32 && sample_fmt != AFX_FMT_S16) { set_error("unknown sample format"); return AFX_ERR_INVALID; }
  if (mem_kind != AFX_MEM_HOST && mem_kind != AFX_MEM_DEVICE) { set_error("unknown mem_kind"); return AFX_ERR_INVALID; }
  if (pl->p.n_fft != 2048 || pl->p.hop != 512 || !pl->use_f3) {
    set_error("afx_spectral_batch: the plan must have frame_length 2048 and hop_length 512 (librosa's defaults for these features)");
    return AFX_ERR_UNSUPPORTED;
  }
  if (flags & AFX_FLAG_TRIM) { set_error("afx_spectral_batch: trim is not applied here; pass the preprocessed signal"); return AFX_ERR_UNSUPPORTED; }
  if (n_clips == 0) return AFX_OK;
  if (n_clips > 32768) { set_error("afx_spectral_batch: at most 32768 clips per call"); return AFX_ERR_INVALID; }
  // octave bands of librosa.feature.spectral_contrast(fmin=200, n_bands=6, quantile=0.02) as bin ranges
  SpecBands sb{};
  {
    const int NB = 1025;
    const double sr = (double)pl->p.sr, df = sr / 2048.0;
    double octa[8];
    octa[0] = 0.0;
    for (int i = 1; i < 8; ++i) octa[i] = 200.0 * std::pow(2.0, (double)(i - 1));
    for (int i = 0; i < 7; ++i)
      if (octa[i] >= 0.5 * sr) { set_error("spectral_contrast: frequency band exceeds Nyquist (sr too low for 6 octave bands from 200 Hz)"); return AFX_ERR_UNSUPPORTED; }
    for (int k = 0; k < 7; ++k) {
      int b0 = -1, b1 = -1;
      for (int b = 0; b < NB; ++b) { const double f = (double)b * df; if (f >= octa[k] && f <= octa[k + 1]) { if (b0 < 0) b0 = b; b1 = b; } }
      if (b0 < 0) { set_error("spectral_contrast: empty band"); return AFX_ERR_UNSUPPORTED; }
      if (k > 0) b0 -= 1;
      if (k == 6) b1 = NB - 1;
      const int n_cur = b1 - b0 + 1;
      sb.cnt[k] = std::max(1, (int)std::nearbyint(0.02 * (double)n_cur));
      sb.lo[k] = b0; sb.hi[k] = (k < 6) ? b1 - 1 : b1;
    }
    sb.hz_per_bin = (float)df; sb.roll_percent = 0.85f;
  }
  (void)hipGetLastError();
  HIP_TRY(hipSetDevice(pl->device));
  if (pl->pend.active) { set_error("afx_spectral_batch: the plan has a submitted batch that has not been collected"); return AFX_ERR_INVALID; }
  hipStream_t s = pl->ctx->stream;
  const int n = n_clips;
  int rc;
  const void* d_samples = samples;
  if (mem_kind == AFX_MEM_HOST) {
    const size_t esz = sample_fmt == AFX_FMT_S16 ? 2 : 4;
    int64_t hi = 0;
    for (int i = 0; i < n; ++i) hi = std::max(hi, offsets[i] + lengths[i]);
    if ((rc = ensure(pl->samples, (size_t)hi * esz + 16)) != AFX_OK) return rc;
    if (hi > 0) HIP_TRY(hipMemcpyAsync(pl->samples.p, samples, (size_t)hi * esz, hipMemcpyHostToDevice, s));
    d_samples = pl->samples.p;
  }
  if ((rc = prepare_descriptors(pl, offsets, lengths, n)) != AFX_OK) return rc;
  int64_t d_lo = INT64_MAX, d_hi = 0;
  std::vector<int64_t> rebased(n);
  for (int i = 0; i < n; ++i) {
    if (desc_offsets[i] < 0) { set_error("negative descriptor offset"); return AFX_ERR_INVALID; }
    d_lo = std::min(d_lo, desc_offsets[i]);
    d_hi = std::max(d_hi, desc_offsets[i] + (int64_t)kSpecFloats * pl->h_clips[i].tmax);
  }
  for (int i = 0; i < n; ++i) rebased[i] = desc_offsets[i] - d_lo;
  if ((rc = ensure(pl->info, n * sizeof(ClipInfo))) != AFX_OK) return rc;
  if ((rc = ensure(pl->frames, (size_t)(d_hi - d_lo) * sizeof(float))) != AFX_OK) return rc;
  if ((rc = ensure(pl->frame_offs, n * sizeof(int64_t))) != AFX_OK) return rc;
  HIP_TRY(hipMemcpyAsync(pl->frame_offs.p, rebased.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(pl->frames.p, 0, (size_t)(d_hi - d_lo) * sizeof(float), s));
  HIP_TRY(hipMemsetAsync(pl->info.p, 0, n * sizeof(ClipInfo), s));
  pl->info_clean_n = 0;      // this path leaves the clip records used
  KParams kp = pl->kp;
  kp.flags = flags; kp.fmt = sample_fmt;
  if (pl->nblocks > 0)
    HIP_TRY(launch_spectral(s, d_samples, (ClipInfo*)pl->info.p, (const BlockDesc*)pl->blocks_spec.p, pl->nblocks, pl->f3, kp,
                            (float*)pl->frames.p, (const int64_t*)pl->frame_offs.p, sb, pl->n_cu));
  HIP_TRY(hipMemcpyAsync(out_desc + d_lo, pl->frames.p, (size_t)(d_hi - d_lo) * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i) out_status[i] = lengths[i] < 2 ? AFX_CLIP_TOO_SHORT : AFX_CLIP_OK;
  return AFX_OK;
}

extern "C" int afx_preprocess(afx_plan* pl, const float* y, int64_t n, float* out_y,
                              int64_t* start, int64_t* end, int32_t* status) {
  if (!pl || !y || !out_y || !start || !end || !status || n < 0) {
    set_error("afx_preprocess: null/invalid argument");
    return AFX_ERR_INVALID;
  }
  (void)hipGetLastError();
  HIP_TRY(hipSetDevice(pl->device));
  if (pl->pend.active) { set_error("afx_preprocess: the plan has a submitted batch that has not been collected"); return AFX_ERR_INVALID; }
  hipStream_t s = pl->ctx->stream;
  const int64_t off = 0;
  int rc;
  if ((rc = prepare_descriptors(pl, &off, &n, 1)) != AFX_OK) return rc;
  if ((rc = ensure(pl->samples, (size_t)n * 4 + 16)) != AFX_OK) return rc;
  if ((rc = ensure(pl->info, sizeof(ClipInfo))) != AFX_OK) return rc;
  if ((rc = ensure(pl->bsum, std::max<int64_t>(pl->total_tblk, 1) * 4 * sizeof(float))) != AFX_OK) return rc;
  if ((rc = ensure(pl->logmel, (size_t)std::max<int64_t>(n, 1) * sizeof(float))) != AFX_OK) return rc;   // y_pre scratch
  if (n > 0) HIP_TRY(hipMemcpyAsync(pl->samples.p, y, (size_t)n * 4, hipMemcpyHostToDevice, s));
  KParams kp = pl->kp;
  kp.flags = AFX_FLAG_PREEMPH | AFX_FLAG_TRIM; kp.fmt = AFX_FMT_F32;
  ClipInfo* d_info = (ClipInfo*)pl->info.p;
  HIP_TRY(hipMemsetAsync(d_info, 0, sizeof(ClipInfo), s));
  pl->info_clean_n = 0;      // this path leaves the clip records used
  HIP_TRY(launch_trim_blocks(s, pl->samples.p, (const ClipDesc*)pl->clips.p, d_info, (float*)pl->bsum.p, 1, pl->max_tblocks, kp));
  HIP_TRY(launch_trim_decide(s, (const ClipDesc*)pl->clips.p, d_info, (const float*)pl->bsum.p, (BlockDesc*)pl->blocks.p, nullptr, 1, kp));
  if (n > 0) {
    HIP_TRY(launch_preemph(s, (const float*)pl->samples.p, (float*)pl->logmel.p, n, kp.preemph_b1));
    HIP_TRY(hipMemcpyAsync(out_y, pl->logmel.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  }
  ClipInfo ci{};
  HIP_TRY(hipMemcpyAsync(&ci, d_info, sizeof(ClipInfo), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  // T is about the MFCC stage; preprocess_audio itself only fails on < 2 samples / non-finite input
  *start = ci.start; *end = ci.end;
  *status = (n < 2) ? AFX_CLIP_TOO_SHORT : (ci.nonfinite ? AFX_CLIP_NONFINITE : AFX_CLIP_OK);
  return AFX_OK;
}
