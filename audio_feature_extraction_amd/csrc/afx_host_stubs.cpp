// Device entry points of the HOST-ONLY build (libafx_host_asan.so, `make asan`): every one fails loudly with
// AFX_ERR_NO_DEVICE.  The sanitizer build exists to run the host-side parsers and table builders (afx_wav.cpp,
// afx_tables.cpp, afx_f0_tables.cpp, afx_host.cpp) under AddressSanitizer + UBSan on the CPU; GPU sanitizers are not
// available on the target pool.  Never linked into libafx.so.
#include <cstddef>
#include <cstdint>

#include "afx.h"
#include "afx_internal.h"

static int no_device(const char* who) {
  afx::set_error(std::string(who) + ": host-only sanitizer build of libafx (no device code)");
  return AFX_ERR_NO_DEVICE;
}

extern "C" {
int afx_device_count(void) { return 0; }
int afx_init(int, afx_ctx** out) { if (out) *out = nullptr; return no_device("afx_init"); }
void afx_destroy(afx_ctx*) {}
int afx_malloc(afx_ctx*, size_t, void**) { return no_device("afx_malloc"); }
int afx_free(afx_ctx*, void*) { return no_device("afx_free"); }
int afx_host_alloc(afx_ctx*, size_t, void** out) { if (out) *out = nullptr; return no_device("afx_host_alloc"); }
int afx_host_free(afx_ctx*, void*) { return no_device("afx_host_free"); }
int afx_memcpy_h2d(afx_ctx*, void*, const void*, size_t) { return no_device("afx_memcpy_h2d"); }
int afx_memcpy_d2h(afx_ctx*, void*, const void*, size_t) { return no_device("afx_memcpy_d2h"); }
int afx_synchronize(afx_ctx*) { return no_device("afx_synchronize"); }
int afx_plan_create(afx_ctx*, const afx_params*, afx_plan** out) { if (out) *out = nullptr; return no_device("afx_plan_create"); }
void afx_plan_destroy(afx_plan*) {}
int afx_extract_batch(afx_plan*, const void*, int, int, const int64_t*, const int64_t*, int, int, float*, int32_t*, int64_t*,
                      int32_t*, float*, const int64_t*) { return no_device("afx_extract_batch"); }
int afx_extract_submit(afx_plan*, const void*, int, int, const int64_t*, const int64_t*, int, int, float*, int32_t*, int64_t*,
                       int32_t*, float*, const int64_t*) { return no_device("afx_extract_submit"); }
int afx_extract_collect(afx_plan*) { return no_device("afx_extract_collect"); }
int afx_f0_batch(afx_plan*, const void*, int, int, const int64_t*, const int64_t*, int, int, double, double, double*, int32_t*,
                 double*, const int64_t*) { return no_device("afx_f0_batch"); }
int afx_zcr_batch(afx_plan*, const void*, int, int, const int64_t*, const int64_t*, int, int, double*, const int64_t*,
                  int32_t*) { return no_device("afx_zcr_batch"); }
int afx_spectral_batch(afx_plan*, const void*, int, int, const int64_t*, const int64_t*, int, int, float*, const int64_t*,
                       int32_t*) { return no_device("afx_spectral_batch"); }
int afx_preprocess(afx_plan*, const float*, int64_t, float*, int64_t*, int64_t*, int32_t*) { return no_device("afx_preprocess"); }
int afx_plan_set_timing(afx_plan*, int) { return no_device("afx_plan_set_timing"); }
int afx_plan_get_timings(afx_plan*, float*, int32_t*, int) { return no_device("afx_plan_get_timings"); }
int afx_plan_get_intervals(afx_plan*, int, double*, double*, int, int32_t*) { return no_device("afx_plan_get_intervals"); }
}
