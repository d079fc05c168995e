// common.h -- context object and error plumbing shared by the HIP translation units.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "../../include/sfmhip.h"

extern thread_local int g_sfmhip_last_hip_error;

#define SFM_HIP_TRY(expr)                                                              \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess) {                                                           \
      g_sfmhip_last_hip_error = (int)e__;                                              \
      fprintf(stderr, "[sfmhip] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr,          \
              hipGetErrorString(e__));                                                 \
      return SFMHIP_ERR_HIP;                                                           \
    }                                                                                  \
  } while (0)

#define SFM_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != SFMHIP_OK) return rc__; \
  } while (0)

struct sfmhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool timing = false;  // record hipEvents between the stages of a run (sfmhip_set_timing)
  int n_cu = 0;
  // reusable pinned + device scratch for the one-shot host-pointer entry points
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // two grow-only device blocks for entry points that would otherwise hipMalloc / hipFree per call (sift.hip)
  void* dev_scratch[2] = {nullptr, nullptr};
  size_t dev_scratch_bytes[2] = {0, 0};
  // worker contexts (own stream + scratch) of the batched entry points (sfmhip_sift_batch): created on demand, freed
  // with the context
  std::vector<sfmhip_ctx*> workers;
  int score_flags = 0;  // OR of the five-point samples' flags of the last sfmhip_score_essential call (score.hip)
  // sfmhip_ba_solve (ba.hip): where the last call's time went, and the problem it keeps for a next call of the same structure
  sfmhip_ba_solve_profile ba_profile = {};
  void* ba_cache = nullptr;
  void (*ba_cache_free)(void*) = nullptr;
  // ... and the memory its problems live in: ONE grow-only device block that every problem of the one-shot entry point is carved
  // from (a problem is 60-odd hipMallocs and as many hipFrees otherwise: 6.4 ms of a 33 ms call at cfg4), and one pinned block
  // for the records the device writes to the host
  void* ba_arena = nullptr;
  size_t ba_arena_bytes = 0, ba_arena_need = 0;  // need: what the last problem took in all (the next call grows the block to it)
  void* ba_pinned = nullptr;
  size_t ba_pinned_bytes = 0;
  void* ba_host_scratch = nullptr;  // (and the host vectors of its set-ups: ba.hip, BaHostScratch)
  void (*ba_host_scratch_free)(void*) = nullptr;
};

int sfm_ctx_pinned(sfmhip_ctx* ctx, size_t bytes, void** out);
// block `which` (0 or 1) of at least `bytes` bytes; its contents do not survive a growing call.  The caller's work on
// it is ordered by the context's stream (the entry points that use it end with a stream synchronisation).
int sfm_ctx_dev_scratch(sfmhip_ctx* ctx, int which, size_t bytes, void** out);

template <typename T>
static inline int sfm_dev_alloc(T** p, size_t n) {
  void* v = nullptr;
  hipError_t e = hipMalloc(&v, (n ? n : 1) * sizeof(T));
  if (e != hipSuccess) {
    g_sfmhip_last_hip_error = (int)e;
    return SFMHIP_ERR_ALLOC;
  }
  // SFMHIP_POISON=1 (tests): fresh device memory holds 0xFF bytes (NaN as float / double, -1 as int) instead of
  // whatever the driver hands out -- a kernel that reads before it writes shows up as NaN, not as a rare flake
  static const bool poison = getenv("SFMHIP_POISON") != nullptr;
  if (poison) {
    // (the fill runs on the null stream: it must be over before a kernel on one of the library's own streams touches the
    // buffer -- an allocation made between two launches, like the front tree's second reduced-system buffer, raced with it)
    hipMemset(v, 0xFF, (n ? n : 1) * sizeof(T));
    hipDeviceSynchronize();
  }
  *p = (T*)v;
  return SFMHIP_OK;
}
