// context.hip -- sfmhip context lifetime and error strings (include/sfmhip.h).
#include "common.h"
#include <string.h>

thread_local int g_sfmhip_last_hip_error = 0;

static int init_common(int device, hipStream_t stream, bool own, sfmhip_ctx** out) {
  if (!out) return SFMHIP_ERR_ARG;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_sfmhip_last_hip_error = (int)e;
    return SFMHIP_ERR_NO_DEVICE;  // no CPU fallback exists behind this ABI
  }
  if (device < 0 || device >= n) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  SFM_HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fprintf(stderr, "[sfmhip] device %d is %s; this library ships gfx950 code objects only\n", device,
            prop.gcnArchName);
    return SFMHIP_ERR_UNSUPPORTED;
  }
  sfmhip_ctx* c = new sfmhip_ctx();
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  if (own) {
    const hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      g_sfmhip_last_hip_error = (int)e;
      delete c;
      return SFMHIP_ERR_HIP;
    }
    c->own_stream = true;
  } else {
    c->stream = stream;
  }
  *out = c;
  return SFMHIP_OK;
}

extern "C" int sfmhip_device_count(void) {
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_sfmhip_last_hip_error = (int)e;
    return SFMHIP_ERR_NO_DEVICE;
  }
  return n;
}

extern "C" int sfmhip_init(int device, sfmhip_ctx** out) { return init_common(device, nullptr, true, out); }

extern "C" int sfmhip_init_on_stream(int device, void* hip_stream, sfmhip_ctx** out) {
  return init_common(device, (hipStream_t)hip_stream, false, out);
}

extern "C" void sfmhip_device_free(void* device_ptr) {
  if (device_ptr) hipFree(device_ptr);
}

extern "C" void sfmhip_host_free(void* host_ptr) { free(host_ptr); }

extern "C" int sfmhip_device_download(sfmhip_ctx* ctx, void* host_dst, const void* device_src, size_t bytes) {
  if (!ctx || (bytes && (!host_dst || !device_src))) return SFMHIP_ERR_ARG;
  if (!bytes) return SFMHIP_OK;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  SFM_HIP_TRY(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SFMHIP_OK;
}

extern "C" void sfmhip_shutdown(sfmhip_ctx* ctx) {
  if (!ctx) return;
  for (sfmhip_ctx* w : ctx->workers) sfmhip_shutdown(w);
  ctx->workers.clear();
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  if (ctx->ba_cache && ctx->ba_cache_free) ctx->ba_cache_free(ctx->ba_cache);  // (the problem sfmhip_ba_solve kept for a next call)
  ctx->ba_cache = nullptr;
  if (ctx->ba_host_scratch && ctx->ba_host_scratch_free) ctx->ba_host_scratch_free(ctx->ba_host_scratch);
  if (ctx->ba_arena) hipFree(ctx->ba_arena);
  if (ctx->ba_pinned) hipHostFree(ctx->ba_pinned);
  if (ctx->pinned) hipHostFree(ctx->pinned);
  for (void* p : ctx->dev_scratch)
    if (p) hipFree(p);
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int sfmhip_synchronize(sfmhip_ctx* ctx) {
  if (!ctx) return SFMHIP_ERR_ARG;
  SFM_HIP_TRY(hipSetDevice(ctx->device));
  SFM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SFMHIP_OK;
}

extern "C" int sfmhip_set_timing(sfmhip_ctx* ctx, int enable) {
  if (!ctx) return SFMHIP_ERR_ARG;
  ctx->timing = enable != 0;
  return SFMHIP_OK;
}

int sfm_ctx_pinned(sfmhip_ctx* ctx, size_t bytes, void** out) {
  if (ctx->pinned_bytes < bytes) {
    if (ctx->pinned) hipHostFree(ctx->pinned);
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    SFM_HIP_TRY(hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
    ctx->pinned_bytes = bytes;
  }
  *out = ctx->pinned;
  return SFMHIP_OK;
}

int sfm_ctx_dev_scratch(sfmhip_ctx* ctx, int which, size_t bytes, void** out) {
  if (which < 0 || which > 1) return SFMHIP_ERR_ARG;
  if (ctx->dev_scratch_bytes[which] < bytes) {
    if (ctx->dev_scratch[which]) hipFree(ctx->dev_scratch[which]);
    ctx->dev_scratch[which] = nullptr;
    ctx->dev_scratch_bytes[which] = 0;
    const size_t want = bytes + bytes / 4;  // (headroom: the next image of a set is about the same size)
    if (hipMalloc(&ctx->dev_scratch[which], want) != hipSuccess) return SFMHIP_ERR_ALLOC;
    ctx->dev_scratch_bytes[which] = want;
  }
  *out = ctx->dev_scratch[which];
  return SFMHIP_OK;
}

extern "C" int sfmhip_device(sfmhip_ctx* ctx) { return ctx ? ctx->device : -1; }
extern "C" void* sfmhip_stream(sfmhip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" const char* sfmhip_error_string(int status) {
  switch (status) {
    case SFMHIP_OK: return "ok";
    case SFMHIP_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
    case SFMHIP_ERR_HIP: return "HIP runtime error";
    case SFMHIP_ERR_ARG: return "invalid argument";
    case SFMHIP_ERR_ALLOC: return "device allocation failed";
    case SFMHIP_ERR_UNSUPPORTED: return "unsupported device or configuration";
    case SFMHIP_ERR_STATE: return "object not in the required state";
    case SFMHIP_ERR_COMM: return "all-reduce callback failed";
    case SFMHIP_ERR_TIMEOUT: return "a bounded spin inside a kernel ran out (scheduling fault or bug, not a property of the data)";
    default: return "unknown status";
  }
}

extern "C" int sfmhip_last_hip_error(void) { return g_sfmhip_last_hip_error; }
extern "C" int sfmhip_version(void) { return SFMHIP_VERSION; }
