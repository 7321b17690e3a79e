// Context management and host-pointer staging for the C ABI (include/bpgpu.h).
#include "engine.h"

extern "C" {

int bp_version(void) { return 100; }

int bp_ctx_create(bp_ctx** out, int device) {
  if (!out) return BP_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return BP_ERR_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return BP_ERR_DEVICE;
  bp_ctx* ctx = new bp_ctx();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return BP_ERR_DEVICE;
  }
  *out = ctx;
  return BP_OK;
}

void bp_ctx_destroy(bp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->msm_ws.release();
  ctx->io.release();
  ctx->gen.release();
  ctx->vws.release();
  ctx->pws.release();
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  for (int i = 0; i < 2; i++)
    if (ctx->msm_ev[i]) (void)hipEventDestroy(ctx->msm_ev[i]);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* bp_last_error(const bp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int bp_ctx_sync(bp_ctx* ctx) {
  if (!ctx) return BP_ERR_ARG;
  BP_TRY(hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

void* bp_ctx_stream(bp_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int bp_ctx_profile(bp_ctx* ctx, int enable) {
  if (!ctx) return BP_ERR_ARG;
  ctx->profile = enable ? 1 : 0;
  for (int i = 0; i < 2; i++) {
    ctx->prof_ms[i] = 0;
    ctx->prof_calls[i] = 0;
    ctx->prof_madds[i] = 0;
  }
  ctx->prof_pending_light = false;
  return BP_OK;
}

int bp_ctx_profile_read(bp_ctx* ctx, int kernel, double* total_ms, uint64_t* launches, double* mixed_adds) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(kernel == 0 || kernel == 1, "kernel must be 0 (k_fixed_multiexp) or 1 (k_light)");
  BP_REQUIRE(total_ms && launches && mixed_adds, "null pointer");
  if (kernel == 1 && ctx->prof_pending_light) {  // fold in the last MSM launch
    BP_TRY(hipSetDevice(ctx->device));
    float ms = 0;
    BP_TRY(hipEventSynchronize(ctx->msm_ev[1]));
    BP_TRY(hipEventElapsedTime(&ms, ctx->msm_ev[0], ctx->msm_ev[1]));
    ctx->prof_ms[1] += ms;
    ctx->prof_calls[1] += 1;
    ctx->prof_pending_light = false;
  }
  *total_ms = ctx->prof_ms[kernel];
  *launches = ctx->prof_calls[kernel];
  *mixed_adds = ctx->prof_madds[kernel];
  return BP_OK;
}

}  // extern "C"
