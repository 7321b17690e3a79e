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
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
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

}  // extern "C"
