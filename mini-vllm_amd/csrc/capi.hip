// Version / error-string entry points of the C-ABI (include/mvhip.h).
#include "common.h"

extern "C" int mv_version(void) { return MV_ABI_VERSION; }

extern "C" const char *mv_strerror(int code) {
  switch (code) {
    case MV_OK: return "ok";
    case MV_E_NULLPTR: return "a required pointer argument is NULL";
    case MV_E_SHAPE: return "unsupported or inconsistent shape";
    case MV_E_DTYPE: return "unknown dtype";
    case MV_E_WORKSPACE: return "workspace too small or misaligned";
    case MV_E_ALIGN: return "pointer or stride not 16-byte aligned";
    case MV_E_RANGE: return "scalar argument out of range";
    case MV_E_ALIAS: return "an output buffer aliases an input that is still being read";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown mvhip error";
}

// a2 (device side): see mvhip.h.  One tiny launch at the end of the captured decode step.
__global__ void decode_advance_kernel(int64_t *ids, const int64_t *tokens, int64_t *pos, int32_t *slots,
                                      int32_t *ctx, int64_t *rng, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    ids[i] = tokens[i];
    pos[i] += 1;
    const int32_t c = ctx[i], s = slots[i];
    ctx[i] = c > 0 ? c + 1 : c;
    slots[i] = s >= 0 ? s + 1 : s;
  }
  if (i == 0 && rng) *rng += 1;
}

extern "C" int mv_decode_advance(int64_t *input_ids, const int64_t *tokens, int64_t *positions,
                                 int32_t *slot_mapping, int32_t *context_lens, int64_t *rng_offset,
                                 int batch, void *stream) {
  if (batch == 0) return MV_OK;
  if (!input_ids || !tokens || !positions || !slot_mapping || !context_lens) return MV_E_NULLPTR;
  if (batch < 0) return MV_E_SHAPE;
  hipLaunchKernelGGL(decode_advance_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     input_ids, tokens, positions, slot_mapping, context_lens, rng_offset, batch);
  MV_LAUNCH_CHECK();
  return MV_OK;
}
