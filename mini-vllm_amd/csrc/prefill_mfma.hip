// K2 (MFMA path) — placeholder until the flash kernel lands: reports "does not
// apply" so that the dispatcher uses the vector-ALU kernel.
#include "prefill.h"
int mv_prefill_mfma_launch(const PrefillParams &p, int dtype, hipStream_t stream) {
  (void)p; (void)dtype; (void)stream;
  return MV_E_SHAPE;
}
