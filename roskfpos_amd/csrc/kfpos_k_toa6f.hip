/*
 * kfpos_k_toa6f.hip -- k_step_toa6 / k_step_toa6_w2, full 6x6 covariance layout: banks that start with the ML initialisation (non-symmetric P)
 */
#include "kfpos_kernels.h"

namespace {

#include "kfpos_k_toa6.inc"

} // namespace

template <typename REAL, typename MREAL>
static kfpos_k::step_kernel_t toa6_full(int as, int heur) { return toa6_kernel<false, REAL, MREAL>(as, heur); }
kfpos_k::step_kernel_t kfpos_k::toa6_full_kernel(int st, int as, int heur) {
    return KFPOS_BY_STORAGE(st, toa6_full, as, heur);
}
