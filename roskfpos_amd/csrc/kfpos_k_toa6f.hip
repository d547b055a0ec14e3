/*
 * kfpos_k_toa6f.hip -- k_step_toa6 / k_step_toa6_w2, full 6x6 covariance layout: banks that start with the ML initialisation (non-symmetric P)
 */
#include "kfpos_kernels.h"

namespace {

#include "kfpos_k_toa6.inc"

} // namespace

kfpos_k::step_kernel_t kfpos_k::toa6_full_kernel(int st, int as, int heur) {
    return st == KFPOS_STORE_F32 ? toa6_kernel<false, float, float>(as, heur)
         : st == KFPOS_STORE_MIXED ? toa6_kernel<false, double, float>(as, heur)
                                   : toa6_kernel<false, double, double>(as, heur);
}
