/*
 * kfpos_k_toa6s.hip -- k_step_toa6 / k_step_toa6_w2, symmetric (packed) covariance layout: banks with a fixed start
 */
#include "kfpos_kernels.h"

namespace {

#include "kfpos_k_toa6.inc"

} // namespace

kfpos_k::step_kernel_t kfpos_k::toa6_sym_kernel(int st, int as, int heur, bool two_waves) {
    return st == KFPOS_STORE_F32 ? toa6_kernel<true, float, float>(as, heur, two_waves)
         : st == KFPOS_STORE_MIXED ? toa6_kernel<true, double, float>(as, heur, two_waves)
                                   : toa6_kernel<true, double, double>(as, heur, two_waves);
}
