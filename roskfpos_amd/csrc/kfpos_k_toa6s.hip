/*
 * kfpos_k_toa6s.hip -- k_step_toa6 / k_step_toa6_w2, symmetric (packed) covariance layout: banks with a fixed start
 */
#include "kfpos_kernels.h"

namespace {

#include "kfpos_k_toa6.inc"

} // namespace

template <typename REAL, typename MREAL>
static kfpos_k::step_kernel_t toa6_sym(int as, int heur, bool two_waves) { return toa6_kernel<true, REAL, MREAL>(as, heur, two_waves); }
kfpos_k::step_kernel_t kfpos_k::toa6_sym_kernel(int st, int as, int heur, bool two_waves) {
    return KFPOS_BY_STORAGE(st, toa6_sym, as, heur, two_waves);
}
