// k_gru, fp32 build (v_mfma_f32_32x32x2_f32): all instantiations
#include "mtadgat_gru_impl.h"

namespace mtadgat {
int launch_gru_big_f32(const GruArgs& a, int ncg, int xmode, bool fc, bool two, hipStream_t s) { return launch_gru_big_t<false>(a, ncg, xmode, fc, two, s); }
}  // namespace mtadgat
