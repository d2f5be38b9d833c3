// k_gru, bf16 operand build (v_mfma_f32_32x32x16_bf16): all instantiations
#include "mtadgat_gru_impl.h"

namespace mtadgat {
int launch_gru_big_bf16(const GruArgs& a, int ncg, int xmode, bool fc, bool two, hipStream_t s) { return launch_gru_big_t<true>(a, ncg, xmode, fc, two, s); }
}  // namespace mtadgat
