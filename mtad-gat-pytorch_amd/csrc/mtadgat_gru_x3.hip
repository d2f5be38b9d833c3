// k_gru, split-bf16 build (three bf16 pieces per operand, six v_mfma_f32_32x32x16_bf16 per product): hidden sizes above 128
#include "mtadgat_gru_impl.h"

namespace mtadgat {
int launch_gru_big_x3_hi(const GruArgs& a, int ncg, int xmode, bool fc, bool two, hipStream_t s) { return launch_gru_big_split<true>(a, ncg, xmode, fc, two, s); }
}  // namespace mtadgat
