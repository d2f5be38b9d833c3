// k_gru, split-bf16 build: hidden sizes up to 128
#include "mtadgat_gru_impl.h"

namespace mtadgat {
int launch_gru_big_x3_lo(const GruArgs& a, int ncg, int xmode, bool fc, bool two, hipStream_t s) { return launch_gru_big_split<false>(a, ncg, xmode, fc, two, s); }
}  // namespace mtadgat
