// K0+K1 (temporal_impl.h) instantiated for one sample format.
#include "temporal_impl.h"
namespace cvvdp {
void launch_fir_yuv16(const FirArgs& a, float* hist_shadow, hipStream_t s) { launch_fir_typed<CVVDP_YUV16>(a, hist_shadow, s); }
}  // namespace cvvdp
