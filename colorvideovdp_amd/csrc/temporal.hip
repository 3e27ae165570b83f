// K0+K1 dispatch by sample format; the kernels live in temporal_impl.h, one translation unit per format.
#include "kernels.h"

namespace cvvdp {

void launch_fir_u8(const FirArgs& a, float* hist_shadow, hipStream_t s);
void launch_fir_u16(const FirArgs& a, float* hist_shadow, hipStream_t s);
void launch_fir_f16(const FirArgs& a, float* hist_shadow, hipStream_t s);
void launch_fir_f32(const FirArgs& a, float* hist_shadow, hipStream_t s);
void launch_fir_dkl(const FirArgs& a, float* hist_shadow, hipStream_t s);
void launch_fir_yuv8(const FirArgs& a, float* hist_shadow, hipStream_t s);
void launch_fir_yuv16(const FirArgs& a, float* hist_shadow, hipStream_t s);

void launch_fir(const FirArgs& a, float* hist_shadow, hipStream_t s) {
  switch (a.dtype) {
    case CVVDP_U8: launch_fir_u8(a, hist_shadow, s); break;
    case CVVDP_U16: launch_fir_u16(a, hist_shadow, s); break;
    case CVVDP_F16: launch_fir_f16(a, hist_shadow, s); break;
    case CVVDP_F32: launch_fir_f32(a, hist_shadow, s); break;
    case CVVDP_YUV8: launch_fir_yuv8(a, hist_shadow, s); break;
    case CVVDP_YUV16: launch_fir_yuv16(a, hist_shadow, s); break;
    default: launch_fir_dkl(a, hist_shadow, s); break;
  }
}

}  // namespace cvvdp
