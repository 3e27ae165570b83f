// K8: do_pooling_and_jods + met2jod (cvvdp_metric.py:610-658).  Q_per_ch[B,C,F,bands] -> JOD[B].
// Tiny; one 256-thread block per batch item, threads stride over frames.
#include "kernels.h"

namespace cvvdp {

__device__ __forceinline__ float spow(float x, float p) { return powf(x + kEps, p) - powf(kEps, p); }

__global__ __launch_bounds__(256) void k_pool(PoolArgs a) {
  __shared__ float s_tmp[4];
  const int b = blockIdx.x, t = threadIdx.x;
  float part = 0.0f, single = 0.0f;
  for (int f = t; f < a.F; f += 256) {
    float tc = 0.0f;
    for (int c = 0; c < a.C; ++c) {
      const float* q = a.q + (((int64_t)b * a.C + c) * a.F + f) * a.L;
      float sc = 0.0f;
      for (int l = 0; l < a.L; ++l) {
        const float wb = (l == a.L - 1) ? a.bb_w[c] : 1.0f;
        sc += spow(q[l] * a.ch_w[c] * wb, a.beta_sch);
      }
      const float Qsc = spow(sc, 1.0f / a.beta_sch);   // over bands, not normalised (:625)
      tc += spow(Qsc, a.beta_tch);
    }
    const float Qtc = spow(tc, 1.0f / a.beta_tch);     // over channels (:633)
    single = Qtc;
    part += spow(Qtc, a.beta_t);
  }
  // block sum over frames
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if ((t & 63) == 0) s_tmp[t >> 6] = part;
  __syncthreads();
  if (t == 0) {
    float Q;
    if (a.F == 1) {
      Q = single * a.image_int;                          // :636
    } else {
      const float s = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
      Q = spow(s / (float)a.F, 1.0f / a.beta_t);         // over frames, normalised (:638)
    }
    const float Qt = 0.1f;                               // met2jod, :646-658
    float jod;
    if (Q <= Qt) jod = 10.0f - a.jod_a * powf(Qt, a.jod_exp - 1.0f) * Q;
    else jod = 10.0f - a.jod_a * powf(Q, a.jod_exp);
    a.jod[b] = jod;
  }
}

void launch_pool(const PoolArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_pool, dim3(a.B), dim3(256), 0, s, a); }

}  // namespace cvvdp
