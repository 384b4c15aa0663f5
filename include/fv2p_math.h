/*
 * Deterministic fp32 sin / cos / atan2 shared by the HIP kernels (device) and by host code.
 *
 * Several integer outputs of the hot path (NMS survivors, point-in-box indices) are decided by
 * comparisons on values that pass through sinf/cosf/atan2f (reference:
 * pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:49-102, roiaware_pool3d_kernel.cu:16-36).  Vendor
 * math libraries (CUDA libm, ROCm ocml, glibc) disagree in the last ulp, which would make those
 * integer outputs platform dependent.  These routines use only IEEE-754 basic operations
 * (+ - * / floor, no fused multiply-add: build with -ffp-contract=off), so they return bit-identical
 * results on gfx950 and on the host; accuracy is within 2 ulp of the correctly rounded value for
 * |x| <= 8192 (checked in tests/test_math.py).  Polynomials: Cephes single-precision (sinf.c, atanf.c).
 */
#ifndef FV2P_MATH_H_
#define FV2P_MATH_H_

#if defined(__HIPCC__)
#define FV2P_HD __host__ __device__ __forceinline__
#else
#define FV2P_HD static inline
#endif

#ifndef FV2P_MATH_NO_STD
#include <math.h>
#endif

#define FV2P_PI_F 3.14159265358979323846f
#define FV2P_PIO2_F 1.57079632679489661923f
#define FV2P_PIO4_F 0.78539816339744830962f

/* sin (want_cos == 0) or cos (want_cos != 0) of x */
FV2P_HD float fv2p_sincos_impl(float x, int want_cos) {
  const float FOPI = 1.27323954473516f; /* 4/pi */
  const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
  float sign = 1.0f;
  float ax = x;
  if (x < 0.0f) {
    ax = -x;
    if (!want_cos) sign = -1.0f;
  }
  int j = (int)(FOPI * ax);
  float y = (float)j;
  if (j & 1) { /* map zeros to origin */
    j += 1;
    y += 1.0f;
  }
  j &= 7;
  if (j > 3) {
    sign = -sign;
    j -= 4;
  }
  if (want_cos && j > 1) sign = -sign;
  /* extended precision modular arithmetic */
  float r = ((ax - y * DP1) - y * DP2) - y * DP3;
  float z = r * r;
  int use_cos_poly = (j == 1 || j == 2);
  if (want_cos) use_cos_poly = !use_cos_poly;
  float v;
  if (use_cos_poly) {
    v = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z;
    v = v - 0.5f * z;
    v = v + 1.0f;
  } else {
    v = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * r;
    v = v + r;
  }
  return sign * v;
}

FV2P_HD float fv2p_sinf(float x) { return fv2p_sincos_impl(x, 0); }
FV2P_HD float fv2p_cosf(float x) { return fv2p_sincos_impl(x, 1); }

FV2P_HD float fv2p_atanf(float xx) {
  float x = xx, sign = 1.0f, y;
  if (xx < 0.0f) {
    sign = -1.0f;
    x = -xx;
  }
  if (x > 2.414213562373095f) { /* tan 3pi/8 */
    y = FV2P_PIO2_F;
    x = -(1.0f / x);
  } else if (x > 0.4142135623730950f) { /* tan pi/8 */
    y = FV2P_PIO4_F;
    x = (x - 1.0f) / (x + 1.0f);
  } else {
    y = 0.0f;
  }
  float z = x * x;
  y = y + ((((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x);
  return sign * y;
}

FV2P_HD float fv2p_atan2f(float y, float x) {
  if (x == 0.0f) {
    if (y > 0.0f) return FV2P_PIO2_F;
    if (y < 0.0f) return -FV2P_PIO2_F;
    return 0.0f;
  }
  float z = fv2p_atanf(y / x);
  if (x < 0.0f) z = (y >= 0.0f) ? z + FV2P_PI_F : z - FV2P_PI_F;
  return z;
}

#endif /* FV2P_MATH_H_ */
