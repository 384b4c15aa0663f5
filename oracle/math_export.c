/* ORACLE — test infrastructure only: exposes include/fv2p_math.h (the deterministic fp32 trig shared by
 * host and device) to the CPU tests so its accuracy can be checked against libm. */
#include <stdint.h>
#include "../include/fv2p_math.h"

void oracle_math_eval(const float* x, const float* y, int64_t n, float* s, float* c, float* a) {
  for (int64_t i = 0; i < n; ++i) {
    s[i] = fv2p_sinf(x[i]);
    c[i] = fv2p_cosf(x[i]);
    a[i] = fv2p_atan2f(y[i], x[i]);
  }
}
