/* TEST INFRASTRUCTURE: exhaustive check of the reciprocal division the streaming kernels use for the border
 * renormalisation at the image edges (csrc/isp_stream.h, div16_by<T>):
 *     q = a * RN(1/T);  e = fma(-q, T, a);  q' = fma(e, RN(1/T), q)   ==   a / T   (IEEE, round to nearest even)
 * for every finite float a with |a| >= 2^-120 (so that the quotient is a normal number) or a == +0, T in {10..22} \ {16} (every in-bounds weight sum of
 * bayer.py:143-149 that occurs at a row or column border or in a corner: tile::make_border_table).   usage: check_recip_div [stride]   (stride 1 = all 2^32 bit patterns) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char** argv) {
  const int64_t stride = argc > 1 ? atoll(argv[1]) : 1;
  const float ds[] = {10, 11, 12, 13, 14, 15, 17, 18, 19, 20, 21, 22};
  int rc = 0;
  for (unsigned k = 0; k < sizeof(ds) / sizeof(ds[0]); ++k) {
    const float d = ds[k], y = 1.0f / d;
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t i = 0; i < ((int64_t)1 << 32); i += stride) {
      const uint32_t u = (uint32_t)i;
      float a;
      memcpy(&a, &u, 4);
      if (!isfinite(a) || (fabsf(a) < 0x1p-120f && u != 0)) continue;      /* NaN, inf, tiny (subnormal quotient), -0 */
      const float ref = a / d;
      float q = a * y;
      const float e = fmaf(-q, d, a);
      q = fmaf(e, y, q);
      uint32_t x, r;
      memcpy(&x, &q, 4);
      memcpy(&r, &ref, 4);
      if (x != r) bad++;
    }
    printf("T=%g: %s (%ld mismatches)\n", d, bad ? "FAIL" : "ok", bad);
    rc |= bad != 0;
  }
  return rc;
}
