// Exhaustive check of k_describe.hip's sincos_2pi (same arithmetic, restated for the host) against the host libm for EVERY float
// angle in [0, 360] degrees: (float)cos((double)rad) and (float)sin((double)rad) must be identical.  ~60 s on one core;
// run by hand (gcc -O2 -ffp-contract=off test_sincos_exhaustive.c -lm); tests/test_host_logic.py runs a strided subset.
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static void sincos_2pi(double x, double* s_out, double* c_out) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632679489655800e+00;
  const double pio2_lo = 6.12323399573676603587e-17;
  const double kd = rint(x * two_over_pi);
  const int k = (int)kd;
  const double r = fma(-kd, pio2_lo, fma(-kd, pio2_hi, x));
  const double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2), S1);
  const double sr = fma(z * r, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * z * pc);
  const int swap = k & 1;
  const double sv = swap ? cr : sr, cv = swap ? sr : cr;
  *s_out = (k & 2) ? -sv : sv;
  *c_out = ((k + 1) & 2) ? -cv : cv;
}
int main(int argc, char** argv) {
  const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;  // 1 = every float
  // every float angle value reachable: angle_deg in [0,360] float, times (float)(pi/180)
  long bad = 0, n = 0; double maxerr = 0;
  float degf = (float)(3.14159265358979323846 / 180.f);
  for (uint32_t bits = 0; bits <= 0x43B40000u; bits += stride) {  // all floats in [0, 360]
    float a; memcpy(&a, &bits, 4);
    float rad = a * degf;
    double s, c; sincos_2pi((double)rad, &s, &c);
    float cf = (float)c, sf = (float)s;
    float cr = (float)cos((double)rad), sr = (float)sin((double)rad);
    if (cf != cr || sf != sr) { bad++; if (bad < 5) printf("mismatch a=%g rad=%.9g: %a %a vs %a %a\n", a, rad, cf, sf, cr, sr); }
    n++;
  }
  printf("checked %ld floats, mismatches %ld\n", n, bad);
  return bad != 0;
}
