/* Is a / b == fma(fma(-b, a * y, a), y, a * y) with y = 1.0f / b for EVERY
 * float a? (The short form of a correctly rounded division by a constant: one
 * multiplication and two fused multiply-adds instead of the ~12-instruction
 * IEEE sequence.) Enumerates all 2^32 numerators for the block and voxel sizes
 * the ray cast divides by -- 0.128 / 0.008 (8 mm grid), 0.064 / 0.004 (4 mm
 * grid) -- and counts the numerators whose quotient differs.
 *
 *   gcc -O2 -fopenmp -ffp-contract=off -o /tmp/chk tools/check_div_by_const.c -lm && /tmp/chk
 *
 * Result (round 3, 92 s on the build host): 0 mismatches for |a| >= 1e-30
 * and a finite quotient, for all four constants; ~9 M mismatches below 1e-30
 * (the residual is no longer exact once a * y is denormal) -- so the form is
 * exact behind a guard on |a|, which is what the integrate role does for ITS
 * constants after the same enumeration on the device (vbg_stream.hip
 * VerifyFastDivision). Tried in the ray cast at the end of round 3 (a march step
 * makes six such divisions): bit-identical maps, -1.4 % at VGA, but the kernel
 * then spilled under its 96-register cap; not kept ("Open"). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float fl(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main(void) {
    const float bs[4] = {0.128f, 0.008f, 0.064f, 0.004f};
    for (int k = 0; k < 4; ++k) {
        const float b = bs[k];
        const float y = 1.0f / b;
        unsigned long long bad = 0, bad_tiny = 0;
#pragma omp parallel for reduction(+ : bad, bad_tiny) schedule(static)
        for (long long i = 0; i < (1ll << 32); ++i) {
            const float a = fl((uint32_t)i);
            if (!isfinite(a)) continue;
            const float want = a / b;
            if (!isfinite(want)) continue; /* |a| > ~4e35: never a position */
            const float q0 = a * y;
            const float r = fmaf(-b, q0, a);
            const float q = fmaf(r, y, q0);
            if (bits(want) != bits(q)) {
                if (fabsf(a) < 1e-30f) ++bad_tiny;
                else ++bad;
            }
        }
        printf("b = %g (y = %.9g): %llu mismatches for |a| >= 1e-30, %llu below\n",
               b, y, bad, bad_tiny);
    }
    return 0;
}
