// Is q1 = fmaf(r, rc, q0), r = fmaf(-q0, b, a), q0 = a * rc, rc = RN(1/b) the correctly rounded a / b in float32?
// All 2^23 significands of a in [1, 2), every divisor b = 1..64 (the quotient's rounding is scale invariant away
// from the subnormal / overflow ranges, which range sums never reach).
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdint.h>
int main(void) {
    long bad_total = 0;
    for (int b = 1; b <= 64; ++b) {
        const float fb = (float)b, rc = 1.0f / fb;
        long bad = 0;
        for (uint32_t m = 0; m < (1u << 23); ++m) {
            uint32_t bits = 0x3f800000u | m;
            float a; memcpy(&a, &bits, 4);
            for (int e = 0; e < 2; ++e) {           // a in [1,2) and a*37 (another binade / alignment)
                const float aa = e ? a * 32.0f : a;
                const float q0 = aa * rc;
                const float r = fmaf(-q0, fb, aa);
                const float q1 = fmaf(r, rc, q0);
                if (q1 != aa / fb) ++bad;
            }
        }
        if (bad) printf("b=%d bad=%ld\n", b, bad);
        bad_total += bad;
    }
    printf("total mismatches: %ld\n", bad_total);
    return 0;
}
