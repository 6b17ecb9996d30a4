/* Checks the quotient hac_persist_k (pyannote-video_amd/csrc/cluster.hip) uses for the size-weighted mean of a merge:
 * y = 1 / den once, then q = n y corrected twice through the exact residual n - den q.  It must equal the IEEE quotient n / den (what the
 * oracle's and the reference's division returns) for every numerator; den = sum of two track sizes.  Run by tests/test_host_logic.py. */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double urand(void) { return (rnd() >> 11) * (1.0 / 9007199254740992.0); }
int main(void) {
    long bad = 0, n_cases = 0;
    for (int rep = 0; rep < 400; ++rep) {
        for (int dd = 2; dd < 4000; ++dd) {
            double szi, szj;
            if (rep & 1) { szi = 1 + rnd() % 1000000; szj = 1 + rnd() % 1000000; }
            else { szi = 1 + rnd() % dd; szj = dd; }
            const double den = szi + szj, y = 1.0 / den;
            for (int t = 0; t < 60; ++t) {
                double a = urand() * 1.2, b = urand() * 1.2;
                if (t == 0) { a = 0; b = 0; } if (t == 1) a = b;
                if (t == 2) { a = ldexp(urand(), -30); b = ldexp(urand(), -40); }
                const double n = szi * a + szj * b;
                double q = n * y;
                double r = fma(-den, q, n);
                q = fma(r, y, q);
                r = fma(-den, q, n);
                q = fma(r, y, q);
                const double ref = n / den;
                ++n_cases;
                if (q != ref) { if (bad < 5) printf("bad: n=%a den=%a q=%a ref=%a\n", n, den, q, ref); ++bad; }
            }
        }
    }
    printf("%ld cases, %ld differ\n", n_cases, bad);
    return bad != 0;
}
