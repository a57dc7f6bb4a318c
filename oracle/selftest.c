/* Sanitizer self-test of the CPU oracle (test infrastructure): random compress -> inflate round trips,
 * random garbage into inflate, tiny output capacities.  Built with -fsanitize=address,undefined by
 * `make -C oracle selftest` and run by tests/test_oracle_sanitizers.py.  (The reference's own assertions
 * are MyHDL intbv range checks; this is their counterpart for the C restatement) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int hdlz_oracle_compress(const uint8_t*, size_t, int, int, uint8_t*, size_t, size_t*);
int hdlz_oracle_inflate(const uint8_t*, size_t, unsigned, uint32_t, uint8_t*, size_t, size_t*);
size_t hdlz_oracle_out_bound(size_t);
static uint32_t rng = 12345;
static uint32_t rnd(void) { rng = rng * 1664525u + 1013904223u; return rng >> 8; }
int main(void) {
    int cws[] = {1, 2, 3, 16, 31, 32, 33, 64, 100, 255, 256};
    for (int it = 0; it < 3000; it++) {
        size_t n = rnd() % (it % 50 == 0 ? 70000 : 600);
        int alpha = 1 + rnd() % (it % 3 == 0 ? 256 : 4);
        uint8_t* x = (uint8_t*)malloc(n + 1);
        for (size_t i = 0; i < n; i++) x[i] = (uint8_t)(rnd() % alpha + (alpha < 200 ? 40 : 0));
        int cw = cws[rnd() % 11], mm = rnd() % 2 ? 10 : 5;
        size_t cap = hdlz_oracle_out_bound(n), ol = 0, bl = 0;
        uint8_t* z = (uint8_t*)malloc(cap + 1);
        int rc = hdlz_oracle_compress(x, n, cw, mm, z, cap, &ol);
        if (n < 5) { if (rc != 1) { printf("short input rc %d\n", rc); return 1; } }
        else {
            if (rc != 0 || ol > cap) { printf("compress rc %d\n", rc); return 1; }
            uint8_t* back = (uint8_t*)malloc(n + 1);
            rc = hdlz_oracle_inflate(z, ol, 0, 0, back, n, &bl);
            if (rc != 0 || bl != n || memcmp(back, x, n)) { printf("round trip failed n=%zu cw=%d rc=%d\n", n, cw, rc); return 1; }
            /* capacity one short, trailer cut, random damage: must fail cleanly, never overrun */
            if (n) hdlz_oracle_inflate(z, ol, 0, 0, back, n - 1, &bl);
            hdlz_oracle_inflate(z, ol > 3 ? ol - 3 : 0, 0, 0, back, n, &bl);
            z[2 + rnd() % (ol - 2)] ^= (uint8_t)(1u << (rnd() % 8));
            hdlz_oracle_inflate(z, ol, rnd() % 2, rnd() % 2 ? 512 : 0, back, n, &bl);
            size_t small = cap / 2; uint8_t* zs = (uint8_t*)malloc(small + 1);
            hdlz_oracle_compress(x, n, cw, mm, zs, small, &ol);
            free(zs); free(back);
        }
        /* pure garbage */
        uint8_t g[64], o[256];
        for (int i = 0; i < 64; i++) g[i] = (uint8_t)rnd();
        hdlz_oracle_inflate(g, rnd() % 65, 0, 0, o, sizeof o, &bl);
        free(x); free(z);
    }
    printf("selftest OK\n");
    return 0;
}
