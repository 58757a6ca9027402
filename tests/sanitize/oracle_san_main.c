/* oracle/cv_remap_oracle.c under -fsanitize=address,undefined: every interpolation x border x channel count on
   maps full of the values the quantiser has to survive (NaN, infinities, +-1e9, ties, image edges). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int orc_remap_u8(const uint8_t*, int, int, int64_t, int, const float*, const float*, int64_t, uint8_t*, int, int, int64_t, int, const uint8_t*);
int orc_remap_nearest_u8(const uint8_t*, int, int, int64_t, int, const float*, const float*, int64_t, uint8_t*, int, int, int64_t, int, const uint8_t*);
int orc_remap_cubic_u8(const uint8_t*, int, int, int64_t, int, const float*, const float*, int64_t, uint8_t*, int, int, int64_t, int, const uint8_t*);

static uint32_t rng = 12345u;
static uint32_t next(void) { rng = rng * 1664525u + 1013904223u; return rng >> 8; }

int main(void)
{
    const float special[] = {NAN, INFINITY, -INFINITY, 1e9f, -1e9f, 40000.0f, -40000.0f, -1.0f, -0.03125f, 0.0f, 1.015625f, 2.046875f};
    unsigned long sum = 0;
    for (int cn = 1; cn <= 4; cn += (cn == 1 ? 2 : 1))
        for (int border = 0; border <= 4; ++border)
            for (int sz = 0; sz < 3; ++sz) {
                const int sw = sz == 0 ? 1 : (sz == 1 ? 37 : 256), sh = sz == 0 ? 1 : (sz == 1 ? 23 : 64), ow = 67, oh = 9;
                uint8_t* src = malloc((size_t)sw * sh * cn);
                uint8_t* dst = malloc((size_t)ow * oh * cn);
                float* U = malloc(sizeof(float) * ow * oh);
                float* V = malloc(sizeof(float) * ow * oh);
                for (int i = 0; i < sw * sh * cn; ++i) src[i] = (uint8_t)next();
                for (int i = 0; i < ow * oh; ++i) {
                    U[i] = (float)(next() % (unsigned)((sw + 6) * 32)) / 32.0f - 3.0f;
                    V[i] = (float)(next() % (unsigned)((sh + 6) * 32)) / 32.0f - 3.0f;
                    if (i < 12) { U[i] = special[i]; }
                    else if (i < 24) { V[i] = special[i - 12]; }
                }
                const uint8_t bv[4] = {1, 2, 3, 4};
                if (orc_remap_u8(src, sw, sh, (int64_t)sw * cn, cn, U, V, ow, dst, ow, oh, (int64_t)ow * cn, border, bv)) return 2;
                sum += dst[0] + dst[ow * oh * cn - 1];
                if (orc_remap_nearest_u8(src, sw, sh, (int64_t)sw * cn, cn, U, V, ow, dst, ow, oh, (int64_t)ow * cn, border, bv)) return 3;
                sum += dst[0] + dst[ow * oh * cn - 1];
                if (orc_remap_cubic_u8(src, sw, sh, (int64_t)sw * cn, cn, U, V, ow, dst, ow, oh, (int64_t)ow * cn, border, bv)) return 4;
                sum += dst[0] + dst[ow * oh * cn - 1];
                free(src); free(dst); free(U); free(V);
            }
    printf("oracle sanitizer run OK (%lu)\n", sum);
    return 0;
}
