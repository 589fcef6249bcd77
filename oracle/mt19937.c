/* C restatement of numpy's legacy RandomState(seed).standard_normal  (TEST ORACLE).
 *
 * Follows numpy 2.2 sources (not vendored in the reference; the reference reaches them via
 * sklearn's _initialize_nmf, sklearn/decomposition/_nmf.py:302-314):
 *   numpy/random/src/mt19937/mt19937.c         mt19937_seed (init_genrand), mt19937_gen
 *   numpy/random/src/mt19937/mt19937.h         mt19937_next_double: (a>>5, b>>6) -> 53-bit double
 *   numpy/random/src/legacy/legacy-distributions.c  legacy_gauss (polar method with cache)
 * Pinned against numpy itself in tests/test_oracle_rng.py; the device kernel
 * (cnmf_amd/csrc/kernels_rng.hip.h) is pinned against numpy in tests/test_gpu_nmf.py.
 * Build: make -C oracle   (gcc -O2 -fPIC -shared, no -ffast-math, no FMA contraction).
 */
#include <math.h>
#include <stdint.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t key[MT_N];
    int pos;
    int has_gauss;
    double gauss;
} mt_state;

static void mt_seed(mt_state* s, uint32_t seed)
{
    for (int i = 0; i < MT_N; ++i) {
        s->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    s->pos = MT_N;
    s->has_gauss = 0;
    s->gauss = 0.0;
}

static void mt_gen(mt_state* s)
{
    uint32_t y;
    int i;
    for (i = 0; i < MT_N - MT_M; ++i) {
        y = (s->key[i] & 0x80000000u) | (s->key[i + 1] & 0x7fffffffu);
        s->key[i] = s->key[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    for (; i < MT_N - 1; ++i) {
        y = (s->key[i] & 0x80000000u) | (s->key[i + 1] & 0x7fffffffu);
        s->key[i] = s->key[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    }
    y = (s->key[MT_N - 1] & 0x80000000u) | (s->key[0] & 0x7fffffffu);
    s->key[MT_N - 1] = s->key[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
    s->pos = 0;
}

static uint32_t mt_next32(mt_state* s)
{
    uint32_t y;
    if (s->pos == MT_N) mt_gen(s);
    y = s->key[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static double mt_next_double(mt_state* s)
{
    int32_t a = mt_next32(s) >> 5, b = mt_next32(s) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

static double legacy_gauss(mt_state* s)
{
    if (s->has_gauss) {
        const double t = s->gauss;
        s->has_gauss = 0;
        s->gauss = 0.0;
        return t;
    } else {
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * mt_next_double(s) - 1.0;
            x2 = 2.0 * mt_next_double(s) - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = sqrt(-2.0 * log(r2) / r2);
        s->gauss = f * x1;
        s->has_gauss = 1;
        return f * x2;
    }
}

/* out[0..n) = RandomState(seed).standard_normal(n) */
void oracle_standard_normal(uint32_t seed, int64_t n, double* out)
{
    mt_state s;
    mt_seed(&s, seed);
    for (int64_t i = 0; i < n; ++i) out[i] = legacy_gauss(&s);
}

/* sklearn init='random' for one restart: H0 [k][G] then W0 [N][k], float32, |avg*z| */
void oracle_random_init(uint32_t seed, double avg, int k, int64_t N, int64_t G, float* W0, float* H0)
{
    mt_state s;
    mt_seed(&s, seed);
    for (int64_t i = 0; i < (int64_t)k * G; ++i) H0[i] = fabsf((float)(avg * legacy_gauss(&s)));
    for (int64_t i = 0; i < N * k; ++i) W0[i] = fabsf((float)(avg * legacy_gauss(&s)));
}
