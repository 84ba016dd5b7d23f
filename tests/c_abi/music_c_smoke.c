/* Plain-C consumer of include/baz_music_hip.h: the binding a C (or cgo / JNI / ctypes) host would write.
 * No C++, no HIP, no torch on this side of the boundary.  Synthesises one emitter on the unit-square array used by
 * SURVEY.md 8d, runs baz_music_process() on host buffers and prints the first DoA bin of every item.
 * Build: gcc -std=c99 -I include tests/c_abi/music_c_smoke.c -L gr_baz_amd/csrc -lbaz_music_hip -lm  */
#include <baz_music_hip.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define M 4
#define NEMIT 1
#define K 64
#define NS (M * K)
#define RES 360
#define BATCH 40

static const double PI = 3.14159265358979323846;

static unsigned long long rng_state = 88172645463325252ull;
static double uniform01(void)
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}
static double gauss(void) { return sqrt(-2.0 * log(uniform01() + 1e-300)) * cos(2.0 * PI * uniform01()); }

int main(int argc, char** argv)
{
    const double theta_deg = argc > 1 ? atof(argv[1]) : 137.0;
    const double pos[M][2] = {{0, 0}, {0.5, 0}, {0.5, 0.5}, {0, 0.5}};   /* half-wavelength unit square, lambda = 1 */
    static float table[RES * M * 2], in[BATCH * NS * 2], ang[BATCH * NEMIT], lvl[BATCH * NEMIT], spec[BATCH * RES];
    for (int s = 0; s < RES; ++s)
        for (int a = 0; a < M; ++a) {   /* python/music_doa_helper.py:40-41: exp(-j 2 pi (p . u) / lambda) */
            const double th = s * 360.0 / RES * PI / 180.0;
            const double ph = -2.0 * PI * (pos[a][0] * cos(th) + pos[a][1] * sin(th));
            table[2 * (s * M + a)] = (float)cos(ph);
            table[2 * (s * M + a) + 1] = (float)sin(ph);
        }
    const double th = theta_deg * PI / 180.0;
    for (int b = 0; b < BATCH; ++b)
        for (int c = 0; c < K; ++c) {
            const double sr = gauss(), si = gauss();
            for (int a = 0; a < M; ++a) {   /* x(r,c) = in[c*m + r], lib/baz_music_doa.cc:82-84 */
                const double ph = -2.0 * PI * (pos[a][0] * cos(th) + pos[a][1] * sin(th));
                const double cr = cos(ph), ci = sin(ph);
                float* o = in + 2 * ((size_t)b * NS + (size_t)c * M + a);
                o[0] = (float)(sr * cr - si * ci + 0.05 * gauss());
                o[1] = (float)(sr * ci + si * cr + 0.05 * gauss());
            }
        }
    baz_music_ctx* ctx = NULL;
    int rc = baz_music_create(&ctx, M, NEMIT, NS, RES, table, -1);
    if (rc != BAZ_MUSIC_OK) { fprintf(stderr, "create: %s\n", baz_music_strerror(rc)); return 2; }
    rc = baz_music_process(ctx, in, BATCH, ang, lvl, spec);
    if (rc != BATCH) { fprintf(stderr, "process: %s (%s)\n", baz_music_strerror(rc), baz_music_last_hip_error(ctx)); return 3; }
    printf("%s devices=%d\n", baz_music_version(), baz_music_device_count());
    for (int b = 0; b < BATCH; ++b) printf("item %d ang %.1f lvl %.6g argmax-consistent %d\n", b, ang[b], lvl[b],
                                           spec[b * RES + (int)(ang[b] * RES / 360.0 + 0.5)] == lvl[b]);
    baz_music_destroy(ctx);
    return 0;
}
