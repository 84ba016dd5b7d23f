/* CPU ORACLE (test infrastructure, NOT the product) -- plain-C fp64 restatement of
 * gr-baz's MUSIC-DoA work():
 *     /root/reference/lib/baz_music_doa.cc:72-161
 * Dependency-free: the Armadillo calls of the reference (cx_mat product .cc:85,
 * eig_sym .cc:88-90, cols .cc:93, trans/norm .cc:114-119) are restated with plain
 * loops and an own cyclic complex Jacobi Hermitian eigensolver.  Armadillo forwards
 * eig_sym to LAPACK zheev/zheevd; it is an un-vendored, un-versioned system
 * dependency of the reference (cmake/Modules/FindArmadillo.cmake:35-73), absent
 * from /root/reference.  The spectrum depends only on the noise-subspace projector
 * G G^H, so any correct fp64 Hermitian EVD gives the same result to ~1e-13
 * (SURVEY.md Appendix C); tests/test_oracle.py checks this file against the numpy
 * (zheevd) restatement and against oracle/_ref (the reference's own .cc built on
 * an API shim) when present.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (gr_baz_amd/) never links or calls it.
 *
 * PARITY PINNING: the reference holds no golden vectors or tests for this path
 * (lib/qa_baz.cc:32-41, python/qa_baz.py:34-56 are empty) -- see DESIGN.md sec. 3.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MUSIC_REF_MAX_M 64

typedef double complex cplx;

/* Cyclic Jacobi for an m x m Hermitian matrix A (row-major, overwritten).
 * On return w[k] ascending, V[:,k] (row-major V[r*m+k]) the unit eigenvectors --
 * the contract of arma::eig_sym(colvec, cx_mat, R), lib/baz_music_doa.cc:88-90. */
static void herm_eig_jacobi(int m, cplx *A, double *w, cplx *V)
{
    int i, j, k, p, q, sweep;
    for (i = 0; i < m; i++)
        for (j = 0; j < m; j++)
            V[i * m + j] = (i == j) ? 1.0 : 0.0;

    for (sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, dia = 0.0;
        for (i = 0; i < m; i++) {
            dia += creal(A[i * m + i]) * creal(A[i * m + i]);
            for (j = i + 1; j < m; j++) {
                double re = creal(A[i * m + j]), im = cimag(A[i * m + j]);
                off += re * re + im * im;
            }
        }
        if (off <= 1e-34 * dia || off == 0.0)
            break;
        for (p = 0; p < m - 1; p++) {
            for (q = p + 1; q < m; q++) {
                cplx apq = A[p * m + q];
                double g = cabs(apq);
                if (g == 0.0)
                    continue;
                {
                    double app = creal(A[p * m + p]), aqq = creal(A[q * m + q]);
                    cplx u = apq / g;            /* e^{i phi} */
                    cplx ub = conj(u);
                    double tau = (aqq - app) / (2.0 * g);
                    double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    double c = 1.0 / sqrt(1.0 + t * t);
                    double s = t * c;
                    /* J = [[c, s],[-s*conj(u), c*conj(u)]] on (p,q);  A <- J^H A J, V <- V J */
                    for (k = 0; k < m; k++) {    /* columns p,q:  A J */
                        cplx akp = A[k * m + p], akq = A[k * m + q];
                        A[k * m + p] = c * akp - s * ub * akq;
                        A[k * m + q] = s * akp + c * ub * akq;
                    }
                    for (k = 0; k < m; k++) {    /* rows p,q:  J^H (A J) */
                        cplx apk = A[p * m + k], aqk = A[q * m + k];
                        A[p * m + k] = c * apk - s * u * aqk;
                        A[q * m + k] = s * apk + c * u * aqk;
                    }
                    A[p * m + q] = 0.0;
                    A[q * m + p] = 0.0;
                    A[p * m + p] = creal(A[p * m + p]);
                    A[q * m + q] = creal(A[q * m + q]);
                    for (k = 0; k < m; k++) {
                        cplx vkp = V[k * m + p], vkq = V[k * m + q];
                        V[k * m + p] = c * vkp - s * ub * vkq;
                        V[k * m + q] = s * vkp + c * ub * vkq;
                    }
                }
            }
        }
    }
    /* ascending order (stable selection sort on the eigenvalue, swapping columns) */
    for (i = 0; i < m; i++)
        w[i] = creal(A[i * m + i]);
    for (i = 0; i < m - 1; i++) {
        int best = i;
        for (j = i + 1; j < m; j++)
            if (w[j] < w[best])
                best = j;
        if (best != i) {
            double tw = w[i];
            w[i] = w[best];
            w[best] = tw;
            for (k = 0; k < m; k++) {
                cplx tv = V[k * m + i];
                V[k * m + i] = V[k * m + best];
                V[k * m + best] = tv;
            }
        }
    }
}

/* One item.  in: N complex64 as interleaved floats (re,im), antenna-interleaved
 * x(r,c)=in[c*m+r]; table: res*m complex64 row-major [bin][antenna].
 * ang, lvl: n floats; spectrum: res floats or NULL; eigvals_out: m doubles or NULL.
 * Returns 1 like the reference (.cc:160), or -1 on bad arguments. */
int music_ref_work(const float *in_ri, const float *table_ri, unsigned m, unsigned n,
                   unsigned nsamples, unsigned resolution,
                   float *ang, float *lvl, float *spectrum, double *eigvals_out)
{
    unsigned i, j, c, t, k, step;
    unsigned K;
    cplx R[MUSIC_REF_MAX_M * MUSIC_REF_MAX_M], V[MUSIC_REF_MAX_M * MUSIC_REF_MAX_M];
    double w[MUSIC_REF_MAX_M];
    cplx *data;
    double *top_ang, *top_str;

    if (m == 0 || m > MUSIC_REF_MAX_M || n == 0 || n >= m || nsamples == 0 ||
        (nsamples % m) != 0 || resolution == 0)
        return -1;

    /* .cc:74-77  widen c64 -> c128 */
    data = (cplx *)malloc(sizeof(cplx) * nsamples);
    for (i = 0; i < nsamples; i++)
        data[i] = (double)in_ri[2 * i] + I * (double)in_ri[2 * i + 1];

    /* .cc:82-85  x = reshape(m, K) column-major => x(r,c) = data[c*m+r];  R = x x^H / K */
    K = nsamples / m;
    for (i = 0; i < m; i++)
        for (j = 0; j < m; j++) {
            cplx acc = 0.0;
            for (c = 0; c < K; c++)
                acc += data[c * m + i] * conj(data[c * m + j]);
            R[i * m + j] = acc / (double)K;
        }

    /* .cc:88-90  eig_sym: ascending eigenvalues, eigenvectors in columns */
    herm_eig_jacobi((int)m, R, w, V);
    if (eigvals_out)
        for (i = 0; i < m; i++)
            eigvals_out[i] = w[i];

    /* .cc:93  G = eigvec.cols(0, m-n-1) -> columns 0..m-n-1 of V */
    /* .cc:95  vDOAs(d_n, (0,0)) */
    top_ang = (double *)calloc(n, sizeof(double));
    top_str = (double *)calloc(n, sizeof(double));

    for (step = 0; step < resolution; step++) {           /* .cc:103 */
        double ss = 0.0, nrm, strength;
        for (k = 0; k < m - n; k++) {                     /* trans(G) * a   .cc:116/118 */
            cplx ck = 0.0;
            for (t = 0; t < m; t++) {
                cplx a = (double)table_ri[2 * (step * m + t)] +
                         I * (double)table_ri[2 * (step * m + t) + 1];   /* .cc:110-112 */
                ck += conj(V[t * m + k]) * a;
            }
            ss += creal(ck) * creal(ck) + cimag(ck) * cimag(ck);
        }
        nrm = sqrt(ss);                                   /* arma::norm(., 2) */
        strength = 1.0 / pow(nrm, 2);                     /* .cc:114-119 */
        if (spectrum)
            spectrum[step] = (float)strength;             /* .cc:120-121 */
        for (i = 0; i < n; i++) {                         /* .cc:129-141 */
            if (strength > top_str[i]) {
                double angle = (double)step * 360.0 / (double)resolution;
                for (j = n - 1; j > i; j--) {             /* insert at i, pop_back */
                    top_ang[j] = top_ang[j - 1];
                    top_str[j] = top_str[j - 1];
                }
                top_ang[i] = angle;
                top_str[i] = strength;
                break;
            }
        }
    }
    for (i = 0; i < n; i++) {                             /* .cc:146-155 */
        ang[i] = (float)top_ang[i];
        if (lvl)
            lvl[i] = (float)top_str[i];
    }
    free(top_ang);
    free(top_str);
    free(data);                                           /* .cc:158 */
    return 1;                                             /* .cc:160 */
}

/* Drive work() once per item, exactly as the GNU Radio scheduler drives the
 * reference block (one item consumed/produced per call). */
int music_ref_work_batch(const float *in_ri, unsigned batch, const float *table_ri,
                         unsigned m, unsigned n, unsigned nsamples, unsigned resolution,
                         float *ang, float *lvl, float *spectrum)
{
    unsigned b;
    for (b = 0; b < batch; b++) {
        int r = music_ref_work(in_ri + (size_t)b * nsamples * 2, table_ri, m, n, nsamples,
                               resolution, ang + (size_t)b * n,
                               lvl ? lvl + (size_t)b * n : NULL,
                               spectrum ? spectrum + (size_t)b * resolution : NULL, NULL);
        if (r != 1)
            return r;
    }
    return (int)batch;
}

/* Exposed for tests: Hermitian EVD of a row-major m x m complex128 matrix given as
 * interleaved doubles. w: m doubles ascending; V_ri: m*m*2 doubles row-major. */
int music_ref_eig(unsigned m, const double *A_ri, double *w, double *V_ri)
{
    cplx A[MUSIC_REF_MAX_M * MUSIC_REF_MAX_M], V[MUSIC_REF_MAX_M * MUSIC_REF_MAX_M];
    unsigned i;
    if (m == 0 || m > MUSIC_REF_MAX_M)
        return -1;
    for (i = 0; i < m * m; i++)
        A[i] = A_ri[2 * i] + I * A_ri[2 * i + 1];
    herm_eig_jacobi((int)m, A, w, V);
    for (i = 0; i < m * m; i++) {
        V_ri[2 * i] = creal(V[i]);
        V_ri[2 * i + 1] = cimag(V[i]);
    }
    return 0;
}
