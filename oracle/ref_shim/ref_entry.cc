// ORACLE-ONLY (test infrastructure): C entry points around the reference's OWN
// baz_music_doa block (compiled from /root/reference/lib/baz_music_doa.cc, see
// oracle/Makefile target "ref"), plus the arma::eig_sym backend of the API shim.
// Drives work() exactly as the GNU Radio scheduler does: one item per call
// (lib/baz_music_doa.cc:160 returns 1).
#include <baz_music_doa.h>

#include <dlfcn.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {
typedef int (*lapacke_zheev_t)(int layout, char jobz, char uplo, int n, void* a, int lda, double* w);
lapacke_zheev_t g_zheev = nullptr;

void jacobi_herm(int m, std::vector<arma::cx_double>& A, std::vector<double>& w,
                 std::vector<arma::cx_double>& V)   // row-major A, V
{
    using arma::cx_double;
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) V[i * m + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, dia = 0;
        for (int i = 0; i < m; ++i) { dia += std::norm(A[i * m + i]); for (int j = i + 1; j < m; ++j) off += std::norm(A[i * m + j]); }
        if (off <= 1e-34 * dia || off == 0.0) break;
        for (int p = 0; p < m - 1; ++p) for (int q = p + 1; q < m; ++q) {
            cx_double apq = A[p * m + q];
            double g = std::abs(apq);
            if (g == 0.0) continue;
            cx_double u = apq / g, ub = std::conj(u);
            double tau = (A[q * m + q].real() - A[p * m + p].real()) / (2.0 * g);
            double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
            double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
            for (int k = 0; k < m; ++k) { cx_double x = A[k * m + p], y = A[k * m + q]; A[k * m + p] = c * x - s * ub * y; A[k * m + q] = s * x + c * ub * y; }
            for (int k = 0; k < m; ++k) { cx_double x = A[p * m + k], y = A[q * m + k]; A[p * m + k] = c * x - s * u * y; A[q * m + k] = s * x + c * u * y; }
            A[p * m + q] = 0; A[q * m + p] = 0;
            for (int k = 0; k < m; ++k) { cx_double x = V[k * m + p], y = V[k * m + q]; V[k * m + p] = c * x - s * ub * y; V[k * m + q] = s * x + c * ub * y; }
        }
    }
    for (int i = 0; i < m; ++i) w[i] = A[i * m + i].real();
}
}  // namespace

namespace arma {
bool eig_sym(colvec& eigval, cx_mat& eigvec, const cx_mat& X)
{
    const int m = (int)X.n_rows;
    if (X.n_rows != X.n_cols) throw std::logic_error("eig_sym(): given matrix must be square sized");
    eigval = colvec(m);
    eigvec = cx_mat(m, m);
    if (g_zheev) {   // LAPACK zheev, column-major, upper triangle -- what Armadillo calls
        std::vector<cx_double> a(X.mem);
        std::vector<double> w(m);
        int info = g_zheev(102 /*LAPACK_COL_MAJOR*/, 'V', 'U', m, a.data(), m, w.data());
        if (info != 0) return false;
        for (int i = 0; i < m; ++i) eigval[i] = w[i];
        eigvec.mem = a;
        return true;
    }
    std::vector<cx_double> A(m * m), V(m * m);
    std::vector<double> w(m);
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) A[i * m + j] = X(i, j);
    jacobi_herm(m, A, w, V);
    std::vector<int> idx(m);
    for (int i = 0; i < m; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return w[a] < w[b]; });
    for (int k = 0; k < m; ++k) {
        eigval[k] = w[idx[k]];
        for (int r = 0; r < m; ++r) eigvec(r, k) = V[r * m + idx[k]];
    }
    return true;
}
}  // namespace arma

extern "C" {

// Register a LAPACKE library (e.g. scipy's bundled OpenBLAS) for eig_sym. Returns 0 on success.
int baz_ref_set_lapack(const char* path, const char* symbol)
{
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    void* f = dlsym(h, symbol);
    if (!f) return -2;
    g_zheev = (lapacke_zheev_t)f;
    return 0;
}
void baz_ref_clear_lapack(void) { g_zheev = nullptr; }
int baz_ref_uses_lapack(void) { return g_zheev != nullptr; }

int baz_ref_work_batch(const float* in_ri, unsigned batch, const float* table_ri, unsigned m,
                       unsigned n, unsigned nsamples, unsigned resolution, float* ang, float* lvl,
                       float* spectrum)
{
    array_response_t table(resolution, antenna_response_t(m));
    for (unsigned s = 0; s < resolution; ++s)
        for (unsigned t = 0; t < m; ++t)
            table[s][t] = gr_complex(table_ri[2 * ((size_t)s * m + t)], table_ri[2 * ((size_t)s * m + t) + 1]);
    baz_music_doa_sptr blk = baz_make_music_doa(m, n, nsamples, table, resolution);
    for (unsigned b = 0; b < batch; ++b) {
        gr_vector_const_void_star in(1);
        gr_vector_void_star out;
        in[0] = in_ri + (size_t)b * nsamples * 2;
        out.push_back(ang + (size_t)b * n);
        out.push_back(lvl + (size_t)b * n);
        if (spectrum) out.push_back(spectrum + (size_t)b * resolution);
        int r = blk->work(1, in, out);
        if (r != 1) return -1;
    }
    return (int)batch;
}

}  // extern "C"
