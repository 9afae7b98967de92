// Host build of csrc/geev_small.h for the CPU tests (tests/test_geev_small_cpu.py): the same source the kernels compile.
#include "../bayesian_cbf_amd/csrc/geev_small.h"
extern "C" int geev_small_eig(int n, const double* A_in, double* wr, double* V_out) {
    double A[4][4], V[4][4], w[4];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i][j] = A_in[i * n + j];
    const int rc = bcbf::geev::geev_real(n, A, w, V);
    for (int i = 0; i < n; ++i) { wr[i] = w[i]; for (int j = 0; j < n; ++j) V_out[i * n + j] = V[i][j]; }
    return rc;
}
extern "C" int geev_small_clean(int n, double* H_io, double eps, int mode) {
    double H[4][4];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) H[i][j] = H_io[i * n + j];
    const int rc = bcbf::geev::clean_hessian(n, H, eps, mode);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) H_io[i * n + j] = H[i][j];
    return rc;
}
