// Host K x K eigensolve of the thick restart by split + rank-one merge: time of each phase (first part -- hidden under
// the GPU's cycle --, tail QL + merge set-up, last row of the eigenvectors, eigenvector columns) against the plain QL.
// g++ -O3 -std=c++17 -mavx2 -ffp-contract=off -pthread -o tools/micro/mergebench tools/micro/mergebench.cpp
#include "../../proxsdp.jl_amd/csrc/host_util.hpp"
#include <cstdio>
#include <random>
using namespace proxsdp;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 127, m = argc > 2 ? atoi(argv[2]) : 64, ncols = argc > 3 ? atoi(argv[3]) : 64;
    std::mt19937_64 g(1);
    std::normal_distribution<double> N(0.0, 1.0);
    std::vector<double> D(std::max(m, 1)), f(std::max(m, 1)), al(K), be(K);
    for (int j = 0; j < m; ++j) { D[j] = 10.0 - 0.1 * j + 0.01 * N(g); f[j] = 1e-3 * N(g); }
    for (int j = 0; j < K; ++j) { al[j] = 3.0 * N(g); be[j] = std::fabs(N(g)) + 0.1; }
    const int k1 = m == 0 ? (18 * K) / 25 : m + 1;
    SplitEig S;
    const int REP = 200;
    double t_first = 0, t_second = 0, t_row = 0, t_vec = 0, t_ql = 0;
    std::vector<double> row(K), U((size_t)K * K), Tw((size_t)K * K), Dasc(K);
    std::vector<int> cols(ncols);
    for (int c = 0; c < ncols; ++c) cols[c] = K - 1 - c;
    for (int r = 0; r < REP; ++r) {
        double t0 = now();
        if (S.first(k1, m, D.data(), f.data(), al.data(), be.data()) != 0) { std::printf("first failed\n"); return 1; }
        double t1 = now();
        if (S.second(K, al.data(), be.data()) != 0) { std::printf("second failed\n"); return 1; }
        double t2 = now();
        S.M.row_of_vectors(K - 1, row.data());
        double t3 = now();
        S.M.vectors(cols.data(), ncols, U.data());
        double t4 = now();
        // the plain path: dense K x K (arrow + tridiagonal) through the two-phase Householder + QL
        std::fill(Tw.begin(), Tw.end(), 0.0);
        for (int j = 0; j < m; ++j) { Tw[(size_t)j * K + j] = D[j]; Tw[(size_t)j * K + m] = Tw[(size_t)m * K + j] = f[j]; }
        for (int j = m; j < K; ++j) { Tw[(size_t)j * K + j] = al[j]; if (j + 1 < K) Tw[(size_t)j * K + j + 1] = Tw[(size_t)(j + 1) * K + j] = be[j]; }
        double t5 = now();
        symeig_dense(K, Tw.data(), Dasc.data(), m == 0, 0);
        double t6 = now();
        t_first += t1 - t0; t_second += t2 - t1; t_row += t3 - t2; t_vec += t4 - t3; t_ql += t6 - t5;
    }
    std::printf("K %d m %d k1 %d ncols %d | first (hidden) %.1f us | second %.1f + last row %.1f + %d columns %.1f = critical %.1f us | plain QL %.1f us\n",
                K, m, k1, ncols, 1e6 * t_first / REP, 1e6 * t_second / REP, 1e6 * t_row / REP, ncols, 1e6 * t_vec / REP,
                1e6 * (t_second + t_row + t_vec) / REP, 1e6 * t_ql / REP);
    return 0;
}
