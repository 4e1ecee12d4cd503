// Host-side (fp64) SHT table construction.  See tables.h.
#include "tables.h"

#include <algorithm>
#include <cmath>

namespace ace {

static const double PI = 3.14159265358979323846264338327950288;

bool parse_grid(const std::string& name, Grid* out) {
    if (name == "legendre-gauss") { *out = GRID_LEGENDRE_GAUSS; return true; }
    if (name == "lobatto") { *out = GRID_LOBATTO; return true; }
    if (name == "equiangular") { *out = GRID_EQUIANGULAR; return true; }
    return false;
}

// Gauss-Legendre: Newton iteration on P_n with the Tricomi initial guess.
// (reference: numpy leggauss via fme/core/disco/_quadrature.py:25-33)
static void legendre_gauss(int n, std::vector<double>& x, std::vector<double>& w) {
    x.assign(n, 0.0);
    w.assign(n, 0.0);
    for (int i = 0; i < n; ++i) {
        long double z = cosl((long double)PI * (i + 0.75L) / (n + 0.5L));  // descending
        long double pp = 1.0L;
        for (int it = 0; it < 100; ++it) {
            long double p0 = 1.0L, p1 = z;
            for (int k = 2; k <= n; ++k) {
                long double p2 = ((2 * k - 1) * z * p1 - (k - 1) * p0) / k;
                p0 = p1;
                p1 = p2;
            }
            if (n == 1) { p0 = 1.0L; p1 = z; }
            pp = n * (z * p1 - p0) / (z * z - 1.0L);
            long double dz = p1 / pp;
            z -= dz;
            if (fabsl(dz) < 1e-19L) break;
        }
        // recompute derivative at the converged root
        long double p0 = 1.0L, p1 = z;
        for (int k = 2; k <= n; ++k) {
            long double p2 = ((2 * k - 1) * z * p1 - (k - 1) * p0) / k;
            p0 = p1;
            p1 = p2;
        }
        pp = n * (z * p1 - p0) / (z * z - 1.0L);
        x[n - 1 - i] = (double)z;  // ascending
        w[n - 1 - i] = (double)(2.0L / ((1.0L - z * z) * pp * pp));
    }
}

// Gauss-Lobatto: Newton iteration from Chebyshev-Lobatto nodes
// (torch-harmonics 0.8.0 lobatto_weights; pinned by sht-regression.pt)
static void lobatto(int n, std::vector<double>& x, std::vector<double>& w) {
    x.assign(n, 0.0);
    w.assign(n, 0.0);
    std::vector<double> t(n), vdm((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) t[i] = -cos(PI * i / (n - 1));
    for (int it = 0; it < 100; ++it) {
        double maxd = 0.0;
        for (int i = 0; i < n; ++i) {
            double* v = &vdm[(size_t)i * n];
            v[0] = 1.0;
            v[1] = t[i];
            for (int k = 2; k < n; ++k) v[k] = ((2 * k - 1) * t[i] * v[k - 1] - (k - 1) * v[k - 2]) / k;
            double tn = t[i] - (t[i] * v[n - 1] - v[n - 2]) / (n * v[n - 1]);
            maxd = std::max(maxd, fabs(tn - t[i]));
            t[i] = tn;
        }
        if (maxd < 1e-16) break;
    }
    for (int i = 0; i < n; ++i) {
        // weights use the Vandermonde of the last iteration's input nodes, as the published routine does;
        // at convergence the difference is below fp64 resolution
        double* v = &vdm[(size_t)i * n];
        x[i] = t[i];
        w[i] = 2.0 / ((double)(n * (n - 1)) * v[n - 1] * v[n - 1]);
    }
}

// Clenshaw-Curtis (closed form of fme/core/disco/_quadrature.py:36-71)
static void clenshaw_curtiss(int n, std::vector<double>& x, std::vector<double>& w) {
    x.assign(n, 0.0);
    w.assign(n, 0.0);
    int n1 = n - 1;
    for (int k = 0; k < n; ++k) {
        double theta = PI - PI * k / n1;  // linspace(pi, 0, n): nodes ascending
        x[k] = cos(theta);
        double s = 0.0;
        for (int j = 1; j <= n1 / 2; ++j) {
            double b = (2 * j == n1) ? 1.0 : 2.0;
            s += b / (4.0 * j * j - 1.0) * cos(2.0 * j * theta);
        }
        double c = (k == 0 || k == n1) ? 1.0 : 2.0;
        w[k] = c / n1 * (1.0 - s);
    }
}

void quadrature(Grid g, int n, std::vector<double>& x, std::vector<double>& w) {
    switch (g) {
        case GRID_LEGENDRE_GAUSS: legendre_gauss(n, x, w); break;
        case GRID_LOBATTO: lobatto(n, x, w); break;
        case GRID_EQUIANGULAR: clenshaw_curtiss(n, x, w); break;
    }
}

static inline int round4(int v) { return (v + 3) & ~3; }

std::string build_sht_tables(int nlat, int nlon, int lmax, int mmax, Grid grid, ShtTables& t) {
    if (nlat < 2 || nlon < 2) return "nlat and nlon must be >= 2";
    if (lmax <= 0) lmax = (grid == GRID_LOBATTO) ? nlat - 1 : nlat;  // fme/sht_fix.py:87-96
    if (mmax <= 0) mmax = nlon / 2 + 1;                               // fme/sht_fix.py:104
    if (mmax > nlon / 2 + 1) return "mmax > nlon/2+1 is not supported";
    t.nlat = nlat; t.nlon = nlon; t.lmax = lmax; t.mmax = mmax;
    t.Hp = (nlat + 31) & ~31;   // zero-padded to the deepest GEMM stage (direct-to-LDS engine reads whole stages)
    t.Lp = (lmax + 31) & ~31;
    t.Kf = nlon / 2 + 1;
    t.Kfp = round4(t.Kf);

    std::vector<double> cost, w;
    quadrature(grid, nlat, cost, w);
    // colatitudes ascending: theta = flip(arccos(cost)); x_k = cos(theta_k)  (fme/sht_fix.py:103)
    std::vector<double> xk(nlat);
    for (int k = 0; k < nlat; ++k) xk[k] = cos(acos(cost[nlat - 1 - k]));

    // orthonormal associated Legendre recursion with Condon-Shortley phase
    // (torch-harmonics _precompute_legpoly).  vdm[m][l] for one node at a time.
    const int nmax = std::max(mmax, lmax);
    t.wt.assign((size_t)mmax * lmax * t.Hp, 0.f);
    t.pt.assign((size_t)mmax * nlat * t.Lp, 0.f);
    std::vector<double> vdm((size_t)nmax * nmax);
    for (int k = 0; k < nlat; ++k) {
        const double x = xk[k];
        std::fill(vdm.begin(), vdm.end(), 0.0);
        auto V = [&](int m, int l) -> double& { return vdm[(size_t)m * nmax + l]; };
        V(0, 0) = 1.0 / sqrt(4.0 * PI);
        for (int l = 1; l < nmax; ++l) {
            V(l - 1, l) = sqrt(2.0 * l + 1.0) * x * V(l - 1, l - 1);
            V(l, l) = sqrt((2.0 * l + 1.0) * (1.0 + x) * (1.0 - x) / 2.0 / l) * V(l - 1, l - 1);
        }
        for (int l = 2; l < nmax; ++l) {
            for (int m = 0; m < l - 1; ++m) {
                double f1 = sqrt((2.0 * l - 1.0) / (l - m) * (2.0 * l + 1.0) / (l + m));
                double f2 = sqrt((double)(l + m - 1) / (l - m) * (2.0 * l + 1.0) / (2.0 * l - 3.0) * (l - m - 1) /
                                 (l + m));
                V(m, l) = x * f1 * V(m, l - 1) - f2 * V(m, l - 2);
            }
        }
        for (int m = 0; m < mmax; ++m) {
            const double cs = (m & 1) ? -1.0 : 1.0;
            for (int l = 0; l < lmax; ++l) {
                double p = cs * V(m, l);
                t.wt[((size_t)m * lmax + l) * t.Hp + k] = (float)(p * w[k]);  // weights unflipped (symmetric)
                t.pt[((size_t)m * nlat + k) * t.Lp + l] = (float)p;
            }
        }
    }

    // folded real-DFT matrices
    const int W = nlon;
    t.fc.assign((size_t)mmax * t.Kfp, 0.f);
    t.fs.assign((size_t)mmax * t.Kfp, 0.f);
    t.gc.assign((size_t)mmax * t.Kfp, 0.f);
    t.gs.assign((size_t)mmax * t.Kfp, 0.f);
    const double s = 2.0 * PI / W;
    for (int m = 0; m < mmax; ++m) {
        const bool nyq = (W % 2 == 0) && (2 * m == W);
        const double g = (m == 0 || nyq) ? 1.0 : 2.0;
        for (int wv = 0; wv < t.Kf; ++wv) {
            long j = ((long)m * wv) % W;
            double c, sn;
            // exact values on the axes so that m=0 / Nyquist imaginary parts vanish identically
            if (j == 0) { c = 1.0; sn = 0.0; }
            else if (2 * j == W) { c = -1.0; sn = 0.0; }
            else if (4 * j == W) { c = 0.0; sn = 1.0; }
            else if (4 * j == 3 * (long)W) { c = 0.0; sn = -1.0; }
            else { c = cos(2.0 * PI * j / W); sn = sin(2.0 * PI * j / W); }
            t.fc[(size_t)m * t.Kfp + wv] = (float)(s * c);
            t.fs[(size_t)m * t.Kfp + wv] = (float)(-s * sn);
            t.gc[(size_t)m * t.Kfp + wv] = (float)(g * c);
            t.gs[(size_t)m * t.Kfp + wv] = (float)(-g * sn);
        }
    }
    return "";
}

}  // namespace ace
