// Host-side (fp64) construction of the one-time SHT tables.
//
// Replaces RealSHT.__init__ / InverseRealSHT.__init__ (fme/sht_fix.py:69-111,
// 154-192) and the torch-harmonics 0.8.0 entry points they call
// (quadrature.{legendre_gauss,lobatto,clenshaw_curtiss}_weights,
// legendre._precompute_legpoly).  Everything is computed in fp64 and rounded to
// fp32 once, as the reference does (`.float()` at sht_fix.py:111,192).
#pragma once
#include <string>
#include <vector>

namespace ace {

enum Grid { GRID_LEGENDRE_GAUSS = 0, GRID_LOBATTO = 1, GRID_EQUIANGULAR = 2 };

// returns false for an unknown grid name
bool parse_grid(const std::string& name, Grid* out);

// nodes (ascending in cos(theta)) and weights on [-1, 1]
void quadrature(Grid g, int n, std::vector<double>& x, std::vector<double>& w);

struct ShtTables {
    int nlat = 0, nlon = 0, lmax = 0, mmax = 0;
    int Hp = 0;   // nlat rounded up to 32 (row pitch of wt, zero padded)
    int Lp = 0;   // lmax rounded up to 32 (row pitch of pt, zero padded)
    int Kf = 0;   // nlon/2 + 1: number of folded longitudes
    int Kfp = 0;  // Kf rounded up to 4 (row pitch of fc/fs/gc/gs)
    // forward Legendre x quadrature weights: wt[m][l][k] (pitch Hp), zero for l < m
    std::vector<float> wt;
    // inverse Legendre, transposed: pt[m][k][l] (pitch Lp), zero for l < m
    std::vector<float> pt;
    // forward folded DFT: fc[m][w] = (2pi/W) cos(2pi m w/W), fs[m][w] = -(2pi/W) sin(...)  (pitch Kfp)
    std::vector<float> fc, fs;
    // inverse folded DFT: gc[m][w] = g_m cos(2pi m w/W), gs[m][w] = -g_m sin(...)  (pitch Kfp)
    std::vector<float> gc, gs;
};

// returns empty string on success, else an error message
std::string build_sht_tables(int nlat, int nlon, int lmax, int mmax, Grid grid, ShtTables& t);

}  // namespace ace
