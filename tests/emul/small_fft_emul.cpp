// Host check of ace_amd/csrc/small_fft.h (the compile-time FFTs behind the longitude transform): complex forward / inverse and
// real-input half spectra against direct double-precision sums, for every level length fft.hip instantiates.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct V2 { float x, y; };
static inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
static inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
static inline V2 operator*(V2 a, V2 b) { return {a.x * b.x, a.y * b.y}; }

#include "../../ace_amd/csrc/small_fft.h"

using namespace ace::sfft;
static double worst = 0.0;

template <int N>
static void check() {
    std::vector<V2> x(N), y(N);
    std::vector<float> r(N);
    for (int j = 0; j < N; ++j) { x[j] = {(float)std::sin(1.3 * j + 0.2) * 2.f, (float)std::cos(0.7 * j * j + 1.0)}; r[j] = (float)std::sin(0.9 * j + 0.1 * j * j); }
    double scale = 0;
    for (int j = 0; j < N; ++j) scale += std::hypot(x[j].x, x[j].y);
    for (int inv = 0; inv < 2; ++inv) {
        if (inv) CFft<N, true, V2>::run([&](int j) { return x[j]; }, [&](int k, V2 v) { y[k] = v; });
        else CFft<N, false, V2>::run([&](int j) { return x[j]; }, [&](int k, V2 v) { y[k] = v; });
        for (int k = 0; k < N; ++k) {
            double re = 0, im = 0;
            for (int j = 0; j < N; ++j) {
                const double a = (inv ? 2.0 : -2.0) * M_PI * (double)((long)j * k % N) / N;
                re += x[j].x * std::cos(a) - x[j].y * std::sin(a);
                im += x[j].x * std::sin(a) + x[j].y * std::cos(a);
            }
            const double e = std::hypot(y[k].x - re, y[k].y - im) / scale;
            if (e > worst) worst = e;
            if (e > 2e-6) { printf("cfft<%d> inv %d k %d: got %g %g want %g %g\n", N, inv, k, y[k].x, y[k].y, re, im); exit(1); }
        }
    }
    std::vector<V2> h(N / 2 + 1);
    RFft<N, V2>::run([&](int j) { return r[j]; }, [&](int k, V2 v) { h[k] = v; });
    double rs = 0;
    for (int j = 0; j < N; ++j) rs += std::fabs(r[j]);
    for (int k = 0; k <= N / 2; ++k) {
        double re = 0, im = 0;
        for (int j = 0; j < N; ++j) {
            const double a = -2.0 * M_PI * (double)((long)j * k % N) / N;
            re += r[j] * std::cos(a);
            im += r[j] * std::sin(a);
        }
        const double e = std::hypot(h[k].x - re, h[k].y - im) / rs;
        if (e > worst) worst = e;
        if (e > 2e-6) { printf("rfft<%d> k %d: got %g %g want %g %g\n", N, k, h[k].x, h[k].y, re, im); exit(1); }
    }
}

int main() {
    check<2>(); check<3>(); check<4>(); check<5>(); check<6>(); check<8>(); check<9>(); check<10>(); check<12>(); check<16>();
    check<18>(); check<20>(); check<24>(); check<30>(); check<36>(); check<40>();
    printf("small ffts ok, worst relative error %.3g\n", worst);
    return 0;
}
