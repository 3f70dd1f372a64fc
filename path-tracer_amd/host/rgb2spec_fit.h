// rgb2spec_fit.h — the Jakob-Hanika spectral upsampling fit, shared by the scene loader (pthost.cpp: colours outside any
// table) and by tools/rgb2spec_opt.cpp, which regenerates the coefficient table the reference embeds
// (pt/src/spectrum/rgb2spec.rs:1-6: `srgb.coeffs`, missing from its tree, .MISSING_LARGE_BLOBS).
#pragma once
#include <algorithm>
#include <cmath>

#ifndef PT_CIE_N
#include "../csrc/cie_data.inc"
#endif

namespace rgb2spec {

static const float kD65[PT_CIE_N] = {PT_CIE_D65_VALUES};
static const double kXyz[PT_CIE_N][3] = {PT_CIE_XYZ_VALUES};

// Jakob & Hanika 2019, "A Low-Dimensional Function Space for Efficient Spectral Upsampling": find c0,c1,c2 of
// s(l) = S(c0 l^2 + c1 l + c2), S(x) = 1/2 + x / (2 sqrt(1+x^2)), whose colour under D65 matches `rgb` (sRGB
// primaries) by Gauss-Newton on the CIELAB residual. The optimisation runs on l normalised to [0,1] and the
// result is re-expressed for l in nm, the form rgb2spec.rs:8-54 reads out of srgb.coeffs.
struct Rgb2SpecFitter {
    double rgb_tbl[3][PT_CIE_N];
    double white[3];
    static constexpr double kXyzToRgb[3][3] = {{3.240479, -1.537150, -0.498535}, {-0.969256, 1.875991, 0.041556}, {0.055648, -0.204043, 1.057311}};
    static constexpr double kRgbToXyz[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
    Rgb2SpecFitter() {
        white[0] = white[1] = white[2] = 0.0;
        for (int i = 0; i < PT_CIE_N; i++) {
            const double w = (i == 0 || i == PT_CIE_N - 1) ? 0.5 : 1.0;  // trapezoid, 1 nm
            const double I = (double)kD65[i];
            for (int k = 0; k < 3; k++) {
                rgb_tbl[k][i] = 0.0;
                for (int j = 0; j < 3; j++) rgb_tbl[k][i] += kXyzToRgb[k][j] * kXyz[i][j] * I * w;
                white[k] += kXyz[i][k] * I * w;
            }
        }
    }
    void lab(double p[3]) const {
        double xyz[3] = {0, 0, 0};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) xyz[i] += p[j] * kRgbToXyz[i][j];
        auto f = [](double t) {
            const double d = 6.0 / 29.0;
            return t > d * d * d ? cbrt(t) : t / (3.0 * d * d) + 4.0 / 29.0;
        };
        const double fx = f(xyz[0] / white[0]), fy = f(xyz[1] / white[1]), fz = f(xyz[2] / white[2]);
        p[0] = 116.0 * fy - 16.0;
        p[1] = 500.0 * (fx - fy);
        p[2] = 200.0 * (fy - fz);
    }
    void residual(const double c[3], const double rgb[3], double r[3]) const {
        double out[3] = {0, 0, 0};
        for (int i = 0; i < PT_CIE_N; i++) {
            const double l = (double)i / (double)(PT_CIE_N - 1);
            const double x = (c[0] * l + c[1]) * l + c[2];
            const double s = 0.5 * x / sqrt(x * x + 1.0) + 0.5;
            for (int j = 0; j < 3; j++) out[j] += rgb_tbl[j][i] * s;
        }
        lab(out);
        double t[3] = {rgb[0], rgb[1], rgb[2]};
        lab(t);
        for (int j = 0; j < 3; j++) r[j] = t[j] - out[j];
    }
    bool gauss_newton(const double rgb[3], double c[3]) const {
        double r = 0.0;
        for (int it = 0; it < 30; it++) {
            double res[3], J[3][3];
            residual(c, rgb, res);
            for (int i = 0; i < 3; i++) {
                const double eps = 1e-4;
                double tmp[3] = {c[0], c[1], c[2]}, r0[3], r1[3];
                tmp[i] -= eps;
                residual(tmp, rgb, r0);
                tmp[i] = c[i] + eps;
                residual(tmp, rgb, r1);
                for (int j = 0; j < 3; j++) J[j][i] = (r1[j] - r0[j]) / (2.0 * eps);
            }
            // solve J x = res (Cramer)
            const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                               J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
            if (det == 0.0 || !std::isfinite(det)) return false;
            double x[3];
            for (int k = 0; k < 3; k++) {
                double M[3][3];
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < 3; b++) M[a][b] = (b == k) ? res[a] : J[a][b];
                x[k] = (M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                        M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0])) /
                       det;
            }
            r = 0.0;
            for (int j = 0; j < 3; j++) {
                c[j] -= x[j];
                r += res[j] * res[j];
            }
            const double mx = std::max(std::max(fabs(c[0]), fabs(c[1])), fabs(c[2]));
            if (mx > 200.0)
                for (int j = 0; j < 3; j++) c[j] *= 200.0 / mx;
            if (r < 1e-6) break;
        }
        return r < 1e-2;
    }
    // rgb: components in [0,1] with max == 0.5 in practice (TristimulusSpectrum::new normalises by 2*max).
    bool fit(const float rgb_in[3], float out[3]) const {
        const double grey = std::max(std::max(rgb_in[0], rgb_in[1]), rgb_in[2]);
        double c[3] = {0.0, 0.0, 0.0};
        bool ok = true;
        const int steps = 16;  // continuation from the grey of equal maximum to the target colour
        for (int s = 1; s <= steps; s++) {
            const double t = (double)s / steps;
            const double rgb[3] = {grey + t * (rgb_in[0] - grey), grey + t * (rgb_in[1] - grey), grey + t * (rgb_in[2] - grey)};
            ok = gauss_newton(rgb, c);
        }
        const double c0 = 360.0, c1 = 1.0 / (830.0 - 360.0);
        const double A = c[0], B = c[1], C = c[2];
        out[0] = (float)(A * (c1 * c1));
        out[1] = (float)(B * c1 - 2.0 * A * c0 * (c1 * c1));
        out[2] = (float)(C - B * c0 * c1 + A * (c0 * c1) * (c0 * c1));
        return ok;
    }
};

}  // namespace rgb2spec
