/*
 * CPU oracle (test infrastructure only, parity unpinned -- see orc.h): disparity -> 3-D points, the
 * step right after the disparity routines in the reference.
 *
 *   orc_reproject_q            cameraCalibration/methods/methods.cpp:2039-2076: [X Y Z W]^T = Q * [row col d 1]^T
 *                              (the reference feeds the ROW index first), point = (X/W, Y/W, Z/W) / 100, zero
 *                              where the disparity is 0.
 *   orc_reproject_triangulate  methods.cpp:2300-2352: depth from the baseline and the focal lengths with the
 *                              row-dependent coefficient f0*f1/sqrt((v-v0)^2 + f1^2); points farther than 20000
 *                              or whose colour is black stay at the origin.
 * All arithmetic in double, stored as float (pcl::PointXYZRGB holds floats).
 */
#include "orc.h"
#include <math.h>
#include <string.h>

void orc_reproject_q(const float* disp, int H, int W, const double Q[16], float* xyz)
{
#pragma omp parallel for
    for (int u = 0; u < H; u++)
        for (int v = 0; v < W; v++) {
            const double d = (double)disp[(long)u * W + v];
            double X = 0, Y = 0, Z = 0;
            if (d != 0) {
                const double in[4] = {(double)u, (double)v, d, 1.0};
                double o[4];
                for (int i = 0; i < 4; i++) {
                    double s = 0;
                    for (int k = 0; k < 4; k++) s += Q[i * 4 + k] * in[k];
                    o[i] = s;
                }
                X = o[0] / o[3]; Y = o[1] / o[3]; Z = o[2] / o[3];
            }
            float* p = xyz + ((long)u * W + v) * 3;
            p[0] = (float)(X / 100); p[1] = (float)(Y / 100); p[2] = (float)(Z / 100);
        }
}

void orc_reproject_triangulate(const float* disp, int H, int W, const double K[9], double baseline, const float* bgr, float* xyz)
{
    const double f0 = K[0], f1 = K[4], u0 = K[2], v0 = K[5];
    memset(xyz, 0, (size_t)H * W * 3 * sizeof(float));
#pragma omp parallel for
    for (int v = 0; v < H; v++) {
        const double temp1 = (v - v0) * (v - v0) + f1 * f1;
        const double coeff = f0 * f1 / sqrt(temp1);
        for (int u = 0; u < W; u++) {
            const double px0 = (u - u0) / f0, px1 = (v - v0) / f1;
            const double d = (double)disp[(long)v * W + u];
            double X = 0, Y = 0, Z = 0;
            if (d != 0) {
                const double cur = fabs((double)disp[(long)v * W + u]);
                Z = baseline * coeff / cur;
                Y = px1 * Z;
                X = px0 * sqrt(Z * Z + Y * Y);
            }
            if (Z > 20000) continue;
            if (bgr) {
                const float* c = bgr + ((long)v * W + u) * 3;
                if (c[0] == 0 && c[1] == 0 && c[2] == 0) continue;
            }
            float* p = xyz + ((long)v * W + u) * 3;
            p[0] = (float)X; p[1] = (float)Y; p[2] = (float)Z;
        }
    }
}

#ifdef _OPENMP
#include <omp.h>
#endif
void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
