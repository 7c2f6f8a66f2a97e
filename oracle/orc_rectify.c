/*
 * CPU oracle (test infrastructure only, parity unpinned -- see orc.h): rectification maps and
 * fixed-point bilinear remap.
 *
 * The reference only calls OpenCV here (cameraCalibration/methods/methods.cpp:1691-1694
 * stereoRectify, :1705-1706 initUndistortRectifyMap(..., CV_16SC2, ...), :1738,1745
 * remap(..., INTER_LINEAR)); OpenCV 4.1.0 imgproc/calib3d are not vendored, so the arithmetic is
 * restated from its published behaviour:
 *   maps: per destination pixel, back-project through inv(P[:3,:3]*R), apply the
 *         k1,k2,p1,p2,k3[,k4,k5,k6] distortion model, iu = round(u*32), iv = round(v*32);
 *         map1 = (iu>>5, iv>>5) as int16x2, map2 = (iv&31)*32 + (iu&31) as uint16.
 *   remap: 4 taps weighted by a 32x32 table of int16 weights (1-fx)(1-fy)*32768 ... that sum to
 *         32768, result (sum + 16384) >> 15, taps outside the source read 0 (BORDER_CONSTANT).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void mat3_mul(const double* A, const double* B, double* C)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}

static void mat3_inv(const double* m, double* o)
{
    double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

void orc_init_undistort_rectify_map(const double K[9], const double* dist, int ndist, const double R[9],
                                    const double P[12], int W, int H, int16_t* map1, uint16_t* map2)
{
    double Ar[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    double ArR[9], ir[9];
    mat3_mul(Ar, R, ArR);
    mat3_inv(ArR, ir);
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ndist && i < 8; i++) k[i] = dist[i];
    double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    for (int i = 0; i < H; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            double w = 1. / _w, x = _x * w, y = _y * w;
            double x2 = x * x, y2 = y * y;
            double r2 = x2 + y2, _2xy = 2 * x * y;
            double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
            double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
            double u = fx * xd + u0, v = fy * yd + v0;
            double us = u * 32, vs = v * 32;
            if (us > 2147483647.0) us = 2147483647.0; if (us < -2147483648.0) us = -2147483648.0;
            if (vs > 2147483647.0) vs = 2147483647.0; if (vs < -2147483648.0) vs = -2147483648.0;
            int iu = (int)lrint(us), iv = (int)lrint(vs);
            map1[((long)i * W + j) * 2] = (int16_t)(iu >> 5);
            map1[((long)i * W + j) * 2 + 1] = (int16_t)(iv >> 5);
            map2[(long)i * W + j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
}

/* weight table entry for fractional position (a/32, b/32).  The products (32-a)(32-b)*32 are
 * exact integers summing to 32768; only a=b=0 overflows int16 (32768 -> 32767), where OpenCV's
 * sum correction moves the missing 1 onto the diagonal tap -- which cannot change an 8-bit result. */
static void bilinear_weights(int a, int b, int wt[4])
{
    wt[0] = (32 - a) * (32 - b) * 32;
    wt[1] = a * (32 - b) * 32;
    wt[2] = (32 - a) * b * 32;
    wt[3] = a * b * 32;
    if (wt[0] > 32767) { wt[0] = 32767; wt[3] += 1; }
}

void orc_remap_bilinear_u8(const uint8_t* src, int sH, int sW, int cn, const int16_t* map1, const uint16_t* map2,
                           int dH, int dW, uint8_t* dst)
{
#pragma omp parallel for
    for (int y = 0; y < dH; y++)
        for (int x = 0; x < dW; x++) {
            int sx = map1[((long)y * dW + x) * 2], sy = map1[((long)y * dW + x) * 2 + 1];
            int f = map2[(long)y * dW + x] & 1023;
            int wt[4];
            bilinear_weights(f & 31, f >> 5, wt);
            for (int c = 0; c < cn; c++) {
                int acc = 0;
                for (int t = 0; t < 4; t++) {
                    int xx = sx + (t & 1), yy = sy + (t >> 1);
                    int v = (xx >= 0 && xx < sW && yy >= 0 && yy < sH) ? src[((long)yy * sW + xx) * cn + c] : 0;
                    acc += v * wt[t];
                }
                int r = (acc + (1 << 14)) >> 15;
                dst[((long)y * dW + x) * cn + c] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
        }
}
