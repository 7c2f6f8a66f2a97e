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

/* ------------------------------------------------------------------------------------------
 * cv::stereoRectify (Bouguet), as called at cameraCalibration/methods/methods.cpp:1691-1694
 * (CALIB_ZERO_DISPARITY, alpha = 0, newImageSize = imageSize).  OpenCV 4.1.0 calib3d is not
 * vendored; restated from its published algorithm (cvStereoRectify):
 *   1. om = -0.5 * rodrigues(R); r_r = rodrigues(om); t = r_r * T
 *   2. rotate so that t lies along the x (or y) axis (two forms, see global_rotation below)
 *   3. R1 = wR * r_r^T, R2 = wR * r_r, t = R2 * T
 *   4. fc = (K1[idx^1][idx^1] + K2[idx^1][idx^1]) / 2; principal point from the four image corners
 *      (0 and n, not n-1), undistorted (5 fixed-point iterations), stored as float32, projected with
 *      R_k and fc, stored as float32, averaged; c_k = n/2 (integer division) - mean; ZERO_DISPARITY
 *      averages both.  The corner / centre convention is the one that reproduces the stored known
 *      answer bit for bit (newer OpenCV releases use n-1).
 *   5. alpha >= 0: scale fc so that the inner (alpha=0) / outer (alpha=1) rectangle of a 9x9 grid
 *      of undistorted points fills the image; alpha < 0: no scaling.
 * PINNED: the stored known answer (unditortStereoCalib.xml: Rectify_R1/R2/P1/P2/Q for the
 * StereoCalib_* inputs of the same file) is reproduced to 2e-16 (R1, R2) and exactly (principal
 * point) with alpha < 0 (its fc equals the plain average) and global_rotation = 0.
 */
static void rodrigues_vec2mat(const double r[3], double R[9])
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < 2.220446049250313e-16) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int i = 0; i < 9; i++) R[i] = c * ((i % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[i] + s * rx[i];
}

static void rodrigues_mat2vec(const double R[9], double r[3])
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; r[0] = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; r[1] = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; r[2] = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
        if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && (R[5] > 0) != (r[1] * r[2] > 0)) r[2] = -r[2];
        double n = theta / sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        r[0] *= n; r[1] *= n; r[2] *= n;
        return;
    }
    double vth = 1 / (2 * s) * theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

/* cvUndistortPoints with criteria (COUNT, 5): pixel -> normalised, distortion removed, then RR applied */
static void undistort_point(double u, double v, const double K[9], const double k[8], const double RR[9], double* ox, double* oy)
{
    double x = (u - K[2]) / K[0], y = (v - K[5]) / K[4];   /* division: reproduces the stored known answer exactly */
    double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        double dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - dX) * icdist;
        y = (y0 - dY) * icdist;
    }
    double xx = RR[0] * x + RR[1] * y + RR[2], yy = RR[3] * x + RR[4] * y + RR[5];
    double ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
    *ox = xx * ww; *oy = yy * ww;
}

/* inner / outer rectangles of the rectified 9x9 grid (icvGetRectangles), float32 like the original */
static void get_rectangles(const double K[9], const double k[8], const double R[9], const double P[12], int W, int H,
                           float inner[4], float outer[4])
{
    const int N = 9;
    double Pm[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}, RR[9];
    mat3_mul(Pm, R, RR);
    float iX0 = -3.402823466e38f, iX1 = 3.402823466e38f, iY0 = -3.402823466e38f, iY1 = 3.402823466e38f;
    float oX0 = 3.402823466e38f, oX1 = -3.402823466e38f, oY0 = 3.402823466e38f, oY1 = -3.402823466e38f;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            float px = (float)x * W / (N - 1), py = (float)y * H / (N - 1);
            double ux, uy;
            undistort_point(px, py, K, k, RR, &ux, &uy);
            float fx = (float)ux, fy = (float)uy;
            if (fx < oX0) oX0 = fx; if (fx > oX1) oX1 = fx;
            if (fy < oY0) oY0 = fy; if (fy > oY1) oY1 = fy;
            if (x == 0 && fx > iX0) iX0 = fx;
            if (x == N - 1 && fx < iX1) iX1 = fx;
            if (y == 0 && fy > iY0) iY0 = fy;
            if (y == N - 1 && fy < iY1) iY1 = fy;
        }
    inner[0] = iX0; inner[1] = iY0; inner[2] = iX1 - iX0; inner[3] = iY1 - iY0;
    outer[0] = oX0; outer[1] = oY0; outer[2] = oX1 - oX0; outer[3] = oY1 - oY0;
}

static double max2(double a, double b) { return a > b ? a : b; }
static double min2(double a, double b) { return a < b ? a : b; }

void orc_stereo_rectify(const double K1[9], const double* D1, const double K2[9], const double* D2, int ndist,
                        int W, int H, const double R[9], const double T[3], double alpha, int zero_disparity,
                        int global_rotation, double R1[9], double R2[9], double P1[12], double P2[12], double Q[16],
                        int roi1[4], int roi2[4])
{
    double om[3], r_r[9], t[3], uu[3] = {0, 0, 0}, ww[3], wR[9];
    rodrigues_mat2vec(R, om);
    for (int i = 0; i < 3; i++) om[i] *= -0.5;
    rodrigues_vec2mat(om, r_r);
    for (int i = 0; i < 3; i++) t[i] = r_r[i * 3] * T[0] + r_r[i * 3 + 1] * T[1] + r_r[i * 3 + 2] * T[2];
    int idx = fabs(t[0]) > fabs(t[1]) ? 0 : 1;
    double c = t[idx], nt = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    uu[idx] = c > 0 ? 1 : -1;
    if (global_rotation == 1) {
        /* minimal rotation of t onto the axis (Rodrigues of t x uu), the cvStereoRectify form */
        ww[0] = t[1] * uu[2] - t[2] * uu[1]; ww[1] = t[2] * uu[0] - t[0] * uu[2]; ww[2] = t[0] * uu[1] - t[1] * uu[0];
        double nw = sqrt(ww[0] * ww[0] + ww[1] * ww[1] + ww[2] * ww[2]);
        if (nw > 0.0) { double sc = acos(fabs(c) / nt) / nw; for (int i = 0; i < 3; i++) ww[i] *= sc; }
        rodrigues_vec2mat(ww, wR);
    } else {
        /* orthonormal basis with the baseline as one axis and the second axis perpendicular to the
         * (averaged) optical axis: e_b = +-t/|t|, e_o = z x e_b (or e_b x z), third = cross product.
         * This is the form the stored known answer follows (to 2e-15); it differs from the minimal
         * rotation by a twist of theta_y*theta_z/2 about the baseline. */
        double eb[3] = {t[0] / nt * uu[idx], t[1] / nt * uu[idx], t[2] / nt * uu[idx]}, eo[3], e3[3];
        if (idx == 0) { eo[0] = -eb[1]; eo[1] = eb[0]; eo[2] = 0; }         /* z x e_b */
        else { eo[0] = eb[1]; eo[1] = -eb[0]; eo[2] = 0; }                  /* e_b x z */
        double no = sqrt(eo[0] * eo[0] + eo[1] * eo[1]);
        eo[0] /= no; eo[1] /= no;
        const double* ex = idx == 0 ? eb : eo;
        const double* ey = idx == 0 ? eo : eb;
        e3[0] = ex[1] * ey[2] - ex[2] * ey[1]; e3[1] = ex[2] * ey[0] - ex[0] * ey[2]; e3[2] = ex[0] * ey[1] - ex[1] * ey[0];
        for (int j = 0; j < 3; j++) { wR[j] = ex[j]; wR[3 + j] = ey[j]; wR[6 + j] = e3[j]; }
    }
    double r_rT[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r_rT[i * 3 + j] = r_r[j * 3 + i];
    mat3_mul(wR, r_rT, R1);
    mat3_mul(wR, r_r, R2);
    for (int i = 0; i < 3; i++) t[i] = R2[i * 3] * T[0] + R2[i * 3 + 1] * T[1] + R2[i * 3 + 2] * T[2];

    double k1v[8] = {0}, k2v[8] = {0};
    for (int i = 0; i < ndist && i < 8; i++) { k1v[i] = D1 ? D1[i] : 0; k2v[i] = D2 ? D2[i] : 0; }
    const int d = (idx ^ 1) * 4;                       /* K[idx^1][idx^1] */
    double fc_new = (K1[d] + K2[d]) * 0.5;             /* ratio = newSize/imageSize/2 = 0.5 */
    double cc[2][2];
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 2; k++) {
        const double* K = k == 0 ? K1 : K2;
        const double* kv = k == 0 ? k1v : k2v;
        /* the projection step converts the rotation matrix to a rotation vector and back */
        double rv[3], Rk[9];
        rodrigues_mat2vec(k == 0 ? R1 : R2, rv);
        rodrigues_vec2mat(rv, Rk);
        double sx = 0, sy = 0;
        for (int i = 0; i < 4; i++) {
            float px = (float)((i % 2) * W), py = (float)((i < 2 ? 0 : 1) * H);
            double ux, uy;
            undistort_point(px, py, K, kv, I3, &ux, &uy);
            float fx = (float)ux, fy = (float)uy;       /* stored as float32 points (x, y, 1) */
            double X = Rk[0] * fx + Rk[1] * fy + Rk[2], Y = Rk[3] * fx + Rk[4] * fy + Rk[5], Z = Rk[6] * fx + Rk[7] * fy + Rk[8];
            if (Z == 0) Z = 1;
            float qx = (float)(X / Z * fc_new), qy = (float)(Y / Z * fc_new);   /* projected, float32 again */
            sx += qx; sy += qy;
        }
        cc[k][0] = W / 2 - sx / 4;                      /* integer division, as written in the original */
        cc[k][1] = H / 2 - sy / 4;
    }
    if (zero_disparity) {
        cc[0][0] = cc[1][0] = (cc[0][0] + cc[1][0]) * 0.5;
        cc[0][1] = cc[1][1] = (cc[0][1] + cc[1][1]) * 0.5;
    } else if (idx == 0) cc[0][1] = cc[1][1] = (cc[0][1] + cc[1][1]) * 0.5;
    else cc[0][0] = cc[1][0] = (cc[0][0] + cc[1][0]) * 0.5;

    memset(P1, 0, 12 * sizeof(double)); memset(P2, 0, 12 * sizeof(double));
    P1[0] = P1[5] = fc_new; P1[2] = cc[0][0]; P1[6] = cc[0][1]; P1[10] = 1;
    P2[0] = P2[5] = fc_new; P2[2] = cc[1][0]; P2[6] = cc[1][1]; P2[10] = 1;
    P2[idx * 4 + 3] = t[idx] * fc_new;

    if (alpha > 1.) alpha = 1.;
    float in1[4], out1[4], in2[4], out2[4];
    get_rectangles(K1, k1v, R1, P1, W, H, in1, out1);
    get_rectangles(K2, k2v, R2, P2, W, H, in2, out2);
    double cx1_0 = cc[0][0], cy1_0 = cc[0][1], cx2_0 = cc[1][0], cy2_0 = cc[1][1];
    double cx1 = cx1_0, cy1 = cy1_0, cx2 = cx2_0, cy2 = cy2_0;   /* newImageSize == imageSize */
    double s = 1.;
    if (alpha >= 0) {
        double s0 = max2(max2(max2(cx1 / (cx1_0 - in1[0]), cy1 / (cy1_0 - in1[1])), (W - cx1) / (in1[0] + in1[2] - cx1_0)),
                         (H - cy1) / (in1[1] + in1[3] - cy1_0));
        s0 = max2(max2(max2(max2(cx2 / (cx2_0 - in2[0]), cy2 / (cy2_0 - in2[1])), (W - cx2) / (in2[0] + in2[2] - cx2_0)),
                       (H - cy2) / (in2[1] + in2[3] - cy2_0)), s0);
        double s1 = min2(min2(min2(cx1 / (cx1_0 - out1[0]), cy1 / (cy1_0 - out1[1])), (W - cx1) / (out1[0] + out1[2] - cx1_0)),
                         (H - cy1) / (out1[1] + out1[3] - cy1_0));
        s1 = min2(min2(min2(min2(cx2 / (cx2_0 - out2[0]), cy2 / (cy2_0 - out2[1])), (W - cx2) / (out2[0] + out2[2] - cx2_0)),
                       (H - cy2) / (out2[1] + out2[3] - cy2_0)), s1);
        s = s0 * (1 - alpha) + s1 * alpha;
    }
    fc_new *= s;
    P1[0] = P1[5] = fc_new; P2[0] = P2[5] = fc_new;
    P2[idx * 4 + 3] = P2[idx * 4 + 3] * s;
    if (roi1) {
        int x = (int)ceil((in1[0] - cx1_0) * s + cx1), y = (int)ceil((in1[1] - cy1_0) * s + cy1);
        int w = (int)floor(in1[2] * s), h = (int)floor(in1[3] * s);
        int x1 = x < 0 ? 0 : x, y1 = y < 0 ? 0 : y, x2 = x + w > W ? W : x + w, y2 = y + h > H ? H : y + h;
        roi1[0] = x1; roi1[1] = y1; roi1[2] = x2 > x1 ? x2 - x1 : 0; roi1[3] = y2 > y1 ? y2 - y1 : 0;
    }
    if (roi2) {
        int x = (int)ceil((in2[0] - cx2_0) * s + cx2), y = (int)ceil((in2[1] - cy2_0) * s + cy2);
        int w = (int)floor(in2[2] * s), h = (int)floor(in2[3] * s);
        int x1 = x < 0 ? 0 : x, y1 = y < 0 ? 0 : y, x2 = x + w > W ? W : x + w, y2 = y + h > H ? H : y + h;
        roi2[0] = x1; roi2[1] = y1; roi2[2] = x2 > x1 ? x2 - x1 : 0; roi2[3] = y2 > y1 ? y2 - y1 : 0;
    }
    if (Q) {
        memset(Q, 0, 16 * sizeof(double));
        Q[0] = 1; Q[3] = -cc[0][0]; Q[5] = 1; Q[7] = -cc[0][1]; Q[11] = fc_new;
        Q[14] = -1. / t[idx];
        Q[15] = (idx == 0 ? cc[0][0] - cc[1][0] : cc[0][1] - cc[1][1]) / t[idx];
    }
}
