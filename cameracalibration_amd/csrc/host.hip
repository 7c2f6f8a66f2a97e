// Host-side pieces of libmdisp.so that run no GPU work: rectification-map construction (double
// arithmetic, done once per calibration) and the synthetic rectified-pair generator used by the
// tests and the benchmark.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <vector>

static void mul33(const double* A, const double* B, double* C)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}

static bool inv33(const double* m, double* o)
{
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

static inline int round_sat_i32(double v)
{
    if (v > 2147483647.0) v = 2147483647.0;
    if (v < -2147483648.0) v = -2147483648.0;
    return (int)lrint(v);   // round half to even, as cvRound
}

// cv::initUndistortRectifyMap(K, D, R, P, size, CV_16SC2, map1, map2) as called at
// cameraCalibration/methods/methods.cpp:1705-1706: per destination pixel back-project through
// inv(P[:3,:3] * R) (accumulated incrementally along the row), apply the
// k1,k2,p1,p2,k3[,k4,k5,k6] model, and store 5-bit fixed-point source coordinates.
extern "C" int mdisp_build_rectify_maps(const double K[9], const double* dist, int ndist, const double R[9],
                                        const double P[12], int width, int height, int16_t* map1, uint16_t* map2)
{
    MD_REQUIRE(K && R && P && map1 && map2 && width > 0 && height > 0, MDISP_E_BADARG, "null argument or empty size");
    MD_REQUIRE(ndist == 0 || dist, MDISP_E_BADARG, "null distortion vector");
    MD_REQUIRE(ndist == 0 || ndist == 4 || ndist == 5 || ndist == 8, MDISP_E_BADARG, "distortion vector must have 0, 4, 5 or 8 entries");
    const double Ar[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    double ArR[9], ir[9];
    mul33(Ar, R, ArR);
    MD_REQUIRE(inv33(ArR, ir), MDISP_E_BADARG, "P[:3,:3]*R is singular");
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ndist; i++) k[i] = dist[i];
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    for (int i = 0; i < height; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < width; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double w = 1. / _w, x = _x * w, y = _y * w;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
            const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
            const double u = fx * xd + u0, v = fy * yd + v0;
            const int iu = round_sat_i32(u * 32), iv = round_sat_i32(v * 32);
            map1[((size_t)i * width + j) * 2] = (int16_t)(iu >> 5);
            map1[((size_t)i * width + j) * 2 + 1] = (int16_t)(iv >> 5);
            map2[(size_t)i * width + j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
    return MDISP_OK;
}

static inline uint32_t xorshift32(uint32_t& s)
{
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return s;
}

static inline int reflect_i(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - p - 1;
    return p;
}

// Synthetic rectified pair: right = xorshift32(seed) white noise low-passed by a 5x5 integer box
// (texture with local structure); ground-truth disparity = piecewise-planar sawtooth
// d(x,y) = clamp(round(D/4 + D/2 * frac(x/(4D) + y/(3D))), 0, D-1) (disparity gradient 1/8 px per px
// inside a plane, a D/2 jump between planes); left(x,y) = right(reflect(x - d), y)
// plus i.i.d. noise in [-2, 2] from xorshift32(seed + 1).
extern "C" int mdisp_synth_pair(uint32_t seed, int rows, int cols, int numDisp, uint8_t* left, uint8_t* right, uint16_t* gt)
{
    MD_REQUIRE(left && right && rows > 0 && cols > 0 && numDisp > 0, MDISP_E_BADARG, "bad arguments");
    const size_t N = (size_t)rows * cols;
    std::vector<uint8_t> noise(N);
    uint32_t s = seed ? seed : 0x5EED0001u;
    for (size_t i = 0; i < N; i++) noise[i] = (uint8_t)(xorshift32(s) >> 24);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            int acc = 0;
            for (int dy = -2; dy <= 2; dy++)
                for (int dx = -2; dx <= 2; dx++)
                    acc += noise[(size_t)reflect_i(y + dy, rows) * cols + reflect_i(x + dx, cols)];
            right[(size_t)y * cols + x] = (uint8_t)(acc / 25);
        }
    // stretch the low-passed texture back to the full 8-bit range
    int mn = 255, mx = 0;
    for (size_t i = 0; i < N; i++) { if (right[i] < mn) mn = right[i]; if (right[i] > mx) mx = right[i]; }
    if (mx > mn)
        for (size_t i = 0; i < N; i++) right[i] = (uint8_t)(((int)right[i] - mn) * 255 / (mx - mn));
    uint32_t s2 = (seed ? seed : 0x5EED0001u) + 1;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            double t = x / (4.0 * numDisp) + y / (3.0 * numDisp);
            t -= floor(t);
            int d = (int)lrint(numDisp / 4.0 + (numDisp / 2.0) * t);
            if (d < 0) d = 0;
            if (d > numDisp - 1) d = numDisp - 1;
            if (gt) gt[(size_t)y * cols + x] = (uint16_t)d;
            int v = right[(size_t)y * cols + reflect_i(x - d, cols)];
            v += (int)(xorshift32(s2) % 5u) - 2;
            left[(size_t)y * cols + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    return MDISP_OK;
}
