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

int md_rectify_map_setup(const double K[9], const double* dist, int ndist, const double R[9], const double P[12], double ir[9], double k[8])
{
    MD_REQUIRE(K && R && P, MDISP_E_BADARG, "null argument");
    MD_REQUIRE(ndist == 0 || dist, MDISP_E_BADARG, "null distortion vector");
    MD_REQUIRE(ndist == 0 || ndist == 4 || ndist == 5 || ndist == 8, MDISP_E_BADARG, "distortion vector must have 0, 4, 5 or 8 entries");
    const double Ar[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    double ArR[9];
    mul33(Ar, R, ArR);
    MD_REQUIRE(inv33(ArR, ir), MDISP_E_BADARG, "P[:3,:3]*R is singular");
    for (int i = 0; i < 8; i++) k[i] = i < ndist ? dist[i] : 0.0;
    return MDISP_OK;
}

// cv::initUndistortRectifyMap(K, D, R, P, size, CV_16SC2, map1, map2) as called at
// cameraCalibration/methods/methods.cpp:1705-1706: per destination pixel back-project through
// inv(P[:3,:3] * R) (accumulated incrementally along the row), apply the
// k1,k2,p1,p2,k3[,k4,k5,k6] model, and store 5-bit fixed-point source coordinates.
extern "C" int mdisp_build_rectify_maps(const double K[9], const double* dist, int ndist, const double R[9],
                                        const double P[12], int width, int height, int16_t* map1, uint16_t* map2)
{
    MD_REQUIRE(map1 && map2 && width > 0 && height > 0, MDISP_E_BADARG, "null argument or empty size");
    double ir[9], k[8];
    MD_TRY(md_rectify_map_setup(K, dist, ndist, R, P, ir, k));
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

// ---------------------------------------------------------------------------------------------
// cv::stereoRectify(K1, D1, K2, D2, size, R, T, R1, R2, P1, P2, Q, flags, alpha, size, roi1, roi2) as
// called at cameraCalibration/methods/methods.cpp:1691-1694 (Bouguet's method).
namespace {
struct V3 { double x, y, z; };
struct M3 {
    double m[9];
    V3 operator*(const V3& v) const
    { return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z}; }
    M3 operator*(const M3& b) const { M3 c; mul33(m, b.m, c.m); return c; }
    M3 t() const { return {{m[0], m[3], m[6], m[1], m[4], m[7], m[2], m[5], m[8]}}; }
};
inline double norm(const V3& v) { return sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
inline V3 cross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// rotation vector <-> matrix (Rodrigues)
M3 rot_from_vec(const V3& r)
{
    const double th = norm(r);
    if (th < 2.220446049250313e-16) return {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    const double c = cos(th), s = sin(th), c1 = 1. - c, it = 1. / th;
    const double x = r.x * it, y = r.y * it, z = r.z * it;
    return {{c + c1 * x * x, c1 * x * y - s * z, c1 * x * z + s * y,
             c1 * x * y + s * z, c + c1 * y * y, c1 * y * z - s * x,
             c1 * x * z - s * y, c1 * y * z + s * x, c + c1 * z * z}};
}
V3 vec_from_rot(const M3& R)
{
    const double* a = R.m;
    V3 r = {a[7] - a[5], a[2] - a[6], a[3] - a[1]};
    const double s = sqrt((r.x * r.x + r.y * r.y + r.z * r.z) * 0.25);
    double c = (a[0] + a[4] + a[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    const double th = acos(c);
    if (s < 1e-5) {
        if (c > 0) return {0, 0, 0};
        double t0 = (a[0] + 1) * 0.5, t1 = (a[4] + 1) * 0.5, t2 = (a[8] + 1) * 0.5;
        V3 v = {sqrt(t0 > 0 ? t0 : 0), sqrt(t1 > 0 ? t1 : 0) * (a[1] < 0 ? -1. : 1.), sqrt(t2 > 0 ? t2 : 0) * (a[2] < 0 ? -1. : 1.)};
        if (fabs(v.x) < fabs(v.y) && fabs(v.x) < fabs(v.z) && (a[5] > 0) != (v.y * v.z > 0)) v.z = -v.z;
        const double n = th / norm(v);
        return {v.x * n, v.y * n, v.z * n};
    }
    const double k = 1 / (2 * s) * th;
    return {r.x * k, r.y * k, r.z * k};
}

struct Camera { const double* K; double k[8]; };

// pixel -> undistorted normalised coordinates (5 fixed-point iterations), then an optional 3x3 map
void undistort_px(const Camera& cam, double u, double v, const M3* M, double& ox, double& oy)
{
    const double x0 = (u - cam.K[2]) / cam.K[0], y0 = (v - cam.K[5]) / cam.K[4];
    const double* k = cam.k;
    double x = x0, y = y0;
    for (int it = 0; it < 5; it++) {
        const double r2 = x * x + y * y;
        const double icd = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        const double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x), dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - dx) * icd; y = (y0 - dy) * icd;
    }
    if (M) {
        const V3 p = (*M) * V3{x, y, 1.0};
        const double w = 1. / p.z;
        x = p.x * w; y = p.y * w;
    }
    ox = x; oy = y;
}

struct RectF { float x, y, w, h; };
// inscribed / circumscribed rectangles of the rectified image of a 9x9 grid of source pixels
void grid_rectangles(const Camera& cam, const M3& R, const double P[12], int W, int H, RectF& inner, RectF& outer)
{
    const M3 Pm = {{P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}};
    const M3 PR = Pm * R;
    const float BIG = 3.402823466e38f;
    float ix0 = -BIG, ix1 = BIG, iy0 = -BIG, iy1 = BIG, ox0 = BIG, ox1 = -BIG, oy0 = BIG, oy1 = -BIG;
    for (int gy = 0; gy < 9; gy++)
        for (int gx = 0; gx < 9; gx++) {
            double ux, uy;
            undistort_px(cam, (float)gx * W / 8, (float)gy * H / 8, &PR, ux, uy);
            const float px = (float)ux, py = (float)uy;
            ox0 = fminf(ox0, px); ox1 = fmaxf(ox1, px); oy0 = fminf(oy0, py); oy1 = fmaxf(oy1, py);
            if (gx == 0) ix0 = fmaxf(ix0, px);
            if (gx == 8) ix1 = fminf(ix1, px);
            if (gy == 0) iy0 = fmaxf(iy0, py);
            if (gy == 8) iy1 = fminf(iy1, py);
        }
    inner = {ix0, iy0, ix1 - ix0, iy1 - iy0};
    outer = {ox0, oy0, ox1 - ox0, oy1 - oy0};
}

void valid_roi(const RectF& in, double c0x, double c0y, double cx, double cy, double s, int W, int H, int roi[4])
{
    const int x = (int)ceil((in.x - c0x) * s + cx), y = (int)ceil((in.y - c0y) * s + cy);
    const int w = (int)floor(in.w * s), h = (int)floor(in.h * s);
    const int xa = x < 0 ? 0 : x, ya = y < 0 ? 0 : y, xb = x + w > W ? W : x + w, yb = y + h > H ? H : y + h;
    roi[0] = xa; roi[1] = ya; roi[2] = xb > xa ? xb - xa : 0; roi[3] = yb > ya ? yb - ya : 0;
}
}  // namespace

extern "C" int mdisp_stereo_rectify(const double K1[9], const double* D1, const double K2[9], const double* D2, int ndist,
                                    int width, int height, const double R[9], const double T[3], int zero_disparity,
                                    double alpha, double R1[9], double R2[9], double P1[12], double P2[12], double Q[16],
                                    int roi1[4], int roi2[4])
{
    MD_REQUIRE(K1 && K2 && R && T && R1 && R2 && P1 && P2 && width > 0 && height > 0, MDISP_E_BADARG, "null argument or empty size");
    MD_REQUIRE(ndist == 0 || (D1 && D2), MDISP_E_BADARG, "null distortion vector");
    MD_REQUIRE(ndist == 0 || ndist == 4 || ndist == 5 || ndist == 8, MDISP_E_BADARG, "distortion vectors must have 0, 4, 5 or 8 entries");
    Camera cam[2] = {{K1, {0}}, {K2, {0}}};
    for (int i = 0; i < ndist; i++) { cam[0].k[i] = D1[i]; cam[1].k[i] = D2[i]; }
    M3 Rm; for (int i = 0; i < 9; i++) Rm.m[i] = R[i];
    const V3 Tv = {T[0], T[1], T[2]};

    // split the relative rotation evenly between the cameras
    V3 om = vec_from_rot(Rm);
    om = {om.x * -0.5, om.y * -0.5, om.z * -0.5};
    const M3 half = rot_from_vec(om);
    V3 t = half * Tv;
    const double tn = norm(t);
    MD_REQUIRE(tn > 0.0, MDISP_E_BADARG, "zero baseline");
    const int idx = fabs(t.x) > fabs(t.y) ? 0 : 1;          // horizontal or vertical rig
    const double sgn = (idx == 0 ? t.x : t.y) > 0 ? 1. : -1.;
    // common orientation: baseline along x (or y), the other image axis perpendicular to the
    // averaged optical axis
    const V3 eb = {t.x / tn * sgn, t.y / tn * sgn, t.z / tn * sgn};
    V3 eo = idx == 0 ? V3{-eb.y, eb.x, 0} : V3{eb.y, -eb.x, 0};
    const double no = sqrt(eo.x * eo.x + eo.y * eo.y);
    eo = {eo.x / no, eo.y / no, 0};
    const V3 ex = idx == 0 ? eb : eo, ey = idx == 0 ? eo : eb, ez = cross(ex, ey);
    const M3 G = {{ex.x, ex.y, ex.z, ey.x, ey.y, ey.z, ez.x, ez.y, ez.z}};
    const M3 r1 = G * half.t(), r2 = G * half;
    t = r2 * Tv;

    double fc = ((idx == 0 ? K1[4] : K1[0]) + (idx == 0 ? K2[4] : K2[0])) * 0.5;
    double cc[2][2];
    for (int k = 0; k < 2; k++) {
        const M3& Rk = k == 0 ? r1 : r2;
        double sx = 0, sy = 0;
        for (int i = 0; i < 4; i++) {
            double ux, uy;
            undistort_px(cam[k], (float)((i & 1) * width), (float)((i >> 1) * height), nullptr, ux, uy);
            const V3 p = Rk * V3{(double)(float)ux, (double)(float)uy, 1.0};     // corner points are held as float32
            const double z = p.z == 0 ? 1 : p.z;
            sx += (float)(p.x / z * fc);
            sy += (float)(p.y / z * fc);
        }
        cc[k][0] = width / 2 - sx / 4;
        cc[k][1] = height / 2 - sy / 4;
    }
    if (zero_disparity || idx == 1) cc[0][0] = cc[1][0] = (cc[0][0] + cc[1][0]) * 0.5;
    if (zero_disparity || idx == 0) cc[0][1] = cc[1][1] = (cc[0][1] + cc[1][1]) * 0.5;

    for (int i = 0; i < 12; i++) P1[i] = P2[i] = 0;
    P1[0] = P1[5] = P2[0] = P2[5] = fc;
    P1[2] = cc[0][0]; P1[6] = cc[0][1]; P2[2] = cc[1][0]; P2[6] = cc[1][1];
    P1[10] = P2[10] = 1;
    const double tb = idx == 0 ? t.x : t.y;
    P2[idx * 4 + 3] = tb * fc;

    RectF in1, out1, in2, out2;
    grid_rectangles(cam[0], r1, P1, width, height, in1, out1);
    grid_rectangles(cam[1], r2, P2, width, height, in2, out2);
    double s = 1.;
    if (alpha > 1.) alpha = 1.;
    if (alpha >= 0) {
        auto fit = [&](const RectF& r, const double c[2], bool inner) {
            const double a = c[0] / (c[0] - r.x), b = c[1] / (c[1] - r.y);
            const double e = (width - c[0]) / (r.x + r.w - c[0]), f = (height - c[1]) / (r.y + r.h - c[1]);
            return inner ? fmax(fmax(fmax(a, b), e), f) : fmin(fmin(fmin(a, b), e), f);
        };
        const double s0 = fmax(fit(in1, cc[0], true), fit(in2, cc[1], true));
        const double s1 = fmin(fit(out1, cc[0], false), fit(out2, cc[1], false));
        s = s0 * (1 - alpha) + s1 * alpha;
    }
    fc *= s;
    P1[0] = P1[5] = P2[0] = P2[5] = fc;
    P2[idx * 4 + 3] *= s;
    if (roi1) valid_roi(in1, cc[0][0], cc[0][1], cc[0][0], cc[0][1], s, width, height, roi1);
    if (roi2) valid_roi(in2, cc[1][0], cc[1][1], cc[1][0], cc[1][1], s, width, height, roi2);
    for (int i = 0; i < 9; i++) { R1[i] = r1.m[i]; R2[i] = r2.m[i]; }
    if (Q) {
        for (int i = 0; i < 16; i++) Q[i] = 0;
        Q[0] = Q[5] = 1; Q[3] = -cc[0][0]; Q[7] = -cc[0][1]; Q[11] = fc;
        Q[14] = -1. / tb;
        Q[15] = (idx == 0 ? cc[0][0] - cc[1][0] : cc[0][1] - cc[1][1]) / tb;
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
