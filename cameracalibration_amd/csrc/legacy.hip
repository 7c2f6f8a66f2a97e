// The fixed-window matchers of namespace stereomatch_1 (cameraCalibration/methods/method_StereoMatching.cpp:32-869):
// computeAD / TAD / SD, computeSAD / STAD / SSD and their _inteOpti / _BFOpti variants.  Left / right CV_8UC1,
// numDisparity is a count, output CV_8U = (char)(16 * winning index).  Sums are exact integers, except the _inteOpti
// kinds, which by default go through float summed-area tables in the reference's arithmetic (md_legacy_inte, stereo.hip).
#include "common.h"

namespace {

struct LegacyArgs {
    const u8* L; size_t lp; const u8* R; size_t rp;
    int H, W, h, T, minD, numD;
    int square, clampT, foot, all;     // foot: 0 direct window, 1 integral-image footprint, 2 sqrBoxFilter
    u8* out; size_t op;
};

__device__ __forceinline__ int refl101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - p - 2;
    return p;
}

// AD / TAD / SD (:32-72, :311-351, :596-637): per pixel, columns i >= numDisparity only.  The reference reads
// the left pixel through at<char> into an 8-bit row, so values above 127 become 0; candidate k is the right
// pixel at column i - numDisparity + k.
__global__ __launch_bounds__(256) void legacy_pixel_kernel(LegacyArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= a.W) return;
    int res = 0;
    if (i >= a.numD) {
        int lv = a.L[(size_t)j * a.lp + i];
        if (lv > 127) lv = 0;
        const u8* r = a.R + (size_t)j * a.rp + i - a.numD;
        int best = 0, bc = 1 << 30;
        for (int k = 0; k < a.numD; k++) {
            int c = abs(lv - (int)r[k]);
            if (a.square) c = min(c * c, 255);
            if (c < bc) { bc = c; best = k; }
        }
        if (a.clampT) best = min(best, a.T);
        res = 16 * best;
    }
    a.out[(size_t)j * a.op + i] = (u8)res;
}

// Window variants.  A workgroup owns a 32 x 8 tile; per candidate the difference image of the tile and
// its halo goes to LDS, is summed along rows, then along columns.
//   foot 0 (:83-141 and twins): full window; a candidate whose centre column i - d lies left of h keeps cost 0
//   foot 1 (_inteOpti with exact sums, MDISP_TUNE_SAT_INT_TABLES): differences are 0 left of column d; the
//          integral-image lookup covers rows j-h .. j+h-1 and columns i-h .. i+h-1
//   foot 2 (_BFOpti): the same difference image, squared by sqrBoxFilter, full window, BORDER_REFLECT_101
#define LG_X 32
#define LG_Y 8
__global__ __launch_bounds__(256) void legacy_window_kernel(LegacyArgs a)
{
    extern __shared__ int lgs[];
    const int h = a.h, TW = LG_X + 2 * h, TH = LG_Y + 2 * h;
    int* dv = lgs;                 // [TH][TW]
    int* hs = lgs + TH * TW;       // [TH][LG_X]
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = blockIdx.x * LG_X, y0 = blockIdx.y * LG_Y;
    const int i = x0 + tx, j = y0 + ty;
    const int ext = a.foot == 1 ? 2 * h : 2 * h + 1;       // taps per direction
    unsigned best = 0, bc = 0xFFFFFFFFu;
    for (int k = 0; k < a.numD; k++) {
        const int d = a.minD + k;
        for (int e = threadIdx.x; e < TH * TW; e += 256) {
            const int yy0 = y0 + e / TW - h, xx0 = x0 + e % TW - h;
            int v = 0;
            int yy = yy0, xx = xx0;
            if (a.foot == 2) { yy = refl101(yy0, a.H); xx = refl101(xx0, a.W); }
            if (yy >= 0 && yy < a.H && xx >= d && xx < a.W) {
                v = abs((int)a.L[(size_t)yy * a.lp + xx] - (int)a.R[(size_t)yy * a.rp + xx - d]);
                if (a.square) v = min(v * v, 255);
                if (a.foot == 2) v = v * v;
            }
            dv[e] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TH * LG_X; e += 256) {
            const int r = e / LG_X, c = e % LG_X;
            int s = 0;
            for (int t = 0; t < ext; t++) s += dv[r * TW + c + t];
            hs[e] = s;
        }
        __syncthreads();
        unsigned c = 0;
        for (int t = 0; t < ext; t++) c += (unsigned)hs[(ty + t) * LG_X + tx];
        if (a.foot == 0 && i - d < h) c = 0;
        if (c < bc) { bc = c; best = (unsigned)k; }
        __syncthreads();
    }
    if (i >= a.W || j >= a.H) return;
    const bool inside = a.all || (j >= h && j < a.H - h && i >= h && i < a.W - h);
    if (a.clampT) best = min(best, (unsigned)a.T);
    a.out[(size_t)j * a.op + i] = inside ? (u8)(16 * best) : (u8)0;
}

}  // namespace

bool md_legacy_uses_tables(int method)
{
    return (method == MDISP_LEGACY_SAD_INTE || method == MDISP_LEGACY_STAD_INTE || method == MDISP_LEGACY_SSD_INTE) &&
           md_tune(MDISP_TUNE_SAT_INT_TABLES) <= 0;
}

int md_legacy_disparity(hipStream_t s, const u8* dL, size_t lp, const u8* dR, size_t rp, int H, int W, int method, int winSize,
                        int T, int minD, int numD, u8* dOut, size_t op, int* I, int* best)
{
    MD_REQUIRE(method >= MDISP_LEGACY_AD && method <= MDISP_LEGACY_SSD_BF, MDISP_E_BADARG, "unknown matcher %d", method);
    MD_REQUIRE(numD > 0 && minD >= 0, MDISP_E_BADARG, "bad disparity range");
    LegacyArgs a;
    a.L = dL; a.lp = lp; a.R = dR; a.rp = rp; a.H = H; a.W = W; a.T = T; a.minD = minD; a.numD = numD; a.out = dOut; a.op = op;
    a.clampT = method == MDISP_LEGACY_TAD || method == MDISP_LEGACY_STAD || method == MDISP_LEGACY_STAD_INTE || method == MDISP_LEGACY_STAD_BF;
    a.square = method == MDISP_LEGACY_SD || method == MDISP_LEGACY_SSD || method == MDISP_LEGACY_SSD_INTE || method == MDISP_LEGACY_SSD_BF;
    const bool pixel = method == MDISP_LEGACY_AD || method == MDISP_LEGACY_TAD || method == MDISP_LEGACY_SD;
    const bool inte = method == MDISP_LEGACY_SAD_INTE || method == MDISP_LEGACY_STAD_INTE || method == MDISP_LEGACY_SSD_INTE;
    const bool bf = method == MDISP_LEGACY_SAD_BF || method == MDISP_LEGACY_STAD_BF || method == MDISP_LEGACY_SSD_BF;
    a.foot = inte ? 1 : bf ? 2 : 0;
    a.all = method == MDISP_LEGACY_SAD_BF;
    a.h = 0;
    if (pixel) {
        // Rect(i - numDisparity, j, numDisparity, 1) of the reference leaves the image otherwise (cv::Exception)
        MD_REQUIRE(minD == 0 && numD <= W, MDISP_E_BADARG, "computeAD / TAD / SD need minDisparity = 0 and numDisparity <= cols");
        legacy_pixel_kernel<<<dim3((W + 255) / 256, H), 256, 0, s>>>(a);
    } else {
        MD_REQUIRE(winSize > 0 && winSize % 2 == 1, MDISP_E_BADARG, "winsize must be odd");
        MD_REQUIRE(winSize <= 63, MDISP_E_UNSUPPORTED, "window above 63 is not supported");
        if (inte || bf) MD_REQUIRE(minD + numD - 1 < W, MDISP_E_BADARG, "disparity range wider than the image");
        a.h = winSize / 2;
        if (md_legacy_uses_tables(method)) {
            // the reference's CV_32F integral tables: the winner follows their rounding once entries pass 2^24
            MD_REQUIRE(I && best, MDISP_E_BADARG, "the integral-image matchers need table scratch");
            return md_legacy_inte(s, dL, lp, dR, rp, H, W, minD, numD, winSize, a.square, a.clampT, T, 1, dOut, op, I, best);
        }
        const size_t lds = ((size_t)(LG_X + 2 * a.h) * (LG_Y + 2 * a.h) + (size_t)(LG_Y + 2 * a.h) * LG_X) * sizeof(int);
        legacy_window_kernel<<<dim3((W + LG_X - 1) / LG_X, (H + LG_Y - 1) / LG_Y), 256, lds, s>>>(a);
    }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}
