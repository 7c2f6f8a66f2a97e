// C++ host layer over the C ABI of libmdisp.so: the reference's function-level stereo API
// (cameraCalibration/disp_method/methods_disp.h, gifs.h, steerFilter.h, methods/methods.h) with the
// image type left as a template parameter, so the same code serves cv::Mat (see mdisp_cv.hpp) and
// the small self-contained mdisp::Mat below (used where OpenCV is not installed).
//
// Behaviour kept from the reference:
//   - parameters are passed by value, outputs are freshly allocated images owned by the caller;
//   - on bad parameters (empty image, size mismatch, even winSize) a message goes to std::cout and an
//     EMPTY image is returned (methods_disp.cpp:209-215, 313-319, 847-853, 1149-1155);
//   - StereoMatchParam::maxDisparity is inclusive (methods_disp.cpp:219).
// Failures of the GPU runtime (no device, HIP error) are not part of the reference's vocabulary:
// they throw mdisp::Error.  There is no CPU fallback.
#ifndef MDISP_HOST_HPP
#define MDISP_HOST_HPP

#include <cstdint>
#include <atomic>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mdisp.h"

namespace mdisp {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error("mdisp error " + std::to_string(c) + ": " + m), code(c) {}
};

// ---- a minimal reference-counted image with cv::Mat's value semantics ------------------------
class Mat {
public:
    int rows = 0, cols = 0, channels = 1, depth = MDISP_8U;
    size_t step = 0;
    Mat() = default;
    Mat(int r, int c, int depth_, int ch = 1) { create(r, c, depth_, ch); }
    void create(int r, int c, int depth_, int ch = 1)
    {
        static const int esz[7] = {1, 1, 2, 2, 4, 4, 8};
        rows = r; cols = c; depth = depth_; channels = ch;
        step = (size_t)c * ch * esz[depth_];
        buf_ = std::make_shared<std::vector<uint8_t>>(step * (size_t)r);
    }
    bool empty() const { return !buf_ || rows == 0 || cols == 0; }
    uint8_t* data() { return buf_ ? buf_->data() : nullptr; }
    const uint8_t* data() const { return buf_ ? buf_->data() : nullptr; }
    template <class T> T* ptr(int y) { return reinterpret_cast<T*>(data() + step * (size_t)y); }
    template <class T> const T* ptr(int y) const { return reinterpret_cast<const T*>(data() + step * (size_t)y); }
    template <class T> T& at(int y, int x) { return ptr<T>(y)[x]; }
    template <class T> const T& at(int y, int x) const { return ptr<T>(y)[x]; }

private:
    std::shared_ptr<std::vector<uint8_t>> buf_;
};

// ---- how the layer looks at an image type ------------------------------------------------------
template <class M> struct mat_traits;
template <> struct mat_traits<Mat> {
    static bool empty(const Mat& m) { return m.empty(); }
    static mdisp_image view(const Mat& m)
    {
        mdisp_image v;
        v.data = const_cast<uint8_t*>(m.data()); v.rows = m.rows; v.cols = m.cols; v.channels = m.channels;
        v.depth = m.depth; v.step_bytes = m.step;
        return v;
    }
    static void create(Mat& m, int rows, int cols, int depth, int channels) { m.create(rows, cols, depth, channels); }
};

// ---- enums of the reference, same enumerators and values --------------------------------------
enum CENSUS_ALGORITHM {   // methods_disp.h:18-27
    BASIC_CENSUS = MDISP_BASIC_CENSUS, CIRCLE_CENSUS, ROTATION_INVARIANT_CENSUS, UNIFORM_CENSUS, MULTISCALE_CENSUS,
    STATISTIC_MULTISCALE_CENSUS, CENSUS_2017
};
enum GIF_TYPE { GIF = MDISP_GIF, WGIF, EGIF, SKWGIF, OURS_GIF, OURS_GIF2, OURS_GIF3 };   // gifs.h:4-13

// StereoMatchParam, methods_disp.h:5-16
template <class M> struct StereoMatchParamT {
    M imgLeft, imgRight, imgLeft_C, imgRight_C;
    int winSize = 7;
    int minDisparity = 0;
    int maxDisparity = 30;
    bool isDispLeft = true;
};

// ---- one context per host thread (the reference runs its L and R calls on two std::async threads)
class Context {
public:
    explicit Context(int device = -1, void* stream = nullptr)
    {
        int rc = mdisp_create(&h_, device, stream);
        if (rc) throw Error(rc, mdisp_last_error());
    }
    ~Context() { if (h_) mdisp_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    mdisp_ctx* handle() const { return h_; }
    static Context& forThisThread()
    {
        thread_local Context c;
        return c;
    }

private:
    mdisp_ctx* h_ = nullptr;
};

namespace detail {
inline void bad_parameters() { std::cout << "bad function parameters, please check image and window size." << std::endl; }
inline void check(int rc) { if (rc) throw Error(rc, mdisp_last_error()); }

template <class M, class F> M pair_call(const StereoMatchParamT<M>& p, int outDepth, F&& fn)
{
    typedef mat_traits<M> T;
    if (T::empty(p.imgLeft) || T::empty(p.imgRight)) {
        std::cout << "empty image error" << std::endl;
        bad_parameters();
        return M();
    }
    mdisp_image l = T::view(p.imgLeft), r = T::view(p.imgRight);
    mdisp_param pod = {p.winSize, p.minDisparity, p.maxDisparity, p.isDispLeft ? 1 : 0};
    M out;
    T::create(out, l.rows, l.cols, outDepth, 1);
    mdisp_image o = T::view(out);
    int rc = fn(Context::forThisThread().handle(), &l, &r, &pod, &o);
    if (rc == MDISP_E_BADARG) { bad_parameters(); return M(); }
    check(rc);
    return out;
}
}  // namespace detail

// methods_disp.h:43-44  -> CV_8U disparity
template <class M> M computeSAD_inteOpti(StereoMatchParamT<M> param)
{ return detail::pair_call(param, MDISP_8U, mdisp_sad_inte); }
template <class M> M computeSAD_BFOpti(StereoMatchParamT<M> param)
{ return detail::pair_call(param, MDISP_8U, mdisp_sad_box); }

// methods_disp.h:62 -> CV_64F disparity.  The reference aggregates every Hamming slice with
// getGuidedFilter(img, slice, winSize, 1e-6) (methods_disp.cpp:978) = MDISP_AGG_GUIDED.
template <class M> M censusStereo(StereoMatchParamT<M> param, CENSUS_ALGORITHM method = BASIC_CENSUS, int agg = MDISP_AGG_GUIDED)
{
    return detail::pair_call(param, MDISP_64F, [&](mdisp_ctx* c, const mdisp_image* l, const mdisp_image* r, const mdisp_param* p,
                                                   mdisp_image* o) { return mdisp_census_stereo(c, l, r, p, (int)method, agg, o); });
}

// The reference's asw_gifs aggregates dense-SIFT descriptor slices (computeDisp_Feature(param, costs_ds, FEATURE_SIFT),
// methods_disp.cpp:1162), which this library does not compute.  A call in the reference's own six-argument form therefore
// takes its slices from the source selected here: MDISP_COST_CENSUS (the default) or MDISP_COST_ABSDIFF.  While the
// choice has never been made explicitly, the first such call prints one note on std::cout, so that swapping the include
// cannot change results silently; callers that hold the SIFT slices pass them with MDISP_COST_VOLUME instead.
namespace detail {
inline std::atomic<int>& asw_source_word() { static std::atomic<int> w(-1); return w; }      // -1: never chosen
}
inline void set_asw_cost_source(int cost_source) { detail::asw_source_word().store(cost_source); }
inline int asw_cost_source()
{
    const int v = detail::asw_source_word().load();
    return v < 0 ? MDISP_COST_CENSUS : v;
}
inline int asw_cost_source_for_reference_call()
{
    static std::atomic<bool> told(false);
    if (detail::asw_source_word().load() < 0 && !told.exchange(true))
        std::cout << "asw_gifs: cost slices come from the GPU census-Hamming cost, not from dense SIFT descriptors "
                     "(methods_disp.cpp:1162 is not built); choose with mdisp::set_asw_cost_source() or pass the slices (MDISP_COST_VOLUME)"
                  << std::endl;
    return asw_cost_source();
}

// methods_disp.h:66 -> CV_32F disparity.  cost_source selects what stands in for the dense-SIFT
// slices of computeDisp_Feature; with MDISP_COST_VOLUME the caller passes those slices.
template <class M> M asw_gifs(StereoMatchParamT<M> param, double eps, GIF_TYPE gifType, int r2, double namda,
                              double h, int cost_source, const float* costs = nullptr)
{
    return detail::pair_call(param, MDISP_32F, [&](mdisp_ctx* c, const mdisp_image* l, const mdisp_image* r, const mdisp_param* p,
                                                   mdisp_image* o) {
        return mdisp_asw_gifs(c, l, r, p, eps, (int)gifType, r2, namda, h, cost_source, costs, o);
    });
}
// the reference's own form (methods_disp.h:66): the slice source is the process-wide choice above
template <class M> M asw_gifs(StereoMatchParamT<M> param, double eps, GIF_TYPE gifType = GIF, int r2 = 5, double namda = 0.1, double h = 0.5)
{ return asw_gifs(param, eps, gifType, r2, namda, h, asw_cost_source_for_reference_call(), nullptr); }

// ximgproc::fastGlobalSmootherFilter(guide, src, dst, lambda, sigma_color, lambda_attenuation, num_iter); dst gets src's type
template <class M> void fastGlobalSmootherFilter(M guide, M src, M& dst, double lambda, double sigma_color, double lambda_attenuation = 0.25,
                                                 int num_iter = 3)
{
    typedef mat_traits<M> T;
    mdisp_image g = T::view(guide), s = T::view(src);
    T::create(dst, s.rows, s.cols, s.depth, 1);
    mdisp_image d = T::view(dst);
    detail::check(mdisp_fgs_filter(Context::forThisThread().handle(), &g, &s, lambda, sigma_color, lambda_attenuation, num_iter, &d));
}

// postProcess_ (methods_disp.h:69, methods_disp.cpp:1299-1313): createDisparityWLSFilterGeneric(false) with lambda 8000 and
// sigmaColor 0.5 over the whole frame = the fast global smoother guided by the left / right view
template <class M> void postProcess_(StereoMatchParamT<M> param, M dispL, M dispR, M& filteredDispL, M& filteredDispR)
{
    fastGlobalSmootherFilter(param.imgLeft, dispL, filteredDispL, 8000.0, 0.5);
    fastGlobalSmootherFilter(param.imgRight, dispR, filteredDispR, 8000.0, 0.5);
}

// countCensusImg* (methods_disp.h:46-58): dst is allocated here
template <class M> void countCensusImg(M src, M& dst, CENSUS_ALGORITHM method = BASIC_CENSUS, int arg = 0)
{
    typedef mat_traits<M> T;
    mdisp_image s = T::view(src);
    T::create(dst, s.rows, s.cols, MDISP_8U, 1);
    mdisp_image d = T::view(dst);
    detail::check(mdisp_census_transform(Context::forThisThread().handle(), &s, (int)method, arg, &d));
}
template <class M> void countCensusImg_rotationInv(M src, M& dst) { countCensusImg(src, dst, ROTATION_INVARIANT_CENSUS); }
template <class M> void countCensusImg_uniform(M src, M& dst) { countCensusImg(src, dst, UNIFORM_CENSUS); }
template <class M> void countCensusImg_multiScale(M src, M& dst, int scale) { countCensusImg(src, dst, MULTISCALE_CENSUS, scale); }
template <class M> void countCensusImg_multiScale2(M src, M& dst, int scale) { countCensusImg(src, dst, STATISTIC_MULTISCALE_CENSUS, scale); }
template <class M> void countCensusImg_circle(M src, M& dst, int radius, int samplePtNum = 8)
{
    if (samplePtNum != 8) throw Error(MDISP_E_UNSUPPORTED, "countCensusImg_circle: only 8 sample points are built");
    countCensusImg(src, dst, CIRCLE_CENSUS, radius);
}

// countHummingDist (methods_disp.h:60): CV_32S popcount(a ^ b)
template <class M> void countHummingDist(M src1, M src2, M& dst)
{
    typedef mat_traits<M> T;
    mdisp_image a = T::view(src1), b = T::view(src2);
    T::create(dst, a.rows, a.cols, MDISP_32S, 1);
    mdisp_image d = T::view(dst);
    detail::check(mdisp_hamming_dist(Context::forThisThread().handle(), &a, &b, &d));
}

// gifs.h:17-33
template <class M> M guidedFilter_(M guide, M P, int r, double eps, int type, const M* steerW, int r2, double namuda, M* beta)
{
    typedef mat_traits<M> T;
    mdisp_image g = T::view(guide), p = T::view(P), sw, bv;
    M out;
    T::create(out, g.rows, g.cols, MDISP_32F, 1);
    mdisp_image o = T::view(out);
    if (steerW) sw = T::view(*steerW);
    if (beta) { T::create(*beta, g.rows, g.cols, MDISP_32F, 1); bv = T::view(*beta); }
    detail::check(mdisp_guided_filter(Context::forThisThread().handle(), &g, &p, r, eps, type, steerW ? &sw : nullptr, r2, namuda, &o,
                                      beta ? &bv : nullptr));
    return out;
}
// a 3- or 6-channel guide takes the multi-channel form of gifs.cpp:63-134 (per-channel coefficients, dot products)
template <class M> M getGuidedFilter(M guidedImg, M inputP, int r, double eps)
{
    typedef mat_traits<M> T;
    mdisp_image g = T::view(guidedImg);
    if (g.channels == 3 || g.channels == 6) {
        mdisp_image p = T::view(inputP);
        if (g.rows != p.rows || g.cols != p.cols) return M();
        M out;
        T::create(out, g.rows, g.cols, MDISP_32F, 1);
        mdisp_image o = T::view(out);
        detail::check(mdisp_guided_filter_multi(Context::forThisThread().handle(), &g, &p, r, eps, MDISP_GIF_MC_GIFS, &o));
        return out;
    }
    return guidedFilter_<M>(guidedImg, inputP, r, eps, MDISP_GIF, nullptr, 0, 0.0, nullptr);
}
template <class M> M getGuidedFilter_wgif(M guidedImg, M inputP, int r, double eps)
{ return guidedFilter_<M>(guidedImg, inputP, r, eps, MDISP_WGIF, nullptr, 0, 0.0, nullptr); }
template <class M> M getGuidedFilter_egif(M guidedImg, M inputP, int r, double eps, M& beta)
{ return guidedFilter_<M>(guidedImg, inputP, r, eps, MDISP_EGIF, nullptr, 0, 0.0, &beta); }
template <class M> M getGuidedFilter_skwgif2(M guidedImg, M inputP, M steerFilterW, int r, int r2, double eps, double namuda, double h = 0.5)
{ (void)h; return guidedFilter_<M>(guidedImg, inputP, r, eps, MDISP_SKWGIF, &steerFilterW, r2, namuda, nullptr); }
// gifs.h:36-38 (`h` is accepted and unused, as in the reference)
template <class M> M getGuidedFilter_ours_gif(M guidedImg, M inputP, int r, int r2, double eps, double namuda, double h)
{ (void)h; return guidedFilter_<M>(guidedImg, inputP, r, eps, MDISP_OURS_GIF, nullptr, r2, namuda, nullptr); }
template <class M> M getGuidedFilter_ours_gif2(M guidedImg, M inputP, int r, int r2, double eps, double namuda, double h)
{ (void)h; return guidedFilter_<M>(guidedImg, inputP, r, eps, MDISP_OURS_GIF2, nullptr, r2, namuda, nullptr); }
template <class M> M getGuidedFilter_ours_gif3(M guidedImg, M inputP, int r, int r2, double eps, double namuda, double h, M& beta_)
{ (void)h; return guidedFilter_<M>(guidedImg, inputP, r, eps, MDISP_OURS_GIF3, nullptr, r2, namuda, &beta_); }

// steerFilter.h:39 -> CV_64F
template <class M> void steerFilter(M src, M& dst)
{
    typedef mat_traits<M> T;
    mdisp_image s = T::view(src);
    T::create(dst, s.rows, s.cols, MDISP_64F, 1);
    mdisp_image d = T::view(dst);
    detail::check(mdisp_steer_filter(Context::forThisThread().handle(), &s, &d));
}

// SGM with the reference-owned census-Hamming cost (StereoSGBM's recursion and selection rules) -> CV_16S x16
template <class M> void sgbmCompute(M left, M right, const mdisp_sgbm_params& p, M& disp16)
{
    typedef mat_traits<M> T;
    mdisp_image l = T::view(left), r = T::view(right);
    T::create(disp16, l.rows, l.cols, MDISP_16S, 1);
    mdisp_image d = T::view(disp16);
    detail::check(mdisp_sgbm(Context::forThisThread().handle(), &l, &r, &p, &d));
}

// the same over a sequence of pairs, copies overlapped with the computation (mdisp_sgbm_batch)
template <class M> void sgbmComputeBatch(const std::vector<M>& lefts, const std::vector<M>& rights, const mdisp_sgbm_params& p,
                                         std::vector<M>& disps16, int chunk = 0)
{
    typedef mat_traits<M> T;
    const size_t n = lefts.size();
    disps16.resize(n);
    if (n == 0 || rights.size() != n) return;
    std::vector<const uint8_t*> pl(n), pr(n);
    std::vector<int16_t*> po(n);
    const mdisp_image first = T::view(lefts[0]);
    for (size_t i = 0; i < n; i++) {
        const mdisp_image l = T::view(lefts[i]), r = T::view(rights[i]);
        if (l.rows != first.rows || l.cols != first.cols || r.rows != first.rows || r.cols != first.cols || l.channels != 1 ||
            r.channels != 1 || l.depth != MDISP_8U || r.depth != MDISP_8U || l.step_bytes != first.step_bytes ||
            r.step_bytes != first.step_bytes) {
            std::cout << "bad function parameters, please check image and window size." << std::endl;
            disps16.clear();
            return;
        }
        T::create(disps16[i], first.rows, first.cols, MDISP_16S, 1);
        const mdisp_image d = T::view(disps16[i]);
        if (d.step_bytes != (size_t)first.cols * 2) { disps16.clear(); return; }
        pl[i] = static_cast<const uint8_t*>(l.data); pr[i] = static_cast<const uint8_t*>(r.data); po[i] = static_cast<int16_t*>(d.data);
    }
    detail::check(mdisp_sgbm_batch(Context::forThisThread().handle(), pl.data(), pr.data(), (int)n, first.rows, first.cols,
                                   first.step_bytes, &p, po.data(), chunk));
}

// cv::StereoSGBM as the reference uses it (methods/methods.cpp:1216-1219, 1291, 1893, 1912): create() with the
// library's argument order and defaults, compute() -> CV_16S x16.  Birchfield-Tomasi block cost, MODE_SGBM / MODE_HH.
template <class M> class StereoSGBMT {
public:
    enum { MODE_SGBM = MDISP_SGBM_MODE_SGBM, MODE_HH = MDISP_SGBM_MODE_HH };
    static StereoSGBMT create(int minDisparity = 0, int numDisparities = 16, int blockSize = 3, int P1 = 0, int P2 = 0,
                              int disp12MaxDiff = 0, int preFilterCap = 0, int uniquenessRatio = 0, int speckleWindowSize = 0,
                              int speckleRange = 0, int mode = MODE_SGBM)
    {
        StereoSGBMT s;
        const mdisp_stereosgbm_params p = {minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap,
                                           uniquenessRatio, speckleWindowSize, speckleRange, mode};
        s.params = p;
        return s;
    }
    StereoSGBMT* operator->() { return this; }          // so that `create(...)->compute(...)` reads as with cv::Ptr
    void compute(M left, M right, M& disparity) const
    {
        typedef mat_traits<M> T;
        mdisp_image l = T::view(left), r = T::view(right);
        T::create(disparity, l.rows, l.cols, MDISP_16S, 1);
        mdisp_image d = T::view(disparity);
        detail::check(mdisp_stereosgbm_compute(Context::forThisThread().handle(), &l, &r, &params, &d));
    }
    mdisp_stereosgbm_params params;
};

// methods/methods.h:33, methods.cpp:1883-1921: StereoSGBM::create(0, 48, 7), compute, disp / 16
// (CV_16S MatExpr divide: round to nearest, ties to even)
template <class M> void getDisparity_SGBM(M srcLeft, M srcRight, M& disparityMap)
{
    typedef mat_traits<M> T;
    if (T::empty(srcLeft) || T::empty(srcRight)) { std::cout << " --(!) Error reading images " << std::endl; return; }
    StereoSGBMT<M>::create(0, 48, 7).compute(srcLeft, srcRight, disparityMap);
    mdisp_image d = T::view(disparityMap);
    for (int y = 0; y < d.rows; y++) {
        int16_t* row = reinterpret_cast<int16_t*>(static_cast<uint8_t*>(d.data) + d.step_bytes * (size_t)y);
        for (int x = 0; x < d.cols; x++) {
            const int v = row[x], q = v >> 4, rem = v & 15;          // floor division by 16
            row[x] = (int16_t)(q + (rem > 8 || (rem == 8 && (q & 1))));
        }
    }
}

// cv::StereoBM as the reference uses it (methods/methods.cpp:1205-1213, 1829-1838): create(numDisparities, blockSize), the
// library's defaults for every other parameter, its setter names, compute() -> CV_16S x16.
template <class M> class StereoBMT {
public:
    static StereoBMT create(int numDisparities = 0, int blockSize = 21)
    {
        StereoBMT s;
        const mdisp_stereobm_params p = {0, numDisparities > 0 ? numDisparities : 64, blockSize, 31, 10, 15, 0, 0, -1};
        s.params = p;
        return s;
    }
    StereoBMT* operator->() { return this; }
    void setMinDisparity(int v) { params.minDisparity = v; }
    void setNumDisparities(int v) { params.numDisparities = v; }
    void setBlockSize(int v) { params.blockSize = v; }
    void setPreFilterCap(int v) { params.preFilterCap = v; }
    void setTextureThreshold(int v) { params.textureThreshold = v; }
    void setUniquenessRatio(int v) { params.uniquenessRatio = v; }
    void setSpeckleWindowSize(int v) { params.speckleWindowSize = v; }
    void setSpeckleRange(int v) { params.speckleRange = v; }
    void setDisp12MaxDiff(int v) { params.disp12MaxDiff = v; }
    void compute(M left, M right, M& disparity) const
    {
        typedef mat_traits<M> T;
        mdisp_image l = T::view(left), r = T::view(right);
        T::create(disparity, l.rows, l.cols, MDISP_16S, 1);
        mdisp_image d = T::view(disparity);
        detail::check(mdisp_stereobm_compute(Context::forThisThread().handle(), &l, &r, &params, &d));
    }
    mdisp_stereobm_params params;
};

// methods/methods.h:32, methods.cpp:1825-1881: the reference's StereoBM parameter set; colour input is converted to gray first
template <class M> void getDisparity_BM(M srcLeft, M srcRight, M& disparityMap)
{
    typedef mat_traits<M> T;
    StereoBMT<M> sbm = StereoBMT<M>::create(16 * 9, 35);
    sbm.setMinDisparity(186);
    sbm.setTextureThreshold(10);
    sbm.setDisp12MaxDiff(-1);
    sbm.setPreFilterCap(31);
    sbm.setUniquenessRatio(25);
    sbm.setSpeckleRange(32);
    sbm.setSpeckleWindowSize(0);
    if (T::empty(srcLeft) || T::empty(srcRight)) { std::cout << " --(!) Error reading images " << std::endl; return; }
    sbm.compute(srcLeft, srcRight, disparityMap);
}

// methods/parametersStereo.h:11-24
enum StereoMatchingAlgorithms {
    BM = 0, SGBM = 1, ADAPTIVE_WEIGHT = 2, ADAPTIVE_WEIGHT_8DIRECT = 3, ADAPTIVE_WEIGHT_GEODESIC = 4, ADAPTIVE_WEIGHT_BILATERAL_GRID = 5,
    ADAPTIVE_WEIGHT_BLO1 = 6, ADAPTIVE_WEIGHT_GUIDED_FILTER = 7, ADAPTIVE_WEIGHT_GUIDED_FILTER_2 = 8, ADAPTIVE_WEIGHT_GUIDED_FILTER_3 = 9,
    ADAPTIVE_WEIGHT_MEDIAN = 10
};

namespace detail {
// stereomatch_1::computeAdaptiveWeight_GuidedF, methods/method_StereoMatching.cpp:3685-3781 (CV_8UC3 views -> CV_32F map)
// which: 1 = computeAdaptiveWeight_GuidedF, 2 = _GuidedF_2, 3 = _GuidedF_3 (:3794-3955)
template <class M> M adaptive_weight_guidedf(M leftImg, M rightImg, bool isLeft, double eps, int winSize, int minDisparity, int numDisparity,
                                             int which = 1)
{
    typedef mat_traits<M> T;
    mdisp_image l = T::view(leftImg), r = T::view(rightImg);
    M out;
    T::create(out, l.rows, l.cols, MDISP_32F, 1);
    mdisp_image o = T::view(out);
    check(mdisp_adaptive_weight_guidedf_n(Context::forThisThread().handle(), &l, &r, which, isLeft, eps, winSize, minDisparity, numDisparity, &o));
    return out;
}
}  // namespace detail

// methods/methods.h:27, methods.cpp:1776-1813.  BM, SGBM and ADAPTIVE_WEIGHT_GUIDED_FILTER run on the GPU; the other
// adaptive-weight research variants of namespace stereomatch_1 are not built and leave disparityMap untouched after a message.
template <class M> void stereoMatching(M srcLeft, M srcRight, M& disparityMap, StereoMatchingAlgorithms algorithmType)
{
    switch (algorithmType) {
    case BM: getDisparity_BM(srcLeft, srcRight, disparityMap); break;
    case SGBM: getDisparity_SGBM(srcLeft, srcRight, disparityMap); break;
    case ADAPTIVE_WEIGHT_GUIDED_FILTER:      // methods.cpp:1801-1803
        disparityMap = detail::adaptive_weight_guidedf(srcLeft, srcRight, true, 0.01, 15, 10, 150);
        break;
    case ADAPTIVE_WEIGHT_GUIDED_FILTER_2:    // methods.cpp:1804-1806
        disparityMap = detail::adaptive_weight_guidedf(srcLeft, srcRight, true, 0.01, 13, 10, 150, 2);
        break;
    case ADAPTIVE_WEIGHT_GUIDED_FILTER_3:    // methods.cpp:1807-1809
        disparityMap = detail::adaptive_weight_guidedf(srcLeft, srcRight, true, 1e-6, 15, 0, 64, 3);
        break;
    default: std::cout << "stereoMatching: algorithm " << (int)algorithmType << " is not built (BM, SGBM and ADAPTIVE_WEIGHT_GUIDED_FILTER* are)" << std::endl;
    }
}

// PatchMatch::postProcess validity test, methods/patchmatch.cpp:817-828
template <class M> void lrCheck(M dispL, M dispR, float thresh, M& validL, M& validR)
{
    typedef mat_traits<M> T;
    mdisp_image a = T::view(dispL), b = T::view(dispR);
    T::create(validL, a.rows, a.cols, MDISP_8U, 1);
    T::create(validR, a.rows, a.cols, MDISP_8U, 1);
    mdisp_image vl = T::view(validL), vr = T::view(validR);
    detail::check(mdisp_lr_check(Context::forThisThread().handle(), &a, &b, thresh, &vl, &vr));
}

// The maps of initUndistortRectifyMap (methods/methods.cpp:1705-1706) kept on the device: built once per
// calibration (on the GPU, double precision), then `remap` = cv::remap(src, dst, map1, map2, INTER_LINEAR) per frame.
class RectifyMaps {
public:
    RectifyMaps(const double K[9], const double* dist, int ndist, const double R[9], const double P[12], int width, int height)
    { detail::check(mdisp_rectify_maps_create(Context::forThisThread().handle(), K, dist, ndist, R, P, width, height, &maps_)); }
    ~RectifyMaps() { mdisp_rectify_maps_destroy(maps_); }
    RectifyMaps(const RectifyMaps&) = delete;
    RectifyMaps& operator=(const RectifyMaps&) = delete;
    template <class M> void remap(M src, M& dst, int width, int height) const
    {
        typedef mat_traits<M> T;
        mdisp_image s = T::view(src);
        T::create(dst, height, width, MDISP_8U, s.channels);
        mdisp_image d = T::view(dst);
        detail::check(mdisp_remap_with(Context::forThisThread().handle(), maps_, &s, &d));
    }
private:
    mdisp_rectify_maps* maps_ = nullptr;
};

// the rectification step of getRectifiedImages (methods/methods.cpp:1705-1745): maps and remap on the GPU.
// K, dist, R, P are row-major.
template <class M> void rectifyImage(M src, const double K[9], const double* dist, int ndist, const double R[9], const double P[12], M& dst)
{
    typedef mat_traits<M> T;
    mdisp_image s = T::view(src);
    RectifyMaps maps(K, dist, ndist, R, P, s.cols, s.rows);
    maps.remap(src, dst, s.cols, s.rows);
}

// disparity -> 3-D points (methods/methods.cpp:2039-2076 and :2300-2352); xyz is CV_32FC3
template <class M> void reprojectQ(M disparityMap, const double Q[16], M& xyz)
{
    typedef mat_traits<M> T;
    mdisp_image d = T::view(disparityMap);
    T::create(xyz, d.rows, d.cols, MDISP_32F, 3);
    mdisp_image o = T::view(xyz);
    detail::check(mdisp_reproject_q(Context::forThisThread().handle(), &d, Q, &o));
}
template <class M> void reprojectTriangulate(M disparityMap, const double K_new[9], double baseline, const M* originBGRMap, M& xyz)
{
    typedef mat_traits<M> T;
    mdisp_image d = T::view(disparityMap), c;
    if (originBGRMap) c = T::view(*originBGRMap);
    T::create(xyz, d.rows, d.cols, MDISP_32F, 3);
    mdisp_image o = T::view(xyz);
    detail::check(mdisp_reproject_triangulate(Context::forThisThread().handle(), &d, K_new, baseline, originBGRMap ? &c : nullptr, &o));
}

typedef StereoMatchParamT<Mat> StereoMatchParam;
typedef StereoSGBMT<Mat> StereoSGBM;
typedef StereoBMT<Mat> StereoBM;

// namespace stereomatch_1, methods/method_StereoMatching.h:94-112: the fixed-window matchers with the
// reference's names, argument order, defaults and error behaviour ("winsize must be odd" + empty image).
namespace stereomatch_1 {
template <class M> M legacy_(int method, M leftImg, M rightImg, int winSize, int threshold_T, int minDisparity, int numDisparity)
{
    typedef mat_traits<M> T;
    const bool windowed = method != MDISP_LEGACY_AD && method != MDISP_LEGACY_TAD && method != MDISP_LEGACY_SD;
    if (windowed && winSize % 2 == 0) { std::cout << "winsize must be odd" << std::endl; return M(); }
    mdisp_image l = T::view(leftImg), r = T::view(rightImg);
    M out;
    T::create(out, l.rows, l.cols, MDISP_8U, 1);
    mdisp_image o = T::view(out);
    detail::check(mdisp_legacy_disparity(Context::forThisThread().handle(), &l, &r, method, winSize, threshold_T, minDisparity,
                                         numDisparity, &o));
    return out;
}
// computeNCC, both overloads (method_StereoMatching.h:116-120; DisparityType: parametersStereo.h:5-9).  An even window
// or a size mismatch returns an empty image / leaves cost_ds alone, without a message, as the reference does.
enum DisparityType { DISPARITY_LEFT = 0, DISPARITY_RIGHT = 1 };
template <class M> M computeNCC(M leftImg, M rightImg, DisparityType dispType = DISPARITY_LEFT, int winSize = 7, int minDisparity = 0,
                                int numDisparity = 30)
{
    typedef mat_traits<M> T;
    mdisp_image l = T::view(leftImg), r = T::view(rightImg);
    if (l.rows != r.rows || l.cols != r.cols || winSize % 2 == 0) return M();
    M out;
    T::create(out, l.rows, l.cols, MDISP_32F, 1);
    mdisp_image o = T::view(out);
    detail::check(mdisp_ncc(Context::forThisThread().handle(), &l, &r, dispType == DISPARITY_LEFT, winSize, minDisparity, numDisparity, &o, nullptr));
    return out;
}
template <class M> void computeNCC(M leftImg, M rightImg, std::vector<M>& cost_ds, DisparityType dispType = DISPARITY_LEFT, int winSize = 7,
                                   int minDisparity = 0, int numDisparity = 30)
{
    typedef mat_traits<M> T;
    mdisp_image l = T::view(leftImg), r = T::view(rightImg);
    if (l.rows != r.rows || l.cols != r.cols || winSize % 2 == 0) return;
    cost_ds.clear();
    std::vector<float> vol((size_t)numDisparity * l.rows * l.cols);
    detail::check(mdisp_ncc(Context::forThisThread().handle(), &l, &r, dispType == DISPARITY_LEFT, winSize, minDisparity, numDisparity, nullptr,
                            vol.data()));
    for (int k = 0; k < numDisparity; k++) {
        M slice;
        T::create(slice, l.rows, l.cols, MDISP_32F, 1);
        mdisp_image sv = T::view(slice);
        for (int y = 0; y < l.rows; y++)
            std::memcpy((char*)sv.data + (size_t)y * sv.step_bytes, vol.data() + ((size_t)k * l.rows + y) * l.cols, (size_t)l.cols * sizeof(float));
        cost_ds.push_back(slice);
    }
}
template <class M> M computeAD(M leftImg, M rightImg, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_AD, leftImg, rightImg, 1, 0, minDisparity, numDisparity); }
template <class M> M computeSAD(M leftImg, M rightImg, int winSize = 7, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_SAD, leftImg, rightImg, winSize, 0, minDisparity, numDisparity); }
template <class M> M computeSAD_inteOpti(M leftImg, M rightImg, int winSize = 7, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_SAD_INTE, leftImg, rightImg, winSize, 0, minDisparity, numDisparity); }
template <class M> M computeSAD_BFOpti(M leftImg, M rightImg, int winSize = 7, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_SAD_BF, leftImg, rightImg, winSize, 0, minDisparity, numDisparity); }
template <class M> M computeTAD(M leftImg, M rightImg, int threshold_T = 30, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_TAD, leftImg, rightImg, 1, threshold_T, minDisparity, numDisparity); }
template <class M> M computeSTAD(M leftImg, M rightImg, int winSize = 7, int threshold_T = 30, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_STAD, leftImg, rightImg, winSize, threshold_T, minDisparity, numDisparity); }
template <class M> M computeSTAD_inteOpti(M leftImg, M rightImg, int winSize = 7, int threshold_T = 30, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_STAD_INTE, leftImg, rightImg, winSize, threshold_T, minDisparity, numDisparity); }
template <class M> M computeSTAD_BFOpti(M leftImg, M rightImg, int winSize = 7, int threshold_T = 30, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_STAD_BF, leftImg, rightImg, winSize, threshold_T, minDisparity, numDisparity); }
template <class M> M computeSD(M leftImg, M rightImg, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_SD, leftImg, rightImg, 1, 0, minDisparity, numDisparity); }
template <class M> M computeSSD(M leftImg, M rightImg, int winSize = 7, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_SSD, leftImg, rightImg, winSize, 0, minDisparity, numDisparity); }
template <class M> M computeSSD_inteOpti(M leftImg, M rightImg, int winSize = 7, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_SSD_INTE, leftImg, rightImg, winSize, 0, minDisparity, numDisparity); }
template <class M> M computeSSD_BFOpti(M leftImg, M rightImg, int winSize = 7, int minDisparity = 0, int numDisparity = 30)
{ return legacy_<M>(MDISP_LEGACY_SSD_BF, leftImg, rightImg, winSize, 0, minDisparity, numDisparity); }
// methods/method_StereoMatching.h:171: one of the views is the one the caller bordered (wider); empty image where the
// reference returns one (even window, other view not wider)
template <class M> M getCostSAD_d(M leftImg, M rightImg, int disparity, DisparityType dispType = DISPARITY_LEFT, int winSize = 35)
{
    typedef mat_traits<M> T;
    mdisp_image l = T::view(leftImg), r = T::view(rightImg);
    if (winSize % 2 == 0) { std::cout << "winsize must be odd" << std::endl; return M(); }
    const mdisp_image& ref = dispType == DISPARITY_LEFT ? l : r;
    const mdisp_image& oth = dispType == DISPARITY_LEFT ? r : l;
    if (oth.cols <= ref.cols) return M();
    M out;
    T::create(out, ref.rows, ref.cols, MDISP_32F, 1);
    mdisp_image o = T::view(out);
    detail::check(mdisp_cost_sad_d(Context::forThisThread().handle(), &l, &r, disparity, dispType == DISPARITY_LEFT, winSize, &o));
    return out;
}
// methods/method_StereoMatching.h:178 with a 3- or 6-channel CV_8U guide; empty image on a size mismatch
template <class M> M getGuidedFilter(M guidedImg, M inputP, int r, double eps)
{
    typedef mat_traits<M> T;
    mdisp_image g = T::view(guidedImg), p = T::view(inputP);
    if (g.rows != p.rows || g.cols != p.cols) return M();
    M out;
    T::create(out, g.rows, g.cols, MDISP_32F, 1);
    mdisp_image o = T::view(out);
    detail::check(mdisp_guided_filter_multi(Context::forThisThread().handle(), &g, &p, r, eps, MDISP_GIF_MC_STEREOMATCH, &o));
    return out;
}
// methods/method_StereoMatching.h:179-181
template <class M> M computeAdaptiveWeight_GuidedF(M leftImg, M rightImg, DisparityType dispType = DISPARITY_LEFT, double eps = 0.01,
                                                   int winSize = 35, int minDisparity = 186, int numDisparity = 144)
{ return detail::adaptive_weight_guidedf(leftImg, rightImg, dispType == DISPARITY_LEFT, eps, winSize, minDisparity, numDisparity); }
// methods/method_StereoMatching.h:182-187
template <class M> M computeAdaptiveWeight_GuidedF_2(M leftImg, M rightImg, DisparityType dispType = DISPARITY_LEFT, double eps = 0.01,
                                                     int winSize = 35, int minDisparity = 186, int numDisparity = 144)
{ return detail::adaptive_weight_guidedf(leftImg, rightImg, dispType == DISPARITY_LEFT, eps, winSize, minDisparity, numDisparity, 2); }
template <class M> M computeAdaptiveWeight_GuidedF_3(M leftImg, M rightImg, DisparityType dispType = DISPARITY_LEFT, double eps = 1e-6,
                                                     int winSize = 35, int minDisparity = 186, int numDisparity = 144)
{ return detail::adaptive_weight_guidedf(leftImg, rightImg, dispType == DISPARITY_LEFT, eps, winSize, minDisparity, numDisparity, 3); }
// computeSimilarity (method_StereoMatching.cpp:1108-1348): border = 0 for the nine-argument overload, winSize / 2 for the other
template <class M> void computeSimilarity_(M leftImg, M rightImg, std::vector<M>& cost_d_imgs, double regularity, double thresC, double thresG,
                                           DisparityType dispType, int winSize, int minDisparity, int numDisparity)
{
    typedef mat_traits<M> T;
    cost_d_imgs.clear();
    mdisp_image l = T::view(leftImg), r = T::view(rightImg);
    if (l.rows != r.rows || l.cols != r.cols) return;
    const int b = winSize / 2, OH = l.rows + 2 * b, OW = l.cols + 2 * b;
    std::vector<float> vol((size_t)numDisparity * OH * OW);
    detail::check(mdisp_compute_similarity(Context::forThisThread().handle(), &l, &r, dispType == DISPARITY_LEFT, regularity, thresC, thresG, winSize,
                                           minDisparity, numDisparity, vol.data()));
    for (int k = 0; k < numDisparity; k++) {
        M slice;
        T::create(slice, OH, OW, MDISP_32F, 1);
        mdisp_image sv = T::view(slice);
        for (int y = 0; y < OH; y++)
            std::memcpy((char*)sv.data + (size_t)y * sv.step_bytes, vol.data() + ((size_t)k * OH + y) * OW, (size_t)OW * sizeof(float));
        cost_d_imgs.push_back(slice);
    }
}
template <class M> void computeSimilarity(M leftImg, M rightImg, std::vector<M>& cost_d_imgs, double regularity, double thresC, double thresG,
                                          DisparityType dispType, int minDisparity, int numDisparity)
{ computeSimilarity_(leftImg, rightImg, cost_d_imgs, regularity, thresC, thresG, dispType, 0, minDisparity, numDisparity); }
template <class M> void computeSimilarity(M leftImg, M rightImg, std::vector<M>& cost_d_imgs, double regularity, double thresC, double thresG,
                                          DisparityType dispType, int winSize, int minDisparity, int numDisparity)
{
    if (winSize % 2 == 0) return;
    computeSimilarity_(leftImg, rightImg, cost_d_imgs, regularity, thresC, thresG, dispType, winSize, minDisparity, numDisparity);
}
}  // namespace stereomatch_1

}  // namespace mdisp
#endif  // MDISP_HOST_HPP
