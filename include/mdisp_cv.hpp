// Drop-in adapter: the reference's own free functions (same names, same cv::Mat signatures) on top
// of libmdisp.so.  Include this header INSTEAD of disp_method/methods_disp.h, gifs.h and
// steerFilter.h in a translation unit that calls the stereo hot path, and link libmdisp.so; the
// callers (cameraCalibration.cpp:163-190 and the disparity branches of methods/methods.cpp) stay
// as they are.  Needs OpenCV's core headers only.
#ifndef MDISP_CV_HPP
#define MDISP_CV_HPP

#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>   // resize(INTER_AREA) of getRectifiedImages stays on the host

#include <algorithm>
#include <cmath>
#include <iostream>
#include <string>
#include <vector>

#include "mdisp_host.hpp"

namespace mdisp {
template <> struct mat_traits<cv::Mat> {
    static bool empty(const cv::Mat& m) { return m.empty(); }
    static mdisp_image view(const cv::Mat& m)
    {
        mdisp_image v;
        v.data = m.data; v.rows = m.rows; v.cols = m.cols; v.channels = m.channels();
        v.depth = m.depth();              // CV_8U .. CV_64F have the MDISP_* values
        v.step_bytes = m.step[0];
        return v;
    }
    static void create(cv::Mat& m, int rows, int cols, int depth, int channels) { m.create(rows, cols, CV_MAKETYPE(depth, channels)); }
};
}  // namespace mdisp

// disp_method/methods_disp.h:5-16, 18-27; gifs.h:4-13
typedef mdisp::StereoMatchParamT<cv::Mat> StereoMatchParam;
using mdisp::CENSUS_ALGORITHM;
using mdisp::BASIC_CENSUS; using mdisp::CIRCLE_CENSUS; using mdisp::ROTATION_INVARIANT_CENSUS; using mdisp::UNIFORM_CENSUS;
using mdisp::MULTISCALE_CENSUS; using mdisp::STATISTIC_MULTISCALE_CENSUS; using mdisp::CENSUS_2017;
using mdisp::GIF_TYPE;
using mdisp::GIF; using mdisp::WGIF; using mdisp::EGIF; using mdisp::SKWGIF; using mdisp::OURS_GIF; using mdisp::OURS_GIF2; using mdisp::OURS_GIF3;

// disp_method/methods_disp.h:43-44, 48-55, 61-62, 66
inline cv::Mat computeSAD_inteOpti(StereoMatchParam param) { return mdisp::computeSAD_inteOpti(param); }
inline cv::Mat computeSAD_BFOpti(StereoMatchParam param) { return mdisp::computeSAD_BFOpti(param); }
inline void countCensusImg(cv::Mat src, cv::Mat& dst) { mdisp::countCensusImg(src, dst); }
inline void countCensusImg_circle(cv::Mat src, cv::Mat& dst, int radius, int samplePtNum = 8) { mdisp::countCensusImg_circle(src, dst, radius, samplePtNum); }
inline void countCensusImg_rotationInv(cv::Mat src, cv::Mat& dst) { mdisp::countCensusImg_rotationInv(src, dst); }
inline void countCensusImg_uniform(cv::Mat src, cv::Mat& dst) { mdisp::countCensusImg_uniform(src, dst); }
inline void countCensusImg_multiScale(cv::Mat src, cv::Mat& dst, int scale) { mdisp::countCensusImg_multiScale(src, dst, scale); }
inline void countCensusImg_multiScale2(cv::Mat src, cv::Mat& dst, int scale) { mdisp::countCensusImg_multiScale2(src, dst, scale); }
inline void countHummingDist(cv::Mat src1, cv::Mat src2, cv::Mat& dst) { mdisp::countHummingDist(src1, src2, dst); }
inline cv::Mat censusStereo(StereoMatchParam param, CENSUS_ALGORITHM method) { return mdisp::censusStereo(param, method); }
inline cv::Mat asw_gifs(StereoMatchParam param, double eps, GIF_TYPE gifType = GIF, int r2 = 5, double namda = 0.1, double h = 0.5)
{ return mdisp::asw_gifs(param, eps, gifType, r2, namda, h); }
// methods_disp.h:69
inline void postProcess_(StereoMatchParam param, cv::Mat dispL, cv::Mat dispR, cv::Mat& filteredDispL, cv::Mat& filteredDispR)
{ mdisp::postProcess_(param, dispL, dispR, filteredDispL, filteredDispR); }

// disp_method/gifs.h:17, 21, 27, 33; steerFilter.h:39
inline cv::Mat getGuidedFilter(cv::Mat guidedImg, cv::Mat inputP, int r, double eps) { return mdisp::getGuidedFilter(guidedImg, inputP, r, eps); }
inline cv::Mat getGuidedFilter_wgif(cv::Mat guidedImg, cv::Mat inputP, int r, double eps) { return mdisp::getGuidedFilter_wgif(guidedImg, inputP, r, eps); }
inline cv::Mat getGuidedFilter_egif(cv::Mat guidedImg, cv::Mat inputP, int r, double eps, cv::Mat& beta_)
{ return mdisp::getGuidedFilter_egif(guidedImg, inputP, r, eps, beta_); }
inline cv::Mat getGuidedFilter_skwgif2(cv::Mat guidedImg, cv::Mat inputP, cv::Mat steerFilterW, int r, int r2, double eps, double namuda, double h)
{ return mdisp::getGuidedFilter_skwgif2(guidedImg, inputP, steerFilterW, r, r2, eps, namuda, h); }
// disp_method/gifs.h:36-38
inline cv::Mat getGuidedFilter_ours_gif(cv::Mat guidedImg, cv::Mat inputP, int r, int r2, double eps, double namuda, double h)
{ return mdisp::getGuidedFilter_ours_gif(guidedImg, inputP, r, r2, eps, namuda, h); }
inline cv::Mat getGuidedFilter_ours_gif2(cv::Mat guidedImg, cv::Mat inputP, int r, int r2, double eps, double namuda, double h)
{ return mdisp::getGuidedFilter_ours_gif2(guidedImg, inputP, r, r2, eps, namuda, h); }
inline cv::Mat getGuidedFilter_ours_gif3(cv::Mat guidedImg, cv::Mat inputP, int r, int r2, double eps, double namuda, double h, cv::Mat& beta_)
{ return mdisp::getGuidedFilter_ours_gif3(guidedImg, inputP, r, r2, eps, namuda, h, beta_); }
inline void steerFilter(cv::Mat src, cv::Mat& dst) { mdisp::steerFilter(src, dst); }

// cv::StereoSGBM::create(...)->compute(...) call sites (methods/methods.cpp:1216-1219, 1291, 1565-1568, 1650, 1893, 1912):
// `mdisp_cv::StereoSGBM::create(0, 64, 7, 490, 1960, 1, 63, 10, 100, 32, mdisp_cv::StereoSGBM::MODE_SGBM)->compute(l, r, d)`
namespace mdisp_cv { typedef mdisp::StereoSGBMT<cv::Mat> StereoSGBM; }
// cv::StereoBM::create(ndisparities, SADWindowSize) + setters + compute (methods/methods.cpp:1205-1213, 1829-1868)
namespace mdisp_cv { typedef mdisp::StereoBMT<cv::Mat> StereoBM; }

// methods/methods.h:33
inline void getDisparity_SGBM(cv::Mat srcLeft, cv::Mat srcRight, cv::Mat& disparityMap) { mdisp::getDisparity_SGBM(srcLeft, srcRight, disparityMap); }
// methods/methods.h:27, 32
inline void getDisparity_BM(cv::Mat srcLeft, cv::Mat srcRight, cv::Mat& disparityMap) { mdisp::getDisparity_BM(srcLeft, srcRight, disparityMap); }
using mdisp::StereoMatchingAlgorithms;
inline void stereoMatching(cv::Mat srcLeft, cv::Mat srcRight, cv::Mat& disparityMap, StereoMatchingAlgorithms algorithmType)
{ mdisp::stereoMatching(srcLeft, srcRight, disparityMap, algorithmType); }

// methods/methods.h:27, methods/methods.cpp:1670-1766.  Same XML keys, same stereoRectify flags
// (CALIB_ZERO_DISPARITY, alpha = 0), maps in CV_16SC2 fixed point, remap on the GPU; the 600-pixel
// canvas, INTER_AREA resize and valid-ROI crop are kept as OpenCV host calls.  (The reference also
// draws debugging rectangles and lines on its canvas after cloning the outputs; that is dropped.)
inline void getRectifiedImages(cv::Mat imgLeft, cv::Mat imgRight, std::string cameraParaPath, cv::Mat& rectifiedLeft,
                               cv::Mat& rectifiedRight)
{
    cv::Size imgSize;
    cv::Mat K1, K2, D1, D2, R, T;
    cv::FileStorage fn(cameraParaPath, cv::FileStorage::READ);
    fn["ImgSize"] >> imgSize;
    fn["Left_CameraInnerPara"] >> K1;   fn["Left_CameraDistPara"] >> D1;
    fn["Right_CameraInnerPara"] >> K2;  fn["Right_CameraDistPara"] >> D2;
    fn["R2L_Rotation_Matrix"] >> R;     fn["R2L_Translate_Matrix"] >> T;
    fn.release();
    auto dbl = [](const cv::Mat& m) { cv::Mat d; m.reshape(1, 1).convertTo(d, CV_64F); return d.clone(); };
    K1 = dbl(K1); K2 = dbl(K2); D1 = dbl(D1); D2 = dbl(D2); R = dbl(R); T = dbl(T);
    double R1[9], R2[9], P1[12], P2[12], Q[16];
    int roi[2][4];
    mdisp::detail::check(mdisp_stereo_rectify(K1.ptr<double>(), D1.ptr<double>(), K2.ptr<double>(), D2.ptr<double>(), (int)D1.total(),
                                              imgSize.width, imgSize.height, R.ptr<double>(), T.ptr<double>(), 1, 0.0,
                                              R1, R2, P1, P2, Q, roi[0], roi[1]));
    const bool isVerticalStereo = std::fabs(P2[7]) > std::fabs(P2[3]);
    if (imgLeft.rows != imgRight.rows && imgLeft.cols != imgRight.cols && imgLeft.rows != imgSize.height && imgLeft.cols != imgSize.width) {
        std::cout << "img reading error" << std::endl;
        return;
    }
    if (imgLeft.empty() || imgRight.empty()) return;
    const double sf = (isVerticalStereo ? 300. : 600.) / std::max(imgSize.width, imgSize.height);
    const int w = cvRound(imgSize.width * sf), h = cvRound(imgSize.height * sf);
    cv::Mat out[2];
    cv::Rect vroi[2];
    for (int k = 0; k < 2; k++) {
        const cv::Mat& src = k == 0 ? imgLeft : imgRight;
        std::vector<int16_t> m1((size_t)imgSize.area() * 2);
        std::vector<uint16_t> m2((size_t)imgSize.area());
        mdisp::detail::check(mdisp_build_rectify_maps(k == 0 ? K1.ptr<double>() : K2.ptr<double>(), k == 0 ? D1.ptr<double>() : D2.ptr<double>(),
                                                      (int)D1.total(), k == 0 ? R1 : R2, k == 0 ? P1 : P2, imgSize.width, imgSize.height,
                                                      m1.data(), m2.data()));
        cv::Mat rimg(imgSize, src.type());
        mdisp_image s = mdisp::mat_traits<cv::Mat>::view(src), d = mdisp::mat_traits<cv::Mat>::view(rimg);
        mdisp::detail::check(mdisp_remap(mdisp::Context::forThisThread().handle(), &s, m1.data(), m2.data(), &d));
        cv::resize(rimg, out[k], cv::Size(w, h), 0, 0, cv::INTER_AREA);
        vroi[k] = cv::Rect(cvRound(roi[k][0] * sf), cvRound(roi[k][1] * sf), cvRound(roi[k][2] * sf), cvRound(roi[k][3] * sf));
    }
    const cv::Rect v = vroi[0] & vroi[1];
    rectifiedLeft = out[0](v).clone();
    rectifiedRight = out[1](v).clone();
}

// methods/method_StereoMatching.h:94-112 (namespace stereomatch_1): the fixed-window matchers on cv::Mat
#ifndef MDISP_CV_NO_DISPARITY_TYPE      // define it when methods/parametersStereo.h (the same enum, :5-9) is included as well
using mdisp::stereomatch_1::DisparityType;
using mdisp::stereomatch_1::DISPARITY_LEFT;
using mdisp::stereomatch_1::DISPARITY_RIGHT;
#define MDISP_CV_DISPARITY_LEFT DISPARITY_LEFT
#else
#define MDISP_CV_DISPARITY_LEFT ((DisparityType)0)
#endif
namespace stereomatch_1 {
inline cv::Mat computeNCC(cv::Mat l, cv::Mat r, DisparityType dispType = MDISP_CV_DISPARITY_LEFT, int winSize = 7, int minDisparity = 0, int numDisparity = 30)
{ return mdisp::stereomatch_1::computeNCC(l, r, (mdisp::stereomatch_1::DisparityType)(int)dispType, winSize, minDisparity, numDisparity); }
inline void computeNCC(cv::Mat l, cv::Mat r, std::vector<cv::Mat>& cost_ds, DisparityType dispType = MDISP_CV_DISPARITY_LEFT, int winSize = 7,
                       int minDisparity = 0, int numDisparity = 30)
{ mdisp::stereomatch_1::computeNCC(l, r, cost_ds, (mdisp::stereomatch_1::DisparityType)(int)dispType, winSize, minDisparity, numDisparity); }
inline cv::Mat computeAD(cv::Mat l, cv::Mat r, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeAD(l, r, minDisparity, numDisparity); }
inline cv::Mat computeSAD(cv::Mat l, cv::Mat r, int winSize = 7, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeSAD(l, r, winSize, minDisparity, numDisparity); }
inline cv::Mat computeSAD_inteOpti(cv::Mat l, cv::Mat r, int winSize = 7, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeSAD_inteOpti(l, r, winSize, minDisparity, numDisparity); }
inline cv::Mat computeSAD_BFOpti(cv::Mat l, cv::Mat r, int winSize = 7, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeSAD_BFOpti(l, r, winSize, minDisparity, numDisparity); }
inline cv::Mat computeTAD(cv::Mat l, cv::Mat r, int threshold_T = 30, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeTAD(l, r, threshold_T, minDisparity, numDisparity); }
inline cv::Mat computeSTAD(cv::Mat l, cv::Mat r, int winSize = 7, int threshold_T = 30, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeSTAD(l, r, winSize, threshold_T, minDisparity, numDisparity); }
inline cv::Mat computeSTAD_inteOpti(cv::Mat l, cv::Mat r, int winSize = 7, int threshold_T = 30, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeSTAD_inteOpti(l, r, winSize, threshold_T, minDisparity, numDisparity); }
inline cv::Mat computeSTAD_BFOpti(cv::Mat l, cv::Mat r, int winSize = 7, int threshold_T = 30, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeSTAD_BFOpti(l, r, winSize, threshold_T, minDisparity, numDisparity); }
inline cv::Mat computeSD(cv::Mat l, cv::Mat r, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeSD(l, r, minDisparity, numDisparity); }
inline cv::Mat computeSSD(cv::Mat l, cv::Mat r, int winSize = 7, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeSSD(l, r, winSize, minDisparity, numDisparity); }
inline cv::Mat computeSSD_inteOpti(cv::Mat l, cv::Mat r, int winSize = 7, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeSSD_inteOpti(l, r, winSize, minDisparity, numDisparity); }
inline cv::Mat computeSSD_BFOpti(cv::Mat l, cv::Mat r, int winSize = 7, int minDisparity = 0, int numDisparity = 30) { return mdisp::stereomatch_1::computeSSD_BFOpti(l, r, winSize, minDisparity, numDisparity); }
// methods/method_StereoMatching.h:171, 178-181
inline cv::Mat getCostSAD_d(cv::Mat leftImg, cv::Mat rightImg, int disparity, DisparityType dispType = MDISP_CV_DISPARITY_LEFT, int winSize = 35)
{ return mdisp::stereomatch_1::getCostSAD_d(leftImg, rightImg, disparity, (mdisp::stereomatch_1::DisparityType)(int)dispType, winSize); }
inline cv::Mat getGuidedFilter(cv::Mat guidedImg, cv::Mat inputP, int r, double eps) { return mdisp::stereomatch_1::getGuidedFilter(guidedImg, inputP, r, eps); }
inline cv::Mat computeAdaptiveWeight_GuidedF(cv::Mat leftImg, cv::Mat rightImg, DisparityType dispType = MDISP_CV_DISPARITY_LEFT, double eps = 0.01,
                                             int winSize = 35, int minDisparity = 186, int numDisparity = 144)
{ return mdisp::stereomatch_1::computeAdaptiveWeight_GuidedF(leftImg, rightImg, (mdisp::stereomatch_1::DisparityType)(int)dispType, eps, winSize, minDisparity, numDisparity); }
// methods/method_StereoMatching.h:123-128, 182-187
inline void computeSimilarity(cv::Mat leftImg, cv::Mat rightImg, std::vector<cv::Mat>& cost_d_imgs, double regularity, double thresC, double thresG,
                              DisparityType dispType, int minDisparity, int numDisparity)
{ mdisp::stereomatch_1::computeSimilarity(leftImg, rightImg, cost_d_imgs, regularity, thresC, thresG, (mdisp::stereomatch_1::DisparityType)(int)dispType, minDisparity, numDisparity); }
inline void computeSimilarity(cv::Mat leftImg, cv::Mat rightImg, std::vector<cv::Mat>& cost_d_imgs, double regularity, double thresC, double thresG,
                              DisparityType dispType, int winSize, int minDisparity, int numDisparity)
{ mdisp::stereomatch_1::computeSimilarity(leftImg, rightImg, cost_d_imgs, regularity, thresC, thresG, (mdisp::stereomatch_1::DisparityType)(int)dispType, winSize, minDisparity, numDisparity); }
inline cv::Mat computeAdaptiveWeight_GuidedF_2(cv::Mat leftImg, cv::Mat rightImg, DisparityType dispType = MDISP_CV_DISPARITY_LEFT, double eps = 0.01,
                                               int winSize = 35, int minDisparity = 186, int numDisparity = 144)
{ return mdisp::stereomatch_1::computeAdaptiveWeight_GuidedF_2(leftImg, rightImg, (mdisp::stereomatch_1::DisparityType)(int)dispType, eps, winSize, minDisparity, numDisparity); }
inline cv::Mat computeAdaptiveWeight_GuidedF_3(cv::Mat leftImg, cv::Mat rightImg, DisparityType dispType = MDISP_CV_DISPARITY_LEFT, double eps = 1e-6,
                                               int winSize = 35, int minDisparity = 186, int numDisparity = 144)
{ return mdisp::stereomatch_1::computeAdaptiveWeight_GuidedF_3(leftImg, rightImg, (mdisp::stereomatch_1::DisparityType)(int)dispType, eps, winSize, minDisparity, numDisparity); }
}  // namespace stereomatch_1

#endif  // MDISP_CV_HPP
