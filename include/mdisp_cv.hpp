// Drop-in adapter: the reference's own free functions (same names, same cv::Mat signatures) on top
// of libmdisp.so.  Include this header INSTEAD of disp_method/methods_disp.h, gifs.h and
// steerFilter.h in a translation unit that calls the stereo hot path, and link libmdisp.so; the
// callers (cameraCalibration.cpp:163-190 and the disparity branches of methods/methods.cpp) stay
// as they are.  Needs OpenCV's core headers only.
#ifndef MDISP_CV_HPP
#define MDISP_CV_HPP

#include <opencv2/core.hpp>

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
using mdisp::GIF; using mdisp::WGIF; using mdisp::EGIF; using mdisp::SKWGIF;

// disp_method/methods_disp.h:43-44, 48-55, 61-62, 66
inline cv::Mat computeSAD_inteOpti(StereoMatchParam param) { return mdisp::computeSAD_inteOpti(param); }
inline cv::Mat computeSAD_BFOpti(StereoMatchParam param) { return mdisp::computeSAD_BFOpti(param); }
inline void countCensusImg(cv::Mat src, cv::Mat& dst) { mdisp::countCensusImg(src, dst); }
inline void countCensusImg_circle(cv::Mat src, cv::Mat& dst, int radius, int samplePtNum = 8) { mdisp::countCensusImg_circle(src, dst, radius, samplePtNum); }
inline void countCensusImg_rotationInv(cv::Mat src, cv::Mat& dst) { mdisp::countCensusImg_rotationInv(src, dst); }
inline void countCensusImg_uniform(cv::Mat src, cv::Mat& dst) { mdisp::countCensusImg_uniform(src, dst); }
inline void countCensusImg_multiScale(cv::Mat src, cv::Mat& dst, int scale) { mdisp::countCensusImg_multiScale(src, dst, scale); }
inline void countHummingDist(cv::Mat src1, cv::Mat src2, cv::Mat& dst) { mdisp::countHummingDist(src1, src2, dst); }
inline cv::Mat censusStereo(StereoMatchParam param, CENSUS_ALGORITHM method) { return mdisp::censusStereo(param, method); }
inline cv::Mat asw_gifs(StereoMatchParam param, double eps, GIF_TYPE gifType = GIF, int r2 = 5, double namda = 0.1, double h = 0.5)
{ return mdisp::asw_gifs(param, eps, gifType, r2, namda, h); }

// disp_method/gifs.h:17, 21, 27, 33; steerFilter.h:39
inline cv::Mat getGuidedFilter(cv::Mat guidedImg, cv::Mat inputP, int r, double eps) { return mdisp::getGuidedFilter(guidedImg, inputP, r, eps); }
inline cv::Mat getGuidedFilter_wgif(cv::Mat guidedImg, cv::Mat inputP, int r, double eps) { return mdisp::getGuidedFilter_wgif(guidedImg, inputP, r, eps); }
inline cv::Mat getGuidedFilter_egif(cv::Mat guidedImg, cv::Mat inputP, int r, double eps, cv::Mat& beta_)
{ return mdisp::getGuidedFilter_egif(guidedImg, inputP, r, eps, beta_); }
inline cv::Mat getGuidedFilter_skwgif2(cv::Mat guidedImg, cv::Mat inputP, cv::Mat steerFilterW, int r, int r2, double eps, double namuda, double h)
{ return mdisp::getGuidedFilter_skwgif2(guidedImg, inputP, steerFilterW, r, r2, eps, namuda, h); }
inline void steerFilter(cv::Mat src, cv::Mat& dst) { mdisp::steerFilter(src, dst); }

// methods/methods.h:33
inline void getDisparity_SGBM(cv::Mat srcLeft, cv::Mat srcRight, cv::Mat& disparityMap) { mdisp::getDisparity_SGBM(srcLeft, srcRight, disparityMap); }

#endif  // MDISP_CV_HPP
