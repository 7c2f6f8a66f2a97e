// Test double, NOT OpenCV: the one imgproc call include/mdisp_cv.hpp keeps on the host.
#ifndef MDISP_FAKE_OPENCV_IMGPROC_HPP
#define MDISP_FAKE_OPENCV_IMGPROC_HPP
#include "core.hpp"
namespace cv {
enum { INTER_LINEAR = 1, INTER_AREA = 3 };
void resize(const Mat& src, Mat& dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
}
#endif
