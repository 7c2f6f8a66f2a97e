// Compile-only check of include/mdisp_cv.hpp against the OpenCV test double in tests/cpp/fake_cv: the adapter is
// included twice (the include guard must cover the whole file) and every boundary function is named once so that
// its body and the templates behind it are instantiated.
#include <vector>
#include "mdisp_cv.hpp"
#include "mdisp_cv.hpp"

void use_every_boundary_function(cv::Mat l, cv::Mat r, std::string xml)
{
    StereoMatchParam p;
    p.imgLeft = l; p.imgRight = r; p.winSize = 7; p.minDisparity = 0; p.maxDisparity = 63; p.isDispLeft = true;
    cv::Mat a = computeSAD_inteOpti(p), b = computeSAD_BFOpti(p), c = censusStereo(p, BASIC_CENSUS);
    cv::Mat d = asw_gifs(p, 0.01), e = asw_gifs(p, 0.01, OURS_GIF3, 15, 0.01, 0.5), beta, fl, fr, w;
    postProcess_(p, d, e, fl, fr);
    countCensusImg(l, a); countCensusImg_circle(l, a, 2); countCensusImg_rotationInv(l, a); countCensusImg_uniform(l, a);
    countCensusImg_multiScale(l, a, 16); countCensusImg_multiScale2(l, a, 16); countHummingDist(a, b, c);
    steerFilter(l, w);
    a = getGuidedFilter(l, b, 7, 1e-6); a = getGuidedFilter_wgif(l, b, 7, 1e-6); a = getGuidedFilter_egif(l, b, 7, 1e-6, beta);
    a = getGuidedFilter_skwgif2(l, b, w, 35, 15, 0.01, 0.01, 0.5);
    a = getGuidedFilter_ours_gif(l, b, 35, 15, 0.01, 0.01, 0.5); a = getGuidedFilter_ours_gif2(l, b, 35, 15, 0.01, 0.01, 0.5);
    a = getGuidedFilter_ours_gif3(l, b, 35, 15, 0.01, 0.01, 0.5, beta);
    getDisparity_SGBM(l, r, a); getDisparity_BM(l, r, a); stereoMatching(l, r, a, mdisp::SGBM);
    mdisp_cv::StereoSGBM::create(0, 64, 7, 490, 1960, 1, 63, 10, 100, 32, mdisp_cv::StereoSGBM::MODE_SGBM)->compute(l, r, a);
    getRectifiedImages(l, r, xml, fl, fr);
    a = stereomatch_1::computeAD(l, r); a = stereomatch_1::computeSAD(l, r); a = stereomatch_1::computeSAD_inteOpti(l, r);
    a = stereomatch_1::computeSAD_BFOpti(l, r); a = stereomatch_1::computeTAD(l, r); a = stereomatch_1::computeSTAD(l, r);
    a = stereomatch_1::computeSTAD_inteOpti(l, r); a = stereomatch_1::computeSTAD_BFOpti(l, r); a = stereomatch_1::computeSD(l, r);
    a = stereomatch_1::computeSSD(l, r); a = stereomatch_1::computeSSD_inteOpti(l, r); a = stereomatch_1::computeSSD_BFOpti(l, r);
    std::vector<cv::Mat> slices;
    a = stereomatch_1::computeNCC(l, r); a = stereomatch_1::computeNCC(l, r, DISPARITY_RIGHT, 5, 2, 16);
    stereomatch_1::computeNCC(l, r, slices); stereomatch_1::computeNCC(l, r, slices, DISPARITY_LEFT, 7, 0, 30);
    a = stereomatch_1::getCostSAD_d(l, r, 3); a = stereomatch_1::getCostSAD_d(l, r, 3, DISPARITY_RIGHT, 15);
    a = stereomatch_1::getGuidedFilter(l, b, 15, 0.01);
    a = stereomatch_1::computeAdaptiveWeight_GuidedF(l, r); a = stereomatch_1::computeAdaptiveWeight_GuidedF(l, r, DISPARITY_LEFT, 0.01, 15, 10, 150);
    stereoMatching(l, r, a, mdisp::ADAPTIVE_WEIGHT_GUIDED_FILTER);
    stereomatch_1::computeSimilarity(l, r, slices, 0.4, 10, 50, DISPARITY_LEFT, 0, 30);
    stereomatch_1::computeSimilarity(l, r, slices, 0.4, 10, 50, DISPARITY_LEFT, 7, 0, 30);
    a = stereomatch_1::computeAdaptiveWeight_GuidedF_2(l, r); a = stereomatch_1::computeAdaptiveWeight_GuidedF_3(l, r, DISPARITY_RIGHT, 1e-6, 15, 0, 64);
}
