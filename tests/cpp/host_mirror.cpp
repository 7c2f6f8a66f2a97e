// Exercises the C++ host layer (include/mdisp_host.hpp) with its built-in image type.
//   host_mirror              : error behaviour only (no GPU needed)
//   host_mirror gpu <H> <W> <maxD> <out.bin> : runs censusStereo, SAD and SGBM on a synthetic pair
//                              and dumps the outputs for comparison against the oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mdisp_host.hpp"

using namespace mdisp;

static int fail(const char* what) { std::printf("FAIL: %s\n", what); return 1; }

int main(int argc, char** argv)
{
    // --- reference error behaviour: message + empty image, nothing thrown
    {
        StereoMatchParam p;          // empty images
        Mat d;
        try { d = computeSAD_BFOpti(p); } catch (...) { return fail("empty input must not throw"); }
        if (!d.empty()) return fail("empty input must give an empty image");
        d = censusStereo(p, BASIC_CENSUS);
        if (!d.empty()) return fail("censusStereo on empty input must give an empty image");
    }
    {   // stereomatch_1: even window -> "winsize must be odd" + empty image (method_StereoMatching.cpp:99-103)
        Mat a(8, 8, MDISP_8U), b(8, 8, MDISP_8U);
        if (!stereomatch_1::computeSAD(a, b, 6).empty()) return fail("even winSize must give an empty image (stereomatch_1)");
    }
    if (argc < 6 || std::strcmp(argv[1], "gpu") != 0) { std::printf("OK (no-GPU checks)\n"); return 0; }

    const int H = std::atoi(argv[2]), W = std::atoi(argv[3]), maxD = std::atoi(argv[4]);
    Mat L(H, W, MDISP_8U), R(H, W, MDISP_8U);
    if (mdisp_synth_pair(0x5EED0001u, H, W, maxD + 1, L.data(), R.data(), nullptr)) return fail("synth");
    StereoMatchParam p;
    p.imgLeft = L; p.imgRight = R; p.winSize = 7; p.minDisparity = 0; p.maxDisparity = maxD; p.isDispLeft = true;
    {   // even window: message + empty image
        StereoMatchParam q = p; q.winSize = 6;
        if (!computeSAD_inteOpti(q).empty()) return fail("even winSize must give an empty image");
    }
    Mat dCensus = censusStereo(p, BASIC_CENSUS, MDISP_AGG_BOX);   // CV_64F
    Mat dSad = computeSAD_BFOpti(p);                               // CV_8U
    Mat d16;
    mdisp_sgbm_params sp = {0, maxD + 1, 8, 32, 10, 1, 8, 1, 1, MDISP_BASIC_CENSUS};
    sgbmCompute(L, R, sp, d16);                                    // CV_16S
    Mat cen;
    countCensusImg(L, cen);
    // the call the reference's main() makes (cameraCalibration.cpp:174), with census costs in place of dense SIFT
    Mat dOurs = asw_gifs(p, 0.01, OURS_GIF3, 15, 0.01, 0.5);       // CV_32F; prints the cost-source note (first call, source never chosen)
    {
        Mat again = asw_gifs(p, 0.01, OURS_GIF3, 15, 0.01, 0.5);   // no second note
        set_asw_cost_source(MDISP_COST_CENSUS);                    // explicit choice: same slices, and no note from now on
        Mat chosen = asw_gifs(p, 0.01, OURS_GIF3, 15, 0.01, 0.5);
        if (again.empty() || chosen.empty() || std::memcmp(again.data(), dOurs.data(), (size_t)H * W * 4) ||
            std::memcmp(chosen.data(), dOurs.data(), (size_t)H * W * 4))
            return fail("asw_gifs: repeated calls differ");
    }
    if (dCensus.empty() || dSad.empty() || d16.empty() || cen.empty() || dOurs.empty()) return fail("empty output");
    FILE* f = std::fopen(argv[5], "wb");
    if (!f) return fail("open output");
    std::fwrite(L.data(), 1, (size_t)H * W, f);
    std::fwrite(R.data(), 1, (size_t)H * W, f);
    std::fwrite(dCensus.data(), 8, (size_t)H * W, f);
    std::fwrite(dSad.data(), 1, (size_t)H * W, f);
    std::fwrite(d16.data(), 2, (size_t)H * W, f);
    std::fwrite(cen.data(), 1, (size_t)H * W, f);
    std::fwrite(dOurs.data(), 4, (size_t)H * W, f);
    Mat dBt, dG;
    StereoSGBM::create(0, 32, 7, 490, 1960, 1, 63, 10, 100, 32, StereoSGBM::MODE_SGBM)->compute(L, R, dBt);   // methods.cpp:1216-1219 shape
    getDisparity_SGBM(L, R, dG);                                                                              // methods.cpp:1883-1921
    if (dBt.empty() || dG.empty()) return fail("empty StereoSGBM output");
    std::fwrite(dBt.data(), 2, (size_t)H * W, f);
    std::fwrite(dG.data(), 2, (size_t)H * W, f);
    Mat dLeg = stereomatch_1::computeSSD_BFOpti(L, R, 7, 0, maxD + 1);                                        // method_StereoMatching.h:112
    if (dLeg.empty()) return fail("empty stereomatch_1 output");
    std::fwrite(dLeg.data(), 1, (size_t)H * W, f);
    Mat dNcc = stereomatch_1::computeNCC(L, R, stereomatch_1::DISPARITY_LEFT, 5, 0, 8);                       // method_StereoMatching.h:116-117
    std::vector<Mat> nccSlices;
    stereomatch_1::computeNCC(L, R, nccSlices, stereomatch_1::DISPARITY_RIGHT, 5, 1, 4);                      // method_StereoMatching.h:118-120
    if (dNcc.empty() || nccSlices.size() != 4 || !stereomatch_1::computeNCC(L, R, stereomatch_1::DISPARITY_LEFT, 4).empty())
        return fail("computeNCC: empty output, wrong slice count, or an even window accepted");
    // cv::StereoBM the way methods.cpp:1829-1838 drives it (smaller range and window), and the stereoMatching() dispatcher
    Mat dBm, dDispatch;
    StereoBM sbm = StereoBM::create(16, 9);
    sbm->setMinDisparity(2);
    sbm->setTextureThreshold(10);
    sbm->setUniquenessRatio(25);
    sbm->compute(L, R, dBm);
    stereoMatching(L, R, dDispatch, SGBM);                                                                    // methods.cpp:1776-1813
    if (dBm.empty() || dDispatch.empty()) return fail("empty StereoBM / stereoMatching output");
    std::fwrite(dBm.data(), 2, (size_t)H * W, f);
    std::fwrite(dDispatch.data(), 2, (size_t)H * W, f);
    // the post-filter main() runs on the two maps (cameraCalibration.cpp:208); here the left map serves as both
    Mat fL, fR;
    postProcess_(p, dOurs, dOurs, fL, fR);
    if (fL.empty() || fR.empty()) return fail("empty postProcess_ output");
    std::fwrite(fL.data(), 4, (size_t)H * W, f);
    std::fwrite(fR.data(), 4, (size_t)H * W, f);
    // the guided-filter matcher on a colour pair (channel c = gray value scaled by (4 - c) / 4), method_StereoMatching.h:179-181,
    // and one of its cost slices (:171) on a bordered right view
    Mat Lc(H, W, MDISP_8U, 3), Rc(H, W, MDISP_8U, 3), Rb(H, W + 8, MDISP_8U, 1);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 3; c++) {
                Lc.ptr<uint8_t>(y)[x * 3 + c] = (uint8_t)(L.at<uint8_t>(y, x) * (4 - c) / 4);
                Rc.ptr<uint8_t>(y)[x * 3 + c] = (uint8_t)(R.at<uint8_t>(y, x) * (4 - c) / 4);
            }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W + 8; x++) Rb.at<uint8_t>(y, x) = R.at<uint8_t>(y, x < 8 ? 7 - x : x - 8);       // BORDER_REFLECT on the left
    Mat dGf = stereomatch_1::computeAdaptiveWeight_GuidedF(Lc, Rc, stereomatch_1::DISPARITY_LEFT, 0.01, 7, 0, 12);
    Mat cSad = stereomatch_1::getCostSAD_d(L, Rb, 3, stereomatch_1::DISPARITY_LEFT, 5);
    if (dGf.empty() || cSad.empty() || !stereomatch_1::getCostSAD_d(L, R, 3, stereomatch_1::DISPARITY_LEFT, 5).empty())
        return fail("computeAdaptiveWeight_GuidedF / getCostSAD_d: empty output, or an unbordered view accepted");
    std::fwrite(dGf.data(), 4, (size_t)H * W, f);
    std::fwrite(cSad.data(), 4, (size_t)H * W, f);
    // the colour + gradient cost slices (method_StereoMatching.h:123-125) and the matcher that filters them (:182-184)
    std::vector<Mat> sim;
    stereomatch_1::computeSimilarity(Lc, Rc, sim, 0.4, 10, 50, stereomatch_1::DISPARITY_LEFT, 0, 6);
    Mat dGf2 = stereomatch_1::computeAdaptiveWeight_GuidedF_2(Lc, Rc, stereomatch_1::DISPARITY_LEFT, 0.01, 7, 0, 12);
    if (sim.size() != 6 || dGf2.empty()) return fail("computeSimilarity / computeAdaptiveWeight_GuidedF_2: wrong slice count or empty output");
    std::fwrite(sim[3].data(), 4, (size_t)H * W, f);
    std::fwrite(dGf2.data(), 4, (size_t)H * W, f);
    std::fclose(f);
    std::printf("OK\n");
    return 0;
}
