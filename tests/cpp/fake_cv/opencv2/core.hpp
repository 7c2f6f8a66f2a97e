// Test double, NOT OpenCV: declares just the part of <opencv2/core.hpp> that include/mdisp_cv.hpp touches, so that the
// adapter can be compiled (never run) on a machine without OpenCV.  Nothing here is defined; linking is not intended.
#ifndef MDISP_FAKE_OPENCV_CORE_HPP
#define MDISP_FAKE_OPENCV_CORE_HPP
#include <cstddef>
#include <string>

#define CV_8U 0
#define CV_16S 3
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))

inline int cvRound(double v) { return (int)(v + (v >= 0 ? 0.5 : -0.5)); }

namespace cv {
struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
    int area() const { return width * height; }
};
struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() {}
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};
Rect operator&(const Rect& a, const Rect& b);

struct MatStep {
    size_t buf[2];
    size_t operator[](int i) const { return buf[i]; }
};
class Mat {
public:
    Mat();
    Mat(Size size, int type);
    Mat(const Mat&);
    Mat& operator=(const Mat&);
    ~Mat();
    bool empty() const;
    int channels() const;
    int depth() const;
    int type() const;
    size_t total() const;
    void create(int rows, int cols, int type);
    Mat reshape(int cn, int rows = 0) const;
    void convertTo(Mat& m, int rtype) const;
    Mat clone() const;
    Mat operator()(const Rect& roi) const;
    template <typename T> T* ptr(int row = 0);
    template <typename T> const T* ptr(int row = 0) const;
    unsigned char* data;
    int rows, cols;
    MatStep step;
};

class FileNode {
public:
    void operator>>(Mat& m) const;
    void operator>>(Size& s) const;
};
class FileStorage {
public:
    enum { READ = 0 };
    FileStorage(const std::string& path, int flags);
    ~FileStorage();
    FileNode operator[](const char* key) const;
    void release();
};
}  // namespace cv
#endif
