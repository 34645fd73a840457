// stvo_compat.h — the few value types the reference API spells with OpenCV / Eigen, in plain C++.
//
// The reference's signatures use cv::Mat, cv::KeyPoint, line_descriptor::KeyLine and Eigen
// fixed-size matrices (include/stereoFrame.h:24-44).  Neither library exists in this image, and
// the image-processing front-end (ORB / LSD / LBD) is out of scope, so a frame
// is handed over as its extracted FEATURES.  Names and fields follow the originals so that code
// written against the reference reads the same.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

namespace StVO {

struct Vector2d { double v[2]; double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };
struct Vector3d { double v[3]; double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };
struct Vector6d { double v[6]; double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };

struct Matrix4d {  // row-major
    double m[16];
    static Matrix4d Identity() { Matrix4d r; for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0 : 0.0; return r; }
    double& operator()(int r, int c) { return m[r * 4 + c]; }
    double operator()(int r, int c) const { return m[r * 4 + c]; }
    bool operator==(const Matrix4d& o) const { for (int i = 0; i < 16; ++i) if (!(m[i] == o.m[i])) return false; return true; }
    bool operator!=(const Matrix4d& o) const { return !(*this == o); }
};
struct Matrix6d {
    double m[36];
    static Matrix6d Identity() { Matrix6d r; for (int i = 0; i < 36; ++i) r.m[i] = (i % 7 == 0) ? 1.0 : 0.0; return r; }
    static Matrix6d Zero() { Matrix6d r; std::memset(r.m, 0, sizeof(r.m)); return r; }
    double& operator()(int r, int c) { return m[r * 6 + c]; }
    double operator()(int r, int c) const { return m[r * 6 + c]; }
};

// cv::KeyPoint: pt (float), octave
struct KeyPoint { float x, y; int octave; };
// line_descriptor::KeyLine: end points (float), angle, octave
struct KeyLine { float startPointX, startPointY, endPointX, endPointY, angle; int octave; };

// N x 32-byte descriptor matrix (cv::Mat CV_8UC1 with 32 columns)
struct DescMat {
    std::vector<uint8_t> data;
    int rows = 0;
    const uint8_t* ptr(int r = 0) const { return data.data() + (size_t)r * 32; }
    void push_back_row(const uint8_t* row) { data.insert(data.end(), row, row + 32); ++rows; }
    bool empty() const { return rows == 0; }
};

// What detectStereoPoints / detectStereoLineSegments leave behind before the stereo association
// (src/stereoFrame.cpp:88-100,191-203): the hand-over point between the out-of-scope front-end
// and the hot path.
struct FrameFeatures {
    int img_cols = 0, img_rows = 0;
    std::vector<KeyPoint> points_l, points_r;
    DescMat pdesc_l, pdesc_r;
    std::vector<KeyLine> lines_l, lines_r;
    DescMat ldesc_l, ldesc_r;
};

// cv::Mat as the image-taking entry points use it (CV_8UC1): rows x cols bytes, `step` bytes from row to row
struct GrayImage {
    const uint8_t* data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;  // 0: rows are contiguous (step == cols)
    bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }
};

}  // namespace StVO
