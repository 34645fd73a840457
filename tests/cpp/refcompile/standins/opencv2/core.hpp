// TEST-ONLY stand-in: declarations of the few OpenCV names the reference's HEADERS and include/stvo_reference_overloads.h
// mention, so that the drop-in header can meet a compiler (tests/test_reference_overloads_compile.py, -fsyntax-only).  Nothing
// here is an implementation, nothing here is used for parity, nothing here ships: with the real OpenCV on the include path this
// directory is simply not given to the compiler.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>
#define CV_8UC1 0
namespace cv {
struct Size { int width = 0, height = 0; };
struct Point2f { float x = 0.f, y = 0.f; };
struct Mat {
    int rows = 0, cols = 0;
    bool isContinuous() const;
    int type() const;
    bool empty() const;
    Mat row(int) const;
    template <typename T> const T* ptr(int r = 0) const;
    template <typename T> T* ptr(int r = 0);
    template <typename T> T& at(int, int);
};
struct KeyPoint { Point2f pt; float size = 0.f, angle = 0.f, response = 0.f; int octave = 0, class_id = -1; };
struct DMatch { int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 0.f; };
template <typename T> using Ptr = std::shared_ptr<T>;
struct Vec4f { float v[4]; float operator()(int i) const { return v[i]; } };
struct BFMatcher {};
struct ORB {};
}  // namespace cv
