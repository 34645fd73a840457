// TEST-ONLY stand-in for the reference's 3rdparty/line_descriptor (which needs OpenCV): the KeyLine fields its headers mention
#pragma once
#include <opencv2/core.hpp>
namespace cv { namespace line_descriptor {
struct KeyLine { float angle = 0.f, response = 0.f, lineLength = 0.f, startPointX = 0.f, startPointY = 0.f, endPointX = 0.f, endPointY = 0.f; int octave = 0, class_id = -1; };
struct LSDDetectorC {};
struct BinaryDescriptor {};
} }
namespace line_descriptor = cv::line_descriptor;
