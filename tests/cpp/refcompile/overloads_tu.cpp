// Syntax / type check of include/stvo_reference_overloads.h against the reference's REAL headers (include/matching.h,
// include/stereoFrameHandler.h and everything they pull in from /root/reference/include) with the test-only declarations of
// standins/ in place of OpenCV / Eigen / line_descriptor.  Compiled with -fsyntax-only by tests/test_reference_overloads_compile.py.
#include <matching.h>
#include <stereoFrameHandler.h>

#include <stvo_reference_overloads.h>

#ifndef STVO_HAVE_REFERENCE_TYPES
#error "stvo_reference_overloads.h did not see the reference's types: its __has_include guard stayed closed"
#endif

#include <type_traits>

// the overloads have EXACTLY the signatures of the functions they replace (include/matching.h:50-60)
static_assert(std::is_same<decltype(&StVO::matchNNR), decltype(&StVO::hip::matchNNR)>::value, "matchNNR");
static_assert(std::is_same<decltype(&StVO::match), decltype(&StVO::hip::match)>::value, "match");
typedef int (*grid_points_fn)(const std::vector<StVO::point_2d>&, const cv::Mat&, const StVO::GridStructure&, const cv::Mat&, const StVO::GridWindow&,
                              std::vector<int>&);
typedef int (*grid_lines_fn)(const std::vector<StVO::line_2d>&, const cv::Mat&, const StVO::GridStructure&, const cv::Mat&,
                             const std::vector<std::pair<double, double>>&, const StVO::GridWindow&, std::vector<int>&);
static grid_points_fn ref_points = &StVO::matchGrid, hip_points = &StVO::hip::matchGrid;
static grid_lines_fn ref_lines = &StVO::matchGrid, hip_lines = &StVO::hip::matchGrid;

// the optimizePose body compiles against the reference's StereoFrameHandler / StereoFrame / PointFeature / LineFeature / Config
void drop_in(StVO::StereoFrameHandler& h, PinholeStereoCamera* cam, stvo_ctx* ctx) {
    StVO::hip::set_context(ctx);
    StVO::hip::optimizePose(h, cam);
    StVO::hip::optimizePose(h, cam, 2);
    (void)ref_points; (void)hip_points; (void)ref_lines; (void)hip_lines;
}
