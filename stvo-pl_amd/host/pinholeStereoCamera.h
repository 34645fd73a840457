// pinholeStereoCamera.h — the camera scalars and the two functions the path uses
// (src/pinholeStereoCamera.cpp:221-237, include/pinholeStereoCamera.h:75-90).  The YAML / rectification
// constructor is input preparation and out of scope; parameters are given directly.
#pragma once
#include "../../include/stvo_types.h"
#include "stvo_compat.h"

namespace StVO {

class PinholeStereoCamera {
public:
    PinholeStereoCamera(int width_, int height_, double fx_, double fy_, double cx_, double cy_, double b_)
        : width(width_), height(height_), fx(fx_), fy(fy_), cx(cx_), cy(cy_), b(b_) {}
    int getWidth() const { return width; }
    int getHeight() const { return height; }
    double getB() const { return b; }
    double getFx() const { return fx; }
    double getFy() const { return fy; }
    double getCx() const { return cx; }
    double getCy() const { return cy; }
    Vector3d backProjection(const double& u, const double& v, const double& disp) const {
        Vector3d P;
        const double bd = b / disp;
        P(0) = bd * (u - cx);
        P(1) = bd * (v - cy);
        P(2) = bd * fx;
        return P;
    }
    Vector2d projection(const Vector3d& P) const {
        Vector2d uv;
        uv(0) = cx + fx * P(0) / P(2);
        uv(1) = cy + fy * P(1) / P(2);
        return uv;
    }
    stvo_cam abi() const { return stvo_cam{fx, fy, cx, cy, b}; }

private:
    int width, height;
    double fx, fy, cx, cy, b;
};

}  // namespace StVO
