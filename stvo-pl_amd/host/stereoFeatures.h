// stereoFeatures.h — PointFeature / LineFeature records with the reference's field names
// (include/stereoFeatures.h:30-121) and constructor behaviour (src/stereoFeatures.cpp).
#pragma once
#include "config.h"
#include "stvo_compat.h"

namespace StVO {

class PointFeature {
public:
    // stereo-association constructor (src/stereoFeatures.cpp:41-47): sigma2 = 1 / scale^(2 level)
    PointFeature(Vector2d pl_, double disp_, Vector3d P_, int idx_, int level_)
        : idx(idx_), pl(pl_), disp(disp_), P(P_), inlier(true), level(level_) {
        for (int i = 0; i < level; i++) sigma2 *= Config::orbScaleFactor();
        sigma2 = 1.0 / (sigma2 * sigma2);
    }
    // full constructor used by safeCopy (:57-64): sigma2 taken as is; idx is NOT copied upstream
    PointFeature(Vector2d pl_, double disp_, Vector3d P_, Vector2d pl_obs_, int /*idx_*/, int level_, double sigma2_,
                 bool inlier_)
        : idx(-1), pl(pl_), pl_obs(pl_obs_), disp(disp_), P(P_), inlier(inlier_), level(level_), sigma2(sigma2_) {}
    PointFeature* safeCopy() { return new PointFeature(pl, disp, P, pl_obs, idx, level, sigma2, inlier); }

    int idx;
    Vector2d pl, pl_obs{};
    double disp;
    Vector3d P;
    bool inlier;
    int level;
    double sigma2 = 1.0;
};

class LineFeature {
public:
    // stereo-association constructor (:107-115)
    LineFeature(Vector2d spl_, double sdisp_, Vector3d sP_, Vector2d epl_, double edisp_, Vector3d eP_, Vector3d le_,
                double angle_, int idx_, int level_)
        : idx(idx_), spl(spl_), epl(epl_), sdisp(sdisp_), edisp(edisp_), angle(angle_), sP(sP_), eP(eP_), le(le_),
          inlier(true), level(level_) {
        for (int i = 0; i < level; i++) sigma2 *= Config::lsdScale();
        sigma2 = 1.0 / (sigma2 * sigma2);
    }
    // full constructor used by safeCopy (:117-129): RE-APPLIES the level scaling to sigma2
    LineFeature(Vector2d spl_, double sdisp_, Vector3d sP_, Vector2d spl_obs_, double sdisp_obs_, Vector2d epl_,
                double edisp_, Vector3d eP_, Vector2d epl_obs_, double edisp_obs_, Vector3d le_, Vector3d le_obs_,
                double angle_, int idx_, int level_, bool inlier_, double sigma2_)
        : idx(idx_), spl(spl_), epl(epl_), spl_obs(spl_obs_), epl_obs(epl_obs_), sdisp(sdisp_), edisp(edisp_),
          angle(angle_), sdisp_obs(sdisp_obs_), edisp_obs(edisp_obs_), sP(sP_), eP(eP_), le(le_), le_obs(le_obs_),
          inlier(inlier_), level(level_), sigma2(sigma2_) {
        for (int i = 0; i < level; i++) sigma2 *= Config::lsdScale();
        sigma2 = 1.0 / (sigma2 * sigma2);
    }
    LineFeature* safeCopy() {
        return new LineFeature(spl, sdisp, sP, spl_obs, sdisp_obs, epl, edisp, eP, epl_obs, edisp_obs, le, le_obs, angle,
                               idx, level, inlier, sigma2);
    }

    int idx;
    Vector2d spl, epl, spl_obs{}, epl_obs{};
    double sdisp, edisp, angle, sdisp_obs = 0.0, edisp_obs = 0.0;
    Vector3d sP, eP;
    Vector3d le, le_obs{};
    bool inlier;
    int level;
    double sigma2 = 1.0;
};

}  // namespace StVO
