/*
 * ref_driver.cpp — TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" driver over the REFERENCE's own grid + Bresenham code, which is STL-only and
 * therefore builds here unmodified: /root/reference/src/gridStructure.cpp and
 * /root/reference/src/lineIterator.cpp are compiled *by path* (see oracle/Makefile, target _ref)
 * into oracle/_ref/libstvo_ref.so.  Nothing from the reference is copied into this repository;
 * the .so is git-ignored and only used by tests/ to pin oracle/stvo_oracle.c's restatement of
 * GridStructure::get / at (src/gridStructure.cpp:43-76) and LineIterator / getLineCoords
 * (src/lineIterator.cpp:34-77, src/gridStructure.cpp:33-41).
 *
 * src/matching.cpp and src/stereoFrameHandler.cpp are NOT built: they need OpenCV 3 and Eigen 3,
 * which the image lacks, and stand-in headers are not allowed.
 */
#include <algorithm>
#include <cstdint>
#include <list>
#include <unordered_set>
#include <utility>
#include <vector>

#include "gridStructure.h" /* -I/root/reference/include */

extern "C" {

/* getLineCoords(x1,y1,x2,y2) -> (x,y) cell list; returns the cell count. */
int ref_line_coords(double x1, double y1, double x2, double y2, int32_t* out_xy, int max_cells) {
    std::list<std::pair<int, int>> coords;
    StVO::getLineCoords(x1, y1, x2, y2, coords);
    int n = 0;
    for (const auto& p : coords) {
        if (n < max_cells) {
            out_xy[2 * n] = p.first;
            out_xy[2 * n + 1] = p.second;
        }
        ++n;
    }
    return n;
}

/* Fill a GridStructure(rows=48, cols=64) with grid.at(x,y).push_back(owner[k]) for every entry
 * (as stereoFrame.cpp:135-139 / :335-337 do), then run get(qx,qy,w) for each query and write
 * the candidate set, SORTED ascending, into out (query q uses out[out_off[q] .. out_off[q+1])). */
int ref_grid_get(const int32_t* ent_xy, const int32_t* owner, int n_entries, int rows, int cols, const int32_t* query_xy,
                 int n_queries, int w_lo, int w_hi, int h_lo, int h_hi, int32_t* out_off, int32_t* out, int out_cap) {
    StVO::GridStructure grid(rows, cols);
    for (int k = 0; k < n_entries; ++k) grid.at(ent_xy[2 * k], ent_xy[2 * k + 1]).push_back(owner ? owner[k] : k);
    StVO::GridWindow w;
    w.width = std::make_pair(w_lo, w_hi);
    w.height = std::make_pair(h_lo, h_hi);
    int total = 0;
    for (int q = 0; q < n_queries; ++q) {
        std::unordered_set<int> cand;
        grid.get(query_xy[2 * q], query_xy[2 * q + 1], w, cand);
        std::vector<int> v(cand.begin(), cand.end());
        std::sort(v.begin(), v.end());
        out_off[q] = total;
        for (int id : v) {
            if (total < out_cap) out[total] = id;
            ++total;
        }
    }
    out_off[n_queries] = total;
    return total;
}
}
