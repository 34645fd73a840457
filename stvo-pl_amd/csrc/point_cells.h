// point_cells.h — the grid of the key-points of ONE frame (/root/reference/src/stereoFrame.cpp:129-139; GridStructure as CSR): the
// right key-points numbered in cell order (= the matcher's scan order), their cells, the cell starts, and the left key-points
// counting-sorted by cell on a grid GRID_LW columns wide.  One workgroup of T threads per frame.
// Shared by point_cells_kernel (seq_pipeline.hip: its own launch, 256 threads) and by the one-workgroup point matcher
// (grid_kernels.hip), which runs it as its FIRST phase when every workgroup of the launch has exactly one frame — single-stream
// operation: one dependent launch and ~8 us less in the chain of a frame.
#pragma once

#include "kernels.h"

namespace stvo {

// (member order and alignment are measured: with hist first and 4-byte alignment the compiler gave up the 16-byte LDS accesses of
// the scans and the 1024-frame launch took 65 instead of 45 us)
template <int T>
struct alignas(16) PointCellsLds {
    int lhist[GRID_LCELLS];
    int fill[STVO_GRID_CELLS];
    int hist[STVO_GRID_CELLS];
    int wave[T / 64];
    int extra;
};

// exclusive scan of hist[0..N) in LDS by T threads; writes start_out[0..N], leaves the starts in hist and returns the total
template <int T, int N>
__device__ __forceinline__ int point_cells_scan(int* hist, int* s_wave, int32_t* start_out) {
    constexpr int PER = (N + T - 1) / T;
    constexpr bool FULL = N % T == 0;  // every thread has PER cells: unguarded accesses, which the compiler merges into 16-byte ones
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int local[PER];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = tid * PER + k;
        local[k] = (FULL || c < N) ? hist[c] : 0;
        sum += local[k];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) {
        const int c = s_wave[w];
        if (w < wv) base += c;
        total += c;
    }
    int run = base + incl - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = tid * PER + k;
        if (FULL || c < N) {
            hist[c] = run;
            start_out[c] = run;
        }
        run += local[k];
    }
    if (tid == T - 1) start_out[N] = run;
    __syncthreads();
    return total;
}

__device__ __forceinline__ bool point_in_grid(int x, int y) {
    return x >= 0 && x < STVO_GRID_COLS && y >= 0 && y < STVO_GRID_ROWS;
}

// LEAN (the one-workgroup point matcher will run, grid_points_fused_ok): only what that matcher reads — the cell-sorted left
// indices and their starts, the scan order of the right key-points and their cells, the CSR starts.  The integer cells and
// candidate ranges of the left key-points, the CSR items / ranks and the empty top-2 records (64 of 125 KB per frame) are inputs
// of the scan formulation only; the matcher rebuilds them for the rare frame it hands to it (fused_misfit_frame).
template <int T, bool LEAN>
__device__ __forceinline__ void point_cells_frame(const PointCells& s, const int b, PointCellsLds<T>* lds) {
    constexpr int LPT = (2048 + T - 1) / T;  // left key-points per thread of the cell sort (host: K <= 2048)
    int* hist = lds->hist;
    int* fill = lds->fill;
    int* lhist = lds->lhist;
    const int tid = threadIdx.x;
    const bool lsort = s.plperm != nullptr;  // host: K <= 2048, window within GRID_LW
    const int nl = s.n_kp_l[b], nr = s.n_kp_r[b];
    const size_t off = (size_t)b * s.K;
    // the key-points of this thread (i = tid + T k), left and right: requested before anything else, together with the counts
    float2 rxy[LPT], lxy[LPT];
#pragma unroll
    for (int k = 0; k < LPT; ++k) rxy[k] = reinterpret_cast<const float2*>(s.kp_r)[off + min(tid + T * k, s.K - 1)];
#pragma unroll
    for (int k = 0; k < LPT; ++k) lxy[k] = reinterpret_cast<const float2*>(s.kp_l)[off + min(tid + T * k, s.K - 1)];  // (also without lsort: no branch)
    __builtin_amdgcn_sched_barrier(0);  // (the conversions of the first coordinates were scheduled between the loads, with a wait)
    const double inv_w = s.inv_wh[2 * b], inv_h = s.inv_wh[2 * b + 1];
    if (!LEAN)
        for (int i = tid; i < nl; i += T) {  // float * double -> int truncation (stereoFrame.cpp:132)
            s.pxy_l[(off + i) * 2 + 0] = (int)((double)s.kp_l[(off + i) * 2 + 0] * inv_w);
            s.pxy_l[(off + i) * 2 + 1] = (int)((double)s.kp_l[(off + i) * 2 + 1] * inv_h);
        }
    for (int c = tid; c < STVO_GRID_CELLS; c += T) {
        hist[c] = 0;
        fill[c] = 0;
    }
    for (int c = tid; c < GRID_LCELLS; c += T) lhist[c] = 0;
    if (!LEAN)
        for (int i = tid; i < s.K; i += T) s.top2_p[off + i] = 0x00000000FFFFFFFFull;  // grid matcher: no eligible candidate yet
    if (tid == 0) {
        if (!LEAN) s.govf_p[b] = 0;
        lds->extra = 0;
    }
    // The cells of the right key-points of this thread, kept for the scatter below; the coordinates were requested all at once above: as loops `for (i = tid; i < nr; i += T)` the histogram and the scatter each were a chain of load -> wait -> LDS
    // atomic per trip (tools/isa_scan.py --waits: every load waited for at once; eight trips each with 256 threads).  Rows past nr
    // are read from a clamped index (inside the frame's slot) and ignored.
    int rcel[LPT];  // cell, -1: outside the grid (the reference's out_of_bounds sink), -2: no key-point
    {
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int x = (int)((double)rxy[k].x * inv_w);
            const int y = (int)((double)rxy[k].y * inv_h);
            rcel[k] = tid + T * k < nr ? (point_in_grid(x, y) ? y * STVO_GRID_COLS + x : -1) : -2;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LPT; ++k)
        if (rcel[k] >= 0) atomicAdd(&hist[rcel[k]], 1);
    for (int i = tid + T * LPT; i < nr; i += T) {  // (capacities beyond 2048 key-points per image: trip by trip)
        const int x = (int)((double)s.kp_r[(off + i) * 2 + 0] * inv_w);
        const int y = (int)((double)s.kp_r[(off + i) * 2 + 1] * inv_h);
        if (point_in_grid(x, y)) atomicAdd(&hist[y * STVO_GRID_COLS + x], 1);
    }
    // counting sort of the LEFT key-points by cell, for the matcher that walks the candidates of a right key-point: the window
    // is clamped, not the cell (src/gridStructure.cpp:67-71), so columns up to 63 + ws still see the last grid columns
    int lcel[LPT], lrnk[LPT];
    if (lsort) {
        // (coordinates first, from a clamped index, as for the right key-points: a load under `if (i < nl)` stayed in its branch and
        //  was waited for there — eight memory round trips one after the other)
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int x = (int)((double)lxy[k].x * inv_w);
            const int y = (int)((double)lxy[k].y * inv_h);
            lcel[k] = (tid + T * k < nl && y >= 0 && y < STVO_GRID_ROWS && x >= 0 && x <= STVO_GRID_COLS - 1 + s.ws) ? y * GRID_LW + x : -1;
        }
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            lrnk[k] = 0;
            if (lcel[k] >= 0) lrnk[k] = atomicAdd(&lhist[lcel[k]], 1);
        }
    }
    __syncthreads();
    const int n_in = point_cells_scan<T, STVO_GRID_CELLS>(hist, lds->wave, s.pstart + (size_t)b * (STVO_GRID_CELLS + 1));
    if (lsort) {
        point_cells_scan<T, GRID_LCELLS>(lhist, lds->wave, reinterpret_cast<int32_t*>(s.plstart) + (size_t)b * GRID_LSTART_STRIDE);
#pragma unroll
        for (int k = 0; k < LPT; ++k)
            if (lcel[k] >= 0) s.plperm[off + lhist[lcel[k]] + lrnk[k]] = tid + T * k;
    }
    // GridStructure::get with the stereo window (matching_s_ws cells to the left, same row; src/gridStructure.cpp:65-76,
    // src/stereoFrame.cpp:141-143): cells x - ws .. x of row y are contiguous in the CSR => candidates = positions [lo, hi)
    if (!LEAN)
        for (int i = tid; i < nl; i += T) {
            const int x = s.pxy_l[(off + i) * 2 + 0], y = s.pxy_l[(off + i) * 2 + 1];
            int lo = 0, hi = 0;
            if (y >= 0 && y < STVO_GRID_ROWS) {
                const int min_x = min(max(0, x - s.ws), STVO_GRID_COLS), max_x = max(min(STVO_GRID_COLS, x + 1), min_x);
                const int c0 = y * STVO_GRID_COLS + min_x, c1 = y * STVO_GRID_COLS + max_x;
                lo = c0 < STVO_GRID_CELLS ? hist[c0] : n_in;
                hi = c1 < STVO_GRID_CELLS ? hist[c1] : n_in;
            }
            s.prange[(off + i) * 2 + 0] = lo;
            s.prange[(off + i) * 2 + 1] = hi;
        }
    __syncthreads();  // hist (the cell starts) is read above and advanced by nobody: fill[] takes the scatter counters
#pragma unroll
    for (int k = 0; k < LPT; ++k) {
        const int i = tid + T * k, c = rcel[k];
        if (c == -2) continue;
        int pos;
        if (c >= 0) {
            pos = hist[c] + atomicAdd(&fill[c], 1);
            if (!LEAN) s.pitems[off + pos] = i;
        } else {  // the reference's out_of_bounds sink: never a candidate, scanned last
            pos = n_in + atomicAdd(&lds->extra, 1);
        }
        s.pperm[off + pos] = i;  // scan order = CSR (spatial) order
        s.pcell[off + pos] = c;
        if (!LEAN) s.prank[off + i] = pos;
    }
    for (int i = tid + T * LPT; i < nr; i += T) {
        const int x = (int)((double)s.kp_r[(off + i) * 2 + 0] * inv_w);
        const int y = (int)((double)s.kp_r[(off + i) * 2 + 1] * inv_h);
        const int c = point_in_grid(x, y) ? y * STVO_GRID_COLS + x : -1;
        int pos;
        if (c >= 0) {
            pos = hist[c] + atomicAdd(&fill[c], 1);
            if (!LEAN) s.pitems[off + pos] = i;
        } else {
            pos = n_in + atomicAdd(&lds->extra, 1);
        }
        s.pperm[off + pos] = i;
        s.pcell[off + pos] = c;
        if (!LEAN) s.prank[off + i] = pos;
    }
}

}  // namespace stvo
