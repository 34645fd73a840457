// debug_switches.h — every developer switch of the library in ONE place.  The STVO_* environment variables are parsed once, on
// first use, into this struct; nothing else in csrc/ calls getenv.  stvo_debug_reparse_env() (C-ABI, tests and tools only) reads
// them again, so that a parity test can drive several variants from one process.  DBG_UNSET = not set: the library's own choice.
#pragma once

#include <climits>

namespace stvo {

constexpr int DBG_UNSET = INT_MIN;

struct DebugSwitches {
    int pose_kernel;     // STVO_POSE_KERNEL     1: pose_kernel.hip for every batch size, 4: pose_kernel2p.hip for every batch size
    int pose2p_nw;       // STVO_POSE2P_NW       waves per frame pair of the batch kernel (2 or 4)
    int pose_prof;       // STVO_POSE_PROF       in-kernel phase ticks (tools/pose_probe.py)
    int pose_lds_t;      // STVO_POSE_LDS_T      0: pose_kernel.hip's throughput variant without its partial LDS record cache
    int pose_los;        // STVO_POSE_LOS        0: the key-lines of pose_kernel.hip always with the worker waves (never the solver wave)
    int knn_mfma;        // STVO_KNN_MFMA        0: VALU matcher (K1 + K1v), else query blocks per wave of K1m
    int knn_nseg;        // STVO_KNN_NSEG        train segments per query tile
    int seq_graph;       // STVO_SEQ_GRAPH       1: hipGraph replay of the per-frame chain
    int seq_prof;        // STVO_SEQ_PROF        host-side phase times of stvo_seq_push
    int seq_inline;      // STVO_SEQ_INLINE      0: events between the streams of a step for small batches too (kernels.h: PoseArgs::wait_flag)
    int line_fork_late;  // STVO_LINE_FORK       start (0) / late (1) / mid (2): where the key-line stream forks off — at the start of the step, after the stereo point stage, behind the cells kernel (unset: mid for batches of >= 64, start below)
    int line_first;      // STVO_LINE_FIRST      1: key-line kernels enqueued before the point-cells kernel
    int line_fused;      // STVO_LINE_FUSED      0 / 1: general / one-workgroup stereo line matcher
    int match_small;     // STVO_MATCH_SMALL     0: the general f2f machinery for the key-line sets too
    int match_lazy;      // STVO_MATCH_LAZY      1: the lazy reverse check for small batches too
    int grid_tail;       // STVO_GRID_TAIL       0: point_tail_kernel as its own launch
    int grid_fused;      // STVO_GRID_FUSED      0: scan formulation of the stereo point matcher
    int grid_fused_cap;  // STVO_GRID_FUSED_CAP  capacity override of the one-workgroup point matcher (tests of the misfit path)
    int lsd_grow;        // STVO_LSD_GROW        0: the plain form of lsd_grow_kernel (candidates one after the other, sums through v_readlane)
    int lsd_waves;       // STVO_LSD_WAVES       0: batches of <= 8 images by lsd_grow_kernel (one wave per image) instead of lsd_grow_xcd_kernel (a committer + speculating workgroups on the CUs of one XCD per image)
    int lsd_xcd_blocks;  // STVO_LSD_XCD_BLOCKS  speculating workgroups (of four waves) per image of lsd_grow_xcd_kernel (unset: 16, or what the XCD's 32 CUs leave per image)
    int lsd_feed_ahead;  // STVO_LSD_FEED_AHEAD  ranks the feeder wave of lsd_grow_xcd_kernel runs ahead of the committer at most
    int lsd_sep;         // STVO_LSD_SEP         least distance (pixels, Chebyshev) of a new seed from every seed in flight
    int lsd_multi;       // STVO_LSD_MULTI       0: the committer of lsd_grow_xcd_kernel takes its seeds one by one (unset: up to four records per pass)
    int lsd_ahead;       // STVO_LSD_AHEAD       ranks the dispatcher's front runs ahead of the committer at most
    int cells_ahead;     // STVO_CELLS_AHEAD     0: point_cells_kernel of a batch in the point stream (unset: on the line stream, ahead of the point stream's step)
    int seq_pipe;        // STVO_SEQ_PIPE        1: pipelined steps (optimizePose(k) on the aux stream beside the stereo association of step k + 1; built and measured in round 6, no gain), 2: the same without the gate kernel
    int lines_ahead;     // STVO_LINES_AHEAD     0: the key-line stream waits for its own step's fork event (until round 5); unset: for batches it runs one step ahead, behind the dispatch of the previous pose kernel
    int grid_dyn;        // STVO_GRID_DYN        0: the persistent point matcher takes its frames by a static stride (unset: from a counter, when the step is pipelined)
    int grid_cells;      // STVO_GRID_CELLS      0: point_cells_kernel as its own launch for small batches too, 1: in the matcher whenever it fits
};

const DebugSwitches& dbg();  // parsed on first use (stvo_capi.hip)
void dbg_reparse();

}  // namespace stvo
