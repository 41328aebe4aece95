// tests/host_emul/lap_box_host.hip -- TEST INFRASTRUCTURE.  The exact-tie step with the TRACKER's cost source (csrc/lap_src.h
// BoxSrc: non-edges priced by csrc/iou_math.h's iou::cost from the stage's boxes) on the CPU -- the code k_lap_exact runs on
// the device, compiled by hipcc for the host (no HIP API is called, no GPU needed).  tests/test_lap_exact_host.py feeds it
// the oracle's boxes and feasible edges of real tracker stages and expects the oracle's dense JV, index for index.
#include "../../bot-sort-onnx-tensorrt_amd/csrc/lap_exact.h"
#include "../../bot-sort-onnx-tensorrt_amd/csrc/lap_src.h"
#include "host_ex.h"
#include <string.h>
#include <vector>

// the stage's IoU cost matrix from the product's own arithmetic (cross-check of iou::cost against the oracle's restatement)
extern "C" void lapx_host_box_matrix(int N, int M, const double *abox, const uint8_t *af32, const double *bbox, const int32_t *bidx, double *out) {
    const BoxSrc S{abox, af32, bbox, bidx};
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < M; ++j) out[(size_t)i * M + j] = S.real(i, j);
}

// e_* [cap]: feasible edges at [0, E), overrides at [o_base, o_base + O).  info: {tied, err, replayed}
extern "C" int lapx_host_exact_box(int N, int M, const double *abox, const uint8_t *af32, const double *bbox, const int32_t *bidx,
                                   const int32_t *e_row, const int32_t *e_col, const double *e_cost, int E, int O, long o_base, double L,
                                   int32_t *x, int32_t *y, int nth, int order, int force_replay, int32_t *info) {
    lapx::Work W;
    memset(&W, 0, sizeof(W));
    std::vector<double> dbl(lapx::work_doubles(N, M, (long)E + O, nth));
    std::vector<int32_t> ints(lapx::work_ints(N, M, (long)E + O, nth));
    if (!lapx::work_carve(W, N, M, (long)E + O, nth, dbl.data(), dbl.size(), ints.data(), ints.size())) return -1;
    W.N = N; W.M = M; W.L = L; W.e_row = e_row; W.e_col = e_col; W.e_cost = e_cost; W.E = E; W.O = O; W.o_base = o_base;
    W.x = x; W.y = y; W.max_steps = 50000000L;
    HostEx ex{nth, order};
    lapx::build_csr(ex, W);
    lapx::tie_check(ex, W);
    info[0] = W.sh[lapx::S_TIED]; info[2] = 0;
    if (info[0] || force_replay) { const BoxSrc S{abox, af32, bbox, bidx}; lapx::jv_replay(ex, W, S); info[2] = 1; }
    info[1] = W.sh[lapx::S_ERR];
    return 0;
}

// ---- the conservative overlap screening of the cost / duplicate tiles (iou::screen_box / screen_overlap) on the CPU -------
// out[i * m + j] = 1 iff the screening says "these two boxes may overlap" (a 0 makes the tile skip the pair's exact IoU)
extern "C" void iou_host_screen(int n, int m, const double *a, const double *b, uint8_t *out) {
    for (int i = 0; i < n; ++i) {
        const iou::ScreenBox sa = iou::screen_box(a + 4 * i);
        for (int j = 0; j < m; ++j) out[(size_t)i * m + j] = iou::screen_overlap(sa, iou::screen_box(b + 4 * j)) ? 1 : 0;
    }
}
extern "C" int iou_host_screen_none(int m, const double *b) {      // screen_none() overlaps nothing: returns how many it "overlaps"
    int hits = 0;
    for (int j = 0; j < m; ++j) hits += iou::screen_overlap(iou::screen_none(), iou::screen_box(b + 4 * j)) ? 1 : 0;
    return hits;
}
