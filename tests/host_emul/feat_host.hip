// tests/host_emul/feat_host.hip -- TEST INFRASTRUCTURE.  The __host__ __device__ pieces of csrc/feat_math.h (the pre-filter's
// scale, checkpoints and the track-side record adjustment) called on the CPU, so that tests/test_prefilter_bound_cpu.py's NumPy
// restatement is checked against the product's own functions and not only against itself.  Built with hipcc (host code only
// is executed; no HIP API is called, no GPU needed).
#include "../../bot-sort-onnx-tensorrt_amd/csrc/feat_math.h"

extern "C" float feat_host_i8_scale(int d) { return feat::i8_scale(d); }
extern "C" void feat_host_checkpoints(int kt_all, int *out4) {
    int c[feat::I8_NCHK];
    feat::i8_checkpoints(kt_all, c);
    for (int k = 0; k < feat::I8_NCHK; ++k) out4[k] = c[k];
}
extern "C" int feat_host_meta_floats() { return feat::I8_META; }
extern "C" void feat_host_track_side(float *meta, float nrm, float gate) { feat::i8_track_side(meta, nrm, gate); }
