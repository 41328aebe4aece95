// tests/host_emul/cost_host.hip -- TEST INFRASTRUCTURE.  csrc/cost_math.h (the fused association costs of a pair and the face
// fix-up, what k_fuse1 / k_fuse3 evaluate per element) on the CPU, element by element like the kernels.
#include "../../bot-sort-onnx-tensorrt_amd/csrc/cost_math.h"
extern "C" void cost_host_fuse1(long count, const double *iou_d, const float *body, const float *face, double prox, float app, double *out) {
    for (long k = 0; k < count; ++k) out[k] = fuse1(iou_d[k], body[k], face ? face_fixup(face[k]) : 0.0f, prox, app);
}
extern "C" void cost_host_fuse3(long count, const double *iou_d, const float *body, double prox, float app, double *out) {
    for (long k = 0; k < count; ++k) out[k] = fuse3(iou_d[k], body[k], prox, app);
}
extern "C" void cost_host_round_f16(long count, const double *in, double *out) {
    for (long k = 0; k < count; ++k) out[k] = round_to_f16(in[k]);
}
