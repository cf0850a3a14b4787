// Question-encoder entry points (placeholder until the encoder kernels land in this round).
#include "vqa_common.h"

extern "C" int vqa_encoder_create(vqa_encoder** out, int, const vqa_encoder_config*, const vqa_encoder_weights*, int32_t) {
    if (out) *out = nullptr;
    vqa_set_error("vqa_encoder_create: the encoder kernels are not part of this build yet");
    return VQA_EINVAL;
}
extern "C" void vqa_encoder_destroy(vqa_encoder*) {}
extern "C" int vqa_encoder_forward(vqa_encoder*, const int32_t*, const int32_t*, int32_t, int32_t, int32_t, int32_t, float*,
                                   void*) {
    vqa_set_error("vqa_encoder_forward: the encoder kernels are not part of this build yet");
    return VQA_EINVAL;
}
