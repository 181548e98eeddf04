// Decoder engine interface (decoder.hip) used by the C ABI in engine.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/maskbit_hip.h"

namespace mb {
mb_dec* dec_create(const mb_dec_cfg& cfg, int max_batch, std::string& err);
void dec_destroy(mb_dec* d);
int dec_load(mb_dec* d, const char* name, const float* data, const int64_t* shape, int ndim, hipStream_t s, std::string& err);
int dec_decode(mb_dec* d, const int64_t* tokens, float* img_nchw, uint8_t* img_nhwc_u8, int B, hipStream_t s, std::string& err);
int dec_saturation_count(mb_dec* d, unsigned* count, bool reset, hipStream_t s);   // synchronises the stream
int enc_encode(mb_dec* d, const float* img, int64_t* indices, float* zq, float* zraw, int B, hipStream_t s, std::string& err);
}  // namespace mb
