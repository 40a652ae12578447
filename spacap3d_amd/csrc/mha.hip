// placeholder until the fused attention kernel lands (next commit)
#include "common.hpp"
extern "C" int spacap_mha_fwd_f32(const float *, const float *, const float *, long, long, long, long, long,
                                  long, long, long, long, const uint8_t *, long, long, const float *, long,
                                  long, long, int, int, int, int, int, float, float, uint64_t, float *,
                                  float *, float *, spacap_stream_t) {
  spacap::set_error("spacap_mha_fwd_f32: not built yet");
  return SPACAP_E_INVALID;
}
extern "C" int spacap_mha_bwd_f32(const float *, const float *, const float *, long, long, long, long, long,
                                  long, long, long, long, const uint8_t *, long, long, const float *, long,
                                  long, long, int, int, int, int, int, float, float, uint64_t, const float *,
                                  const float *, const float *, float *, float *, float *, spacap_stream_t) {
  spacap::set_error("spacap_mha_bwd_f32: not built yet");
  return SPACAP_E_INVALID;
}
