// Radar-spectrum encoder (model/models_radar_encoder.py Encoder) - placeholder until the
// implicit-GEMM Conv3d kernels land: every entry point fails loudly, nothing falls back.
#include "dit.h"

namespace rald {
struct RadarEncoder::Impl {};
int RadarEncoder::create(int, int, int, int, int, int, DeviceArena*) { return 0; }
void RadarEncoder::expected_keys(const std::string&, std::set<std::string>&) const {}
int RadarEncoder::load_weight(const std::string& name, const float*, int64_t, Stager&) {
    RALD_CHECK(false, "radar encoder not built yet: cannot load '" + name + "'");
}
int RadarEncoder::load_token_weight(const std::string& name, const float*, int64_t, Stager&) {
    RALD_CHECK(false, "radar encoder not built yet: cannot load '" + name + "'");
}
int RadarEncoder::tokens(const float*, int, float**, hipStream_t) { RALD_CHECK(false, "radar encoder not built yet"); }
int RadarEncoder::encode(const float*, int, int, float**, hipStream_t) { RALD_CHECK(false, "radar encoder not built yet"); }
RadarEncoder::~RadarEncoder() { delete impl; }
}  // namespace rald
