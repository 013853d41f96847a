// Process-wide handle on the HIP side (one madicp_ctx on the device named by MAD_ICP_DEVICE, default 0).
// Every failure of the C ABI is turned into a std::runtime_error carrying madicp_last_error(): the host
// classes have no CPU implementation to fall back to.
#pragma once
#include <stdexcept>
#include <string>

#include "madicp_hip.h"

namespace madicp_host {

inline void check(int rc, const char* what) {
  if (rc != MADICP_OK) throw std::runtime_error(std::string(what) + ": " + madicp_last_error());
}

class Device {
 public:
  static madicp_ctx* ctx();  // creates the context on first use; throws if no MI355X / HIP device is usable
  static void shutdown();    // releases it (tests)
};

}  // namespace madicp_host
