// Process-wide handle on the HIP side (one madicp_ctx on the device named by MAD_ICP_DEVICE, default 0), shared by
// every host object (MADtree, MADicp, Pipeline).  Every failure of the C ABI is turned into a std::runtime_error
// carrying madicp_last_error(): the host classes have no CPU implementation to fall back to.
//
// Sharing rules.  The C ABI is not re-entrant, so every host-class method that talks to the device holds
// Device::mutex() for the duration of its call sequence: several Pipelines / wrappers may live in one process and be
// driven from different threads (their device work is serialised on the context's streams).  Device ids belong to
// the context GENERATION that issued them: after Device::shutdown() a host object neither releases its old ids into
// a newer context nor creates a context from its destructor.
#pragma once
#include <mutex>
#include <stdexcept>
#include <string>

#include "madicp_hip.h"

namespace madicp_host {

inline void check(int rc, const char* what) {
  if (rc != MADICP_OK) throw std::runtime_error(std::string(what) + ": " + madicp_last_error());
}

class Device {
 public:
  static madicp_ctx* ctx();      // creates the context on first use; throws if no MI355X / HIP device is usable
  static madicp_ctx* current(unsigned generation);  // the live context if it is still that generation, else nullptr; never creates
  static unsigned generation();  // bumped by every shutdown()
  static void shutdown();        // releases the context (tests)
  static std::recursive_mutex& mutex();
};

using DeviceLock = std::lock_guard<std::recursive_mutex>;

}  // namespace madicp_host
