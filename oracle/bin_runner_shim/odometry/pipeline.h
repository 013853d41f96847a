// ORACLE test infrastructure (oracle/build_bin_runner.sh): where the reference's apps/cpp_runners/bin_runner.cpp says
// `#include <odometry/pipeline.h>` (bin_runner.cpp:43) it gets the PRODUCT's Pipeline — same class name, constructor and
// methods (mad_icp_amd/csrc/host/pipeline.h mirrors mad_icp/src/odometry/pipeline.h:45-103).
#pragma once
#include "../../../mad_icp_amd/csrc/host/pipeline.h"
