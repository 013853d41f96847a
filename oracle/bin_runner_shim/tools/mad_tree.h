// ORACLE test infrastructure (oracle/build_bin_runner.sh): `#include <tools/mad_tree.h>` of bin_runner.cpp:44 -> the
// product's MADtree and ContainerType (mad_icp_amd/csrc/host/mad_tree.h, types.h mirror mad_icp/src/tools/mad_tree.h:42-102).
#pragma once
#include "../../../mad_icp_amd/csrc/host/mad_tree.h"
#include "../../../mad_icp_amd/csrc/host/types.h"
