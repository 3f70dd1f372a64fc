// pt_guided.h — guided integrator state (SD-tree on device + host refinement). Filled in below.
#pragma once
#include "pt_types.h"
namespace pt {
struct GuidedState;
void guided_free(GuidedState* g);
}  // namespace pt
