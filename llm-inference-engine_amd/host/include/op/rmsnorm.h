// Include path of the reference (kuiper/include/op/rmsnorm.h) kept for drop-in sources: RmsNormLayer is
// declared with the other support layers.
#pragma once
#include "op/support_layers.h"
