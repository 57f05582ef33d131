// Include path of the reference (kuiper/include/op/add.h) kept for drop-in sources: VecAddLayer is
// declared with the other support layers.
#pragma once
#include "op/support_layers.h"
