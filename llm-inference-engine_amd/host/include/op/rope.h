// Include path of the reference (kuiper/include/op/rope.h) kept for drop-in sources: RoPELayer is
// declared with the other support layers.
#pragma once
#include "op/support_layers.h"
