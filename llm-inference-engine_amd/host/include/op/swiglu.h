// Include path of the reference (kuiper/include/op/swiglu.h) kept for drop-in sources: SwiGLULayer is
// declared with the other support layers.
#pragma once
#include "op/support_layers.h"
