// Include path of the reference (kuiper/include/op/embedding.h) kept for drop-in sources: EmbeddingLayer is
// declared with the other support layers.
#pragma once
#include "op/support_layers.h"
