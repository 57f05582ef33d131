#include "op/support_layers.h"
