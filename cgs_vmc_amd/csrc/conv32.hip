// 17 .. 32 filters: two channel blocks
#define CONV_NCB 2
#include "conv_wide.hpp"
