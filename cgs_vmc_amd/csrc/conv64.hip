// 49 .. 64 filters: four channel blocks
#define CONV_NCB 4
#include "conv_wide.hpp"
