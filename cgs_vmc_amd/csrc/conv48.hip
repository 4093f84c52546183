// 33 .. 48 filters: three channel blocks
#define CONV_NCB 3
#include "conv_wide.hpp"
