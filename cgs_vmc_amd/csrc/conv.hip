// Convolutional ansatz kernels (Conv2DNetwork / ResNet2D, wavefunctions.py:531-615, 710-809) for
// gfx950.  See DESIGN.md 4 "Convolutional ansatz".
#include "common.hpp"
