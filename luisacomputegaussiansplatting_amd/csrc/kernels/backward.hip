// backward.hip -- render-backward and preprocess-backward kernels (filled in below).
#include "launch.hpp"
