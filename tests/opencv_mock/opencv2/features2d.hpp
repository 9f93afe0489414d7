// TEST-ONLY stand-in (see core.hpp next to this file): cv::KeyPoint lives in the core mock.
#include "core.hpp"
