// TEST-ONLY stand-in (see opencv2/core.hpp in this directory)
#pragma once
#include <opencv2/core.hpp>
