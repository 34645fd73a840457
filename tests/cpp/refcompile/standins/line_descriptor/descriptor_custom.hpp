// TEST-ONLY stand-in
#pragma once
#include <line_descriptor_custom.hpp>
