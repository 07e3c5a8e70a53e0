#pragma once
#include "mat4x4.hpp"
