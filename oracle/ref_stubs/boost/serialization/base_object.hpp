#pragma once
#include "nvp.hpp"
