// TEST DOUBLE, see protoboard.hpp
#pragma once
#include "protoboard.hpp"
