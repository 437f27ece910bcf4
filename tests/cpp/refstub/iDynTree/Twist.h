// tests/cpp/refstub/iDynTree/Twist.h -- TEST INFRASTRUCTURE, not iDynTree (see ../README.md)
#include "Core.h"
