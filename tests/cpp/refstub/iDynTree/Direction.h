// tests/cpp/refstub/iDynTree/Direction.h -- TEST INFRASTRUCTURE, not iDynTree (see ../README.md)
#include "Core.h"
