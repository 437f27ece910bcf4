// tests/cpp/refstub/iDynTree/Position.h -- TEST INFRASTRUCTURE, not iDynTree (see ../README.md)
#include "Core.h"
