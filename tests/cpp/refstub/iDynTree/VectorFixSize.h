// tests/cpp/refstub/iDynTree/VectorFixSize.h -- TEST INFRASTRUCTURE, not iDynTree (see ../README.md)
#include "Core.h"
