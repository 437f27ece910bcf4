// tests/cpp/refstub/iDynTree/Transform.h -- TEST INFRASTRUCTURE, not iDynTree (see ../README.md)
#include "Core.h"
