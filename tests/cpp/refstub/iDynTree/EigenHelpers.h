// tests/cpp/refstub/iDynTree/EigenHelpers.h -- TEST INFRASTRUCTURE, not iDynTree (see ../README.md)
#include "Core.h"
