// tests/cpp/refstub/iDynTree/Rotation.h -- TEST INFRASTRUCTURE, not iDynTree (see ../README.md)
#include "Core.h"
