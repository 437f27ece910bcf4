// tests/cpp/refstub -- TEST INFRASTRUCTURE (see ../../README.md): QPInput.h includes this; nothing on the path uses it.
#ifndef REFSTUB_BLF_VECTORS_COLLECTION_SERVER_H
#define REFSTUB_BLF_VECTORS_COLLECTION_SERVER_H
namespace BipedalLocomotion { namespace YarpUtilities { class VectorsCollectionServer {}; } }
#endif
