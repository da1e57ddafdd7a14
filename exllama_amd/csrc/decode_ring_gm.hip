// The group-size 32 / 64 instantiations of dec_ring_kernel as a translation unit of their own (see the note on EXL_RING_PART in
// decode_ring.hip): halves the longest compile of a parallel build.
#define EXL_RING_PART 1
#include "decode_ring.hip"
