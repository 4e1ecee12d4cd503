// Fused post-step physics - see below.
#include <hip/hip_runtime.h>
#include "kernels.h"
namespace ace {}
