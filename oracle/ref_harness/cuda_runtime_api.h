/* Empty stand-in for <cuda_runtime_api.h>; see cupti.h in this directory. TEST INFRASTRUCTURE ONLY. */
#pragma once
