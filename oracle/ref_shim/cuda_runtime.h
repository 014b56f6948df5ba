// Stand-in for <cuda_runtime.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref build): the reference's generated
// Jacobian header (APP/bundle_adjustment/joint_optimization_jacobians.h) is marked __host__ __device__ so that
// its CUDA path can share it; the host compiler only needs the qualifiers to vanish.
#ifndef CBA_REF_SHIM_CUDA_RUNTIME_
#define CBA_REF_SHIM_CUDA_RUNTIME_
#define __host__
#define __device__
#ifndef __forceinline__
#define __forceinline__ inline
#endif
#endif
