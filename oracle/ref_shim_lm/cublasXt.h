#pragma once
typedef void* cublasXtHandle_t;
enum { CUBLAS_STATUS_SUCCESS = 0, CUBLAS_STATUS_NOT_SUPPORTED = 15 };
enum { CUBLAS_OP_N = 0, CUBLAS_OP_T = 1 };
inline int cublasXtCreate(cublasXtHandle_t*) { return CUBLAS_STATUS_NOT_SUPPORTED; }
inline int cublasXtDestroy(cublasXtHandle_t) { return CUBLAS_STATUS_NOT_SUPPORTED; }
inline int cublasXtDeviceSelect(cublasXtHandle_t, int, int*) { return CUBLAS_STATUS_NOT_SUPPORTED; }
template <class... A> inline int cublasXtSgemm(A...) { return CUBLAS_STATUS_NOT_SUPPORTED; }
template <class... A> inline int cublasXtDgemm(A...) { return CUBLAS_STATUS_NOT_SUPPORTED; }
