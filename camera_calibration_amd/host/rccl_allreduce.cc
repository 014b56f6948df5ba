// Native RCCL all-reduce callback (include/cba_rccl.h): what a C++ host passes as cba_config.allreduce / allreduce_user.
#include "../../include/cba_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

static_assert(sizeof(ncclUniqueId) == CBA_RCCL_ID_BYTES, "ncclUniqueId size");

struct cba_rccl {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int device = 0, rank = 0, world = 1;
};

static thread_local std::string g_err;
static int fail(const char* what, const char* detail) {
  g_err = std::string(what) + ": " + detail;
  return -1;
}
#define RCCL_TRY(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return fail(#expr, ncclGetErrorString(_r)); } while (0)
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(#expr, hipGetErrorString(_e)); } while (0)

extern "C" {

const char* cba_rccl_last_error(void) { return g_err.c_str(); }

int cba_rccl_unique_id(char id[CBA_RCCL_ID_BYTES]) {
  ncclUniqueId u;
  RCCL_TRY(ncclGetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

int cba_rccl_create(int rank, int world, const char id[CBA_RCCL_ID_BYTES], int device, cba_rccl** out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail("cba_rccl_create", "bad argument");
  HIP_TRY(hipSetDevice(device));
  cba_rccl* c = new cba_rccl();
  c->device = device; c->rank = rank; c->world = world;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  if (ncclCommInitRank(&c->comm, world, u, rank) != ncclSuccess) { delete c; return fail("ncclCommInitRank", "failed"); }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { ncclCommDestroy(c->comm); delete c; return fail("hipStreamCreate", "failed"); }
  *out = c;
  return 0;
}

int cba_rccl_create_via_file(int rank, int world, const char* path, int device, cba_rccl** out) {
  if (!path) return fail("cba_rccl_create_via_file", "null path");
  char id[CBA_RCCL_ID_BYTES];
  if (rank == 0) {
    if (cba_rccl_unique_id(id) != 0) return -1;
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) { if (f) std::fclose(f); return fail("cba_rccl_create_via_file", "cannot write the id file"); }
    std::fclose(f);
    if (std::rename(tmp.c_str(), path) != 0) return fail("cba_rccl_create_via_file", "rename failed");
  } else {
    bool got = false;
    for (int i = 0; i < 6000 && !got; ++i) {           // up to 60 s
      FILE* f = std::fopen(path, "rb");
      if (f) { got = std::fread(id, 1, sizeof(id), f) == sizeof(id); std::fclose(f); }
      if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    if (!got) return fail("cba_rccl_create_via_file", "timed out waiting for the id file");
  }
  return cba_rccl_create(rank, world, id, device, out);
}

void cba_rccl_destroy(cba_rccl* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
  if (c->comm) ncclCommDestroy(c->comm);
  delete c;
}

int cba_rccl_allreduce(void* device_ptr, int64_t count, void* user) {
  cba_rccl* c = static_cast<cba_rccl*>(user);
  if (!c || !device_ptr || count < 0) return fail("cba_rccl_allreduce", "bad argument");
  if (count == 0) return 0;
  // the engine calls with its own stream idle (include/cba.h), so the buffer is complete; the reduction runs on this
  // communicator's stream and only that stream is waited for
  RCCL_TRY(ncclAllReduce(device_ptr, device_ptr, (size_t)count, ncclDouble, ncclSum, c->comm, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
