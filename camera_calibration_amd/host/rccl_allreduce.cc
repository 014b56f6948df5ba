// Native RCCL all-reduce callback (include/cba_rccl.h): what a C++ host passes as cba_config.allreduce / allreduce_user.
#include "../../include/cba_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <cstdlib>
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

// Waits for the id file and reads it.  A file older than `max_age_s` is a leftover of a run that died before it could remove
// its file (see below) and is ignored -- rank 0 of THIS launch replaces it.  0 = read, -1 = timed out.
int cba_rccl_debug_read_id_file(const char* path, char id[CBA_RCCL_ID_BYTES], int timeout_ms, int max_age_s) {
  for (int waited = 0; waited <= timeout_ms; waited += 10) {
    struct stat st;
    if (stat(path, &st) == 0 && st.st_size == CBA_RCCL_ID_BYTES && std::difftime(std::time(nullptr), st.st_mtime) <= max_age_s) {
      FILE* f = std::fopen(path, "rb");
      if (f) {
        const bool got = std::fread(id, 1, CBA_RCCL_ID_BYTES, f) == CBA_RCCL_ID_BYTES;
        std::fclose(f);
        if (got) return 0;
      }
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  return -1;
}

// The id file lives only while the communicator is being set up: rank 0 removes whatever is at `path` first, writes the new id
// atomically (rename), every rank joins, a one-element all-reduce proves that ALL ranks have read the file, and rank 0 removes
// it again.  A later launch with the same path therefore never finds this run's id; the leftover of a run that crashed in
// between is recognised by its age (ranks start within seconds of each other, a stale file is minutes old or older).
int cba_rccl_create_via_file(int rank, int world, const char* path_in, int device, cba_rccl** out) {
  if (!path_in || !out) return fail("cba_rccl_create_via_file", "null argument");
  *out = nullptr;
  // launch nonce in the file name: a rank that starts before rank 0 must never accept the leftover of an earlier launch that is
  // still young enough for the age test (crash + quick relaunch, clock skew).  CBA_RCCL_NONCE, or MASTER_PORT (the same on every
  // rank of one torchrun / mpirun launch, different between launches), is appended when set.
  std::string eff(path_in);
  if (const char* n = std::getenv("CBA_RCCL_NONCE")) eff += std::string(".") + n;
  else if (const char* mp = std::getenv("MASTER_PORT")) eff += std::string(".") + mp;
  const char* path = eff.c_str();
  char id[CBA_RCCL_ID_BYTES];
  if (rank == 0) {
    ::unlink(path);
    if (cba_rccl_unique_id(id) != 0) return -1;
    const std::string tmp = eff + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) { if (f) std::fclose(f); return fail("cba_rccl_create_via_file", "cannot write the id file"); }
    std::fclose(f);
    if (std::rename(tmp.c_str(), path) != 0) return fail("cba_rccl_create_via_file", "rename failed");
  } else {
    if (cba_rccl_debug_read_id_file(path, id, /*timeout_ms*/ 60000, /*max_age_s*/ 120) != 0)
      return fail("cba_rccl_create_via_file", "timed out waiting for the id file");
  }
  int rc = cba_rccl_create(rank, world, id, device, out);
  if (rc != 0) { if (rank == 0) ::unlink(path); return rc; }
  // handshake: when it returns on rank 0, every rank has passed ncclCommInitRank, i.e. has read the file
  double* one = nullptr;
  if (hipMalloc(&one, sizeof(double)) != hipSuccess) rc = fail("cba_rccl_create_via_file", "hipMalloc failed (handshake)");
  else {
    hipMemset(one, 0, sizeof(double));
    rc = cba_rccl_allreduce(one, 1, *out);
    hipFree(one);
  }
  if (rank == 0) ::unlink(path);
  if (rc != 0) { cba_rccl_destroy(*out); *out = nullptr; }       // no half-initialised communicator reaches the caller
  return rc;
}

// ncclCommCount of the communicator (-1 on error): what a host reports as the number of ranks its collectives really span
int cba_rccl_comm_count(cba_rccl* c) {
  if (!c || !c->comm) return -1;
  int n = -1;
  if (ncclCommCount(c->comm, &n) != ncclSuccess) return -1;
  return n;
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

// cba_collective_fn (include/cba.h): the collectives of the distributed reduced solve, same stream discipline
int cba_rccl_collective(int32_t op, void* sendbuf, void* recvbuf, int64_t count, void* user) {
  cba_rccl* c = static_cast<cba_rccl*>(user);
  if (!c || !recvbuf || count < 0) return fail("cba_rccl_collective", "bad argument");
  if (count == 0) return 0;
  switch (op) {
    case 0:   // CBA_COLL_ALLREDUCE_SUM
      RCCL_TRY(ncclAllReduce(recvbuf, recvbuf, (size_t)count, ncclDouble, ncclSum, c->comm, c->stream));
      break;
    case 1:   // CBA_COLL_REDUCE_SCATTER_SUM
      if (!sendbuf) return fail("cba_rccl_collective", "reduce-scatter without a send buffer");
      RCCL_TRY(ncclReduceScatter(sendbuf, recvbuf, (size_t)count, ncclDouble, ncclSum, c->comm, c->stream));
      break;
    case 2:   // CBA_COLL_ALLGATHER
      if (!sendbuf) return fail("cba_rccl_collective", "all-gather without a send buffer");
      RCCL_TRY(ncclAllGather(sendbuf, recvbuf, (size_t)count, ncclDouble, c->comm, c->stream));
      break;
    default:
      return fail("cba_rccl_collective", "unknown operation");
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
