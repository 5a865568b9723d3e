// RCCL binding and rendezvous of smj_comm_init / smj_allgather_returns / smj_comm_destroy (include/smj.h).  HOST code only -- no HIP
// call, no rccl header: smj_capi.hip wraps these with hipSetDevice, and tests/rccl_stub/comm_harness.cpp compiles the same file with
// g++ to run the rendezvous at world size 2 on a box without a GPU (against tests/rccl_stub/librccl_stub.so).
#pragma once
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <string>

// the five RCCL entry points this path uses, with their ABI spelled out (rccl.h: ncclUniqueId = 128 opaque bytes passed by value,
// ncclComm_t / hipStream_t = pointers, ncclResult_t / ncclDataType_t = int-sized enums; smj_capi.hip static_asserts the match)
struct SmjNcclId { char internal[128]; };
enum { SMJ_NCCL_SUCCESS = 0, SMJ_NCCL_FLOAT32 = 7 };

// RCCL is bound at run time (dlopen): a single-GPU user never loads it, and inside a PyTorch process the copy PyTorch has
// already mapped is reused instead of a second one.  SMJ_RCCL_LIB names another library file (a site's own build; the tests' stub).
struct RcclApi {
  void* h = nullptr;
  int (*GetUniqueId)(SmjNcclId*) = nullptr;
  int (*CommInitRank)(void**, int, SmjNcclId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, void*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi* rccl_api(std::string& err) {
  static RcclApi api;
  if (api.h) return &api;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  if (const char* own = getenv("SMJ_RCCL_LIB")) {
    h = dlopen(own, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { err = std::string("SMJ_RCCL_LIB: ") + dlerror(); return nullptr; }
  }
  for (int i = 0; !h && i < 3; i++) h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
  for (int i = 0; !h && i < 3; i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) { err = std::string("librccl not found: ") + dlerror(); return nullptr; }
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
  api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy || !api.GetErrorString) {
    err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
    return nullptr;
  }
  api.h = h;
  return &api;
}

struct SmjComm {
  void* comm = nullptr;   // ncclComm_t
  int rank = 0, world = 1;
};

static uint64_t smj_job_nonce() {
  // a hash of what every rank of ONE job shares and two jobs do not: SMJ_JOB_NONCE if the launcher sets it, else MASTER_ADDR :
  // MASTER_PORT : TORCHELASTIC_RUN_ID : WORLD_SIZE as torchrun exports them
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const char* sv) { for (const char* p = sv ? sv : ""; *p; p++) { h ^= (unsigned char)*p; h *= 1099511628211ull; } h ^= 0xff; h *= 1099511628211ull; };
  if (getenv("SMJ_JOB_NONCE")) mix(getenv("SMJ_JOB_NONCE"));
  else { mix(getenv("MASTER_ADDR")); mix(getenv("MASTER_PORT")); mix(getenv("TORCHELASTIC_RUN_ID")); mix(getenv("WORLD_SIZE")); }
  return h;
}

// Rank 0 creates the ncclUniqueId and publishes it atomically in the file `id_path`; the other ranks wait for it (<= timeout_s).
// The id file: {magic, job nonce, ncclUniqueId}.  A file left behind by an earlier job at the same path -- right size, wrong job --
// is not accepted, and rank 0 removes whatever is there before it publishes.  (With torch.distributed up, parallel.init_comm also
// puts a barrier between that removal and the readers' first look.)  Returns 0, or -7 with `err` set.
static int smj_comm_rendezvous(RcclApi* R, int rank, int world, const char* id_path, double timeout_s, SmjNcclId* id, std::string& err) {
  char buf[768];
  struct IdFile { char magic[8]; uint64_t nonce; SmjNcclId id; } rec;
  memset(&rec, 0, sizeof rec);
  memcpy(rec.magic, "SMJRCCL1", 8);
  rec.nonce = smj_job_nonce();
  memset(id, 0, sizeof *id);
  if (rank == 0) {
    if (world > 1) remove(id_path);   // a stale file of an earlier job must not be readable while the new id is being made
    const int r = R->GetUniqueId(id);
    if (r != SMJ_NCCL_SUCCESS) { snprintf(buf, sizeof buf, "ncclGetUniqueId: %s", R->GetErrorString(r)); err = buf; return -7; }
    if (world > 1) {   // publish atomically: write a temporary, then rename
      rec.id = *id;
      std::string tmp = std::string(id_path) + ".tmp";
      FILE* f = fopen(tmp.c_str(), "wb");
      if (!f || fwrite(&rec, sizeof rec, 1, f) != 1) { if (f) fclose(f); err = "cannot write " + tmp; return -7; }
      fclose(f);
      if (rename(tmp.c_str(), id_path) != 0) { err = std::string("cannot publish ") + id_path; return -7; }
    }
    return 0;
  }
  const double t_end = (timeout_s > 0 ? timeout_s : 120.0);
  double waited = 0;
  bool foreign = false;
  for (;;) {
    FILE* f = fopen(id_path, "rb");
    if (f) {
      IdFile got;
      const size_t n = fread(&got, 1, sizeof got, f);
      fclose(f);
      if (n == sizeof got && !memcmp(got.magic, rec.magic, 8) && got.nonce == rec.nonce) { *id = got.id; return 0; }
      foreign = foreign || n > 0;   // something is there, but not this job's record: keep waiting for rank 0 to replace it
    }
    if (waited >= t_end) {
      snprintf(buf, sizeof buf, foreign ? "timed out: the RCCL id file %s belongs to another job (stale file? nonce mismatch)" : "timed out waiting for the RCCL id file %s", id_path);
      err = buf;
      return -7;
    }
    struct timespec ts = {0, 20 * 1000 * 1000};
    nanosleep(&ts, nullptr);
    waited += 0.02;
  }
}

static int smj_comm_open(SmjComm& cm, int rank, int world, const char* id_path, double timeout_s, std::string& err) {
  char buf[512];
  if (world < 1 || rank < 0 || rank >= world) { snprintf(buf, sizeof buf, "bad rank %d / world %d", rank, world); err = buf; return -1; }
  if (cm.comm) { err = "communicator already initialised"; return -1; }
  if (world > 1 && (!id_path || !*id_path)) { err = "id_path is required for world > 1"; return -1; }
  RcclApi* R = rccl_api(err);
  if (!R) return -7;
  SmjNcclId id;
  const int rc = smj_comm_rendezvous(R, rank, world, id_path, timeout_s, &id, err);
  if (rc) return rc;
  const int r = R->CommInitRank(&cm.comm, world, id, rank);
  if (r != SMJ_NCCL_SUCCESS) { cm.comm = nullptr; snprintf(buf, sizeof buf, "ncclCommInitRank: %s", R->GetErrorString(r)); err = buf; return -7; }
  cm.rank = rank;
  cm.world = world;
  return 0;
}

// world > 1 only (the single-GPU gather is a device copy, done by the caller)
static int smj_comm_allgather(SmjComm& cm, const float* send, float* recv, int count, void* stream, std::string& err) {
  RcclApi* R = rccl_api(err);
  if (!R) return -7;
  const int r = R->AllGather(send, recv, (size_t)count, SMJ_NCCL_FLOAT32, cm.comm, stream);
  if (r != SMJ_NCCL_SUCCESS) { err = std::string("ncclAllGather: ") + R->GetErrorString(r); return -7; }
  return 0;
}

static void smj_comm_close(SmjComm& cm) {
  if (cm.comm) {
    std::string e;
    RcclApi* R = rccl_api(e);
    if (R) R->CommDestroy(cm.comm);
  }
  cm = SmjComm();
}
